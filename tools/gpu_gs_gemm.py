"""The two dense products behind the fused weight-gradient kernel's per-edge radial gradient gs [E, ch] (hamgnn_amd/backward_mp.py:tp_weight_grads_fused):
g_W3 = h^T gs  ([H, E] @ [E, ch]) and g_h = gs W3^T ([E, ch] @ [ch, H]) at Si-512 / set-A sizes, by library and formulation.
python tools/gpu_gs_gemm.py [--edges 44032] [--ch 3589]"""
import argparse, time
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--edges", type=int, default=44032)
ap.add_argument("--ch", type=int, default=3589)
a = ap.parse_args()
dev = torch.device("cuda:0")
E, C, H = a.edges, a.ch, 64
torch.manual_seed(0)
gs = torch.randn(E, C, device=dev)
h = torch.randn(E, H, device=dev)
W3 = torch.randn(H, C, device=dev)

def timed(f, n=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

forms = {
    "gW3: h.t() @ gs": lambda: h.t() @ gs,
    "gW3: (gs.t() @ h).t()": lambda: (gs.t() @ h).t(),
    "gW3: split-K 16 (bmm)": lambda: torch.bmm(h[: E // 16 * 16].view(16, E // 16, H).transpose(1, 2), gs[: E // 16 * 16].view(16, E // 16, C)).sum(0),
    "gW3: split-K 64 (bmm)": lambda: torch.bmm(h[: E // 64 * 64].view(64, E // 64, H).transpose(1, 2), gs[: E // 64 * 64].view(64, E // 64, C)).sum(0),
    "gh: gs @ W3.t()": lambda: gs @ W3.t(),
    "gh: gs @ W3t (contiguous)": (lambda W3t: (lambda: gs @ W3t))(W3.t().contiguous()),
}
for lib in ("default", "hipblaslt", "cublas"):
    try:
        if lib != "default":
            torch.backends.cuda.preferred_blas_library(lib)
    except Exception as ex:
        print(lib, "not selectable:", ex); continue
    for name, f in forms.items():
        try:
            ms = timed(f)
            print(f"{lib:10s} {name:32s} {ms:7.3f} ms  {2 * E * C * H / ms / 1e9:7.1f} TFLOP/s  {gs.numel() * 4 / ms / 1e9:6.2f} TB/s of gs")
        except Exception as ex:
            print(lib, name, "failed:", str(ex)[:100])
