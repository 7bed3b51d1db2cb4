"""A WHOLE forward (backbone + head, every kernel of the library and the torch glue between them) while the MFMA-only aggressor (tests/csrc/xdl_aggressor.hip) runs on a side stream,
against the same forward on an idle GPU, bit for bit.  profiles/r06_tp_is.md section 8.     python tools/gpu_aggressor3.py /tmp/libxdl_aggressor.so [--workload sio2_10k]"""
import argparse, ctypes, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
from hamgnn_amd.models.model import Model
ap = argparse.ArgumentParser()
ap.add_argument("lib"); ap.add_argument("--workload", default="sio2_10k"); ap.add_argument("--forwards", type=int, default=8); ap.add_argument("--grid", type=int, default=256)
a = ap.parse_args()
AG = ctypes.CDLL(a.lib)
AG.aggressor_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
irr = B.IRREPS["A"]
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = Model(HamGNNConvE3(B.make_cfg(irr)), HamGNNPlusPlusOut(irr, irr, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True,
                                                             soc_switch=False, calculate_sparsity=True, zero_point_shift=False)).to(dev)
g = B.make_graph(a.workload, 19).to(dev)
def fwd():
    with torch.no_grad():
        return model(g)["hamiltonian"]
ref = fwd().clone()
for _ in range(2):
    assert torch.equal(fwd(), ref)
torch.cuda.synchronize(); t0 = time.perf_counter(); fwd(); torch.cuda.synchronize(); t_alone = (time.perf_counter() - t0) * 1e3
side = torch.cuda.Stream()
names = {0: "dependent chains of v_mfma_f32_16x16x32_f16", 1: "independent v_mfma_f32_16x16x32_f16", 3: "chains of v_mfma_f32_16x16x4_f32 (control)", 4: "chains of v_mfma_f32_16x16x32_bf16"}
for mode in (3, 0, 1, 4):
    torch.cuda.synchronize(); t0 = time.perf_counter(); AG.aggressor_launch(mode, a.grid, 20000, ctypes.c_void_p(side.cuda_stream)); torch.cuda.synchronize(); t_ag = (time.perf_counter() - t0) * 1e3
    iters = max(1000, int(20000 * (3 * t_alone + 30.0) / max(t_ag, 1e-3)))
    bad, worst, overlapped, tv = 0, 0.0, 0, []
    for _ in range(a.forwards):
        torch.cuda.synchronize()
        assert AG.aggressor_launch(mode, a.grid, iters, ctypes.c_void_p(side.cuda_stream)) == 0
        time.sleep(0.003)
        t0 = time.perf_counter()
        h = fwd()
        torch.cuda.current_stream().synchronize()
        tv.append((time.perf_counter() - t0) * 1e3)
        overlapped += int(not side.query())
        torch.cuda.synchronize()
        if not torch.equal(h, ref):
            bad += 1
            worst = max(worst, float((h - ref).abs().max() / ref.abs().max()))
    print(json.dumps({"victim": f"whole forward of {a.workload} (backbone + head)", "aggressor": names[mode], "forwards": a.forwards, "forwards_overlapped_to_their_end": overlapped,
                      "forwards_that_differ": bad, "worst_rel": worst, "ms_alone": round(t_alone, 2), "ms_with_aggressor": round(sum(tv) / len(tv), 2)}), flush=True)
