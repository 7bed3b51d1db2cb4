#!/usr/bin/env python
"""bench.py -- edges/sec of the equivariant message-passing forward (HamGNNConvE3 + HamGNNPlusPlusOut) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload sio2_10k|si512|mos2_1200|si2|uni8] [--irreps A|B]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is ONE inference forward of the whole hot path over one synthetic periodic crystal already resident in HBM:
backbone (embedding + num_layers x (ConvBlockE3 + PairInteractionBlock)) + pair read-out head (ham_only, symmetrize,
add_H0, no SOC), random-init weights (seed 666), fp32.  value = directed edges of the crystal * steps / time, whole job.
For N > 1 the undirected pairs are sharded over the ranks (hamgnn_amd/parallel.py) with one RCCL all-reduce of the node
aggregates per layer; the crystal (total work) is fixed => "scaling": "strong".
Prints ONE JSON line (rank 0) incl. `accuracy` (max|H - H_oracle| / max|H_oracle| and MAE of this model on a bounded sub-crystal, fp64 oracle
in a child process), `compile_s` (host planner + upload), for N > 1 `per_rank` (edges, edge-kernel / all-reduce / other ms per step of every
rank) and `ranks_seen`, `roofline` for the dominant kernel (the fused MessagePackBlock launches: hg_tp_is, or
hg_tp_fused on the fallback path; timed live with HIP events on the launch stream inside the timed region) and `cpu_baseline` (the oracle = unfused pure-torch port of the reference path,
timed on the host cores over a bounded sample of the same workload; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

IRREPS = {
    "A": "64x0e+64x0o+32x1o+16x1e+12x2o+25x2e+18x3o+9x3e+4x4o+9x4e+4x5o+4x5e+2x6e",       # shipped default (examples/V2.x/config.yaml)
    "B": "64x0e+32x1o+16x1e+8x2o+20x2e+8x3o+4x3e+4x4e",                                   # "lmax=4" set (examples/V1.0)
}
SH = "0e+1o+2e+3o+4e+5o"
# algorithmic cost of ONE fused MessagePackBlock per edge in the REFERENCE formulation (non-zero CG entries)
REF_FLOPS_PER_EDGE_BLOCK = {"A": 4.55e6, "B": 1.73e6}
REF_BYTES_PER_EDGE_BLOCK = {"A": 14432.0, "B": 7888.0}
# HBM bytes per edge of one MessagePackBlock launch from the rocprofv3 PMC passes in profiles/r04_tp_is_pmc.md (set-B: r02b_tp_is_hbm_pmc.md) / r01c_tp_fused_hbm_pmc.md
# (separate --pmc FETCH_SIZE / WRITE_SIZE runs on tests/bench_tp.py, 131072 edges; FETCH_SIZE x 2: gfx950 correction for
# 16-B/lane reads, MI355X_MICROARCH.md "HBM"): measured offline for this kernel build, scaled to the launch's edge count.
# kernel "is" = input-stationary tp_is_kernel, "seg" = segment-stationary tp_fused_kernel.  r5: ("is", "A") is measured ON THE BENCHMARKED LAUNCHES (FETCH_SIZE /
# WRITE_SIZE passes over `python bench.py`, sio2_10k: profiles/r05_tp_is_pmc.md, re-measured in r6: r06_tp_is_pmc.md), launch by launch of a forward: 6.6 (first ConvBlock: reduced program) / 24.0 (first
# PairInteractionBlock) / 26.6 (ConvBlock, fused scatter) / 28.2 (PairInteractionBlock, one row per edge) / 26.7 / 26.4 (last pair block: unread irreps left out) KB per edge; their mean is used (the synthetic
# bench_tp launch with random sender / receiver indices measured 34.0 KB per edge in r4: the real crystal's neighbour locality keeps more node rows in L2 / Infinity Cache)
PMC_HBM_BYTES_PER_EDGE_BLOCK = {("is", "A"): 23.6e3, ("is", "B"): 14.2e3, ("seg", "A"): 173.2e3, ("seg", "B"): 59.9e3}
# ms per million edges of ONE MessagePackBlock launch on one MI355X at full size, mean over the six launches of a forward (profiles/r05_bench_sio2_10k_setA.json: 34.52 ms
# per 822 350 edges; r05_bench_si512_setB.json: 0.835 ms per 44 032 edges): the yardstick of the N > 1 lines (per_rank.tp_is_efficiency_vs_1gpu = edge-proportional time
# at that rate / measured time)
REF_TP_MS_PER_MEDGE_LAUNCH = {"A": 41.98, "B": 18.96}
PEAK_FP32_TFLOPS = 157.3           # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
PEAK_HBM_GBS = 8000.0
PEAK_F16_TFLOPS = 2500.0           # MI355X_MICROARCH.md: dense BF16 / FP16 MFMA peak (16 x the fp32-input rate); prices the radial scale's half-precision products in `frac_of_mixed_pipe_floor`


def make_cfg(irreps, lite=False):
    return dict(num_types=96, irreps_edge_sh=SH, edge_sh_normalization="component", edge_sh_normalize=True,
                build_internal_graph=False, cutoff=26.0, rbf_func="bessel", num_radial=64, num_layers=3,
                irreps_node_features=irreps, use_kan=False, radial_MLP=[64, 64], correlation=2, num_hidden_features=16,
                radius_type="openmx", use_corr_prod=False, legacy_edge_update=False, lite_mode=bool(lite))


def make_graph(workload, nao, soc=False):
    from hamgnn_amd.data import synthetic as S
    if workload == "sio2_10k":
        g = S.amorphous_sio2(10002, seed=1)
    elif workload == "si512":
        g = S.si_diamond(4, 4, 4, jitter=0.05, seed=0)
    elif workload == "mos2_1200":
        g = S.mos2_monolayer(20, 20)
    elif workload == "si2":
        g = S.si_diamond(primitive=True)
    elif workload == "si64":
        g = S.si_diamond(2, 2, 2, jitter=0.05, seed=0)
    elif workload == "mos2_48":
        g = S.mos2_monolayer(4, 4)
    elif workload.startswith("sio2_"):
        g = S.amorphous_sio2(int(workload.split("_")[1]), seed=1)
    else:
        raise SystemExit(f"unknown workload {workload}")
    return S.add_random_targets(g, nao, seed=0, soc=soc)


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(workload, irreps_key, nao, budget_s=30.0, lite=False, soc=False, reps=3, full=False):
    """Oracle (unfused torch port of the reference op graph: one einsum chain per e3nn instruction, materialised `mid`, index_add_ scatter) on the
    host cores, on a BOUNDED sample of the same workload: the largest crystal of the workload's generator whose forward fits budget_s / (reps + 1)
    seconds, `reps` timed forwards (median reported, all times listed), the thread count picked from 8 ... all host threads by a calibration run.
    A reported baseline, not the target: the GPU / CPU ratio says the port is slow, not that the kernel is good (roofline.frac is the figure of merit)."""
    from oracle import e3, hamgnn_ref as R
    from hamgnn_amd.data import synthetic as S
    e3.CONTRACTION = "optimized"               # the timed port contracts weights first (what opt_einsum_fx does for e3nn's generated code); the parity oracle keeps the naive order
    ncpu = os.cpu_count() or 1
    if "HG_CPU_THREADS" in os.environ:
        cand = sorted({min(ncpu, int(c)) for c in os.environ["HG_CPU_THREADS"].split(",")})
    elif full:                                 # (beyond 32 threads the port gets slower on every host measured: the sweep stops at 64 -- a 256-thread forward takes minutes)
        cand = sorted({min(ncpu, c) for c in (8, 16, 32, 64)})
    else:
        cand = sorted({min(ncpu, c) for c in (8, 16, 32, 64, 128, ncpu)})
    torch.set_num_threads(cand[0])
    irreps = IRREPS[irreps_key]
    torch.manual_seed(666)
    model = R.HamGNNConvE3(make_cfg(irreps, lite)).float()
    head = R.HamGNNPlusPlusOut(irreps, irreps, nao_max=nao, ham_type="openmx", symmetrize=True, add_H0=True,
                               **(dict(soc_switch=True, soc_basis="so3") if soc else {})).float()

    def graph(n_atoms):
        if workload.startswith("sio2"):
            g = S.amorphous_sio2(n_atoms, seed=1)
        elif workload.startswith("mos2"):
            k = max(1, int(round((n_atoms / 3) ** 0.5)))
            g = S.mos2_monolayer(k, k)
        else:
            k = max(1, int(round((n_atoms / 8) ** (1 / 3))))
            g = S.si_diamond(k, k, k, jitter=0.05, seed=0)
        return S.add_random_targets(g, nao, seed=0, soc=soc)

    def run(g):
        t0 = time.perf_counter()
        with torch.no_grad():
            head(g, model(g))
        return time.perf_counter() - t0

    if full:
        # the whole configuration instead of a bounded sample (BASELINE config #2 is the one the port finishes in seconds: si512, set-B, 43 k edges): every
        # thread count up to the host's, `reps` forwards each, median per count -- minutes of host time, NOT part of the default run (tools/cpu_baseline_full.sh)
        g = make_graph(workload, nao, soc=soc)
        run(graph(12))
        sweep = {}
        for c in cand:
            torch.set_num_threads(c)
            run(g)
            sweep[c] = sorted(run(g) for _ in range(reps))[reps // 2]
        cores = min(sweep, key=sweep.get)
        return {"value": g.num_edges / sweep[cores], "unit": "edges/s", "cores": cores, "kind": "port", "reps": reps, "cpu_model": cpu_model(), "host_threads": ncpu,
                "edges_per_s_by_threads": {str(k): round(g.num_edges / v, 1) for k, v in sweep.items()}, "seconds_per_forward_by_threads": {str(k): round(v, 3) for k, v in sweep.items()},
                "contraction": e3.CONTRACTION,
                "sample": f"the WHOLE {workload} crystal ({g.num_nodes} atoms / {g.num_edges} directed edges), irreps set-{irreps_key}, median of {reps} forwards per thread count, fp32"}
    g0 = graph(12)
    run(g0)                                    # warm-up (first touch, thread pool, the cached 3j tensors)
    tried = {}
    for c in cand:                             # these small ops scale poorly with threads: try the counts in rising order, stop at the first that is slower
        torch.set_num_threads(c)               # than the one before it (measured on the 256-thread host: beyond 32 threads the port gets slower), keep the fastest
        tried[c] = run(g0)
        if len(tried) > 1 and tried[c] > 1.15 * min(tried.values()):
            break
    cores = min(tried, key=tried.get)
    torch.set_num_threads(cores)
    per_rep = max(2.0, (budget_s - sum(tried.values())) / (reps + 0.5))
    n_atoms = max(12, min(int(g0.num_nodes * per_rep / tried[cores]), 4000))
    g = graph(n_atoms) if n_atoms > g0.num_nodes else g0
    times = [run(g) for _ in range(reps)]
    med = sorted(times)[len(times) // 2]
    return {"value": g.num_edges / med, "unit": "edges/s", "cores": cores, "kind": "port", "reps": reps,
            "seconds_per_forward": [round(t, 2) for t in times], "threads_tried": {str(k): round(g0.num_edges / v, 1) for k, v in tried.items()},
            "host_threads": ncpu, "cpu_model": cpu_model(), "contraction": e3.CONTRACTION,
            "sample": f"{workload}-like crystal, {g.num_nodes} atoms / {g.num_edges} directed edges{', SOC/so3 head' if soc else ''}, median of {reps} forwards, fp32, "
                      f"torch {cores} threads (fastest of {sorted(tried)} tried in rising order on a {g0.num_edges}-edge calibration crystal, stopping at the first slower count: "
                      f"edges/s per count in threads_tried)"}


# BASELINE config #5 (Uni-HamGNN universal model, mixed-Z periodic-table batch): Z drawn from the 26-orbital OpenMX table -- light ... heavy,
# s / p / d / f shells -- 8 crystals of 32-128 atoms (seed 2)
UNI_ZS = (1, 6, 8, 14, 22, 26, 31, 42, 47, 56, 74, 79, 83)


def uni_config(irreps, soc):
    pre = dict(make_cfg(irreps), num_types=96)
    out = dict(nao_max=26, ham_type="openmx", ham_only=True, symmetrize=True, calculate_band_energy=False, num_k=4, k_path=None,
               band_num_control=None, soc_switch=soc, nonlinearity_type="gate", add_H0=True, spin_constrained=False, collinear_spin=False,
               minMagneticMoment=0.5)
    return dict(representation_nets=dict(HamGNN_pre=pre), output_nets=dict(HamGNN_out=out))


def run_uni8(args, dev):
    """--workload uni8: the two-model chain of Uni-HamiltonianPredictor.py:290-319 (non-SOC universal model -> Hon_nonsoc / Hoff_nonsoc ->
    SOC / so3 model with add_H_nonsoc) over 8 mixed-Z crystals of 32-128 atoms, nao 26, set-A.  A step = all 8 crystals through both
    models.  Three ways to issue the same work: one crystal per forward as the reference's DataLoader(batch_size=1) does (eager, and as
    HIP-graph replays), and all 8 crystals collated into ONE batch per model (the per-launch choices -- split edge-kernel launches for
    few tiles, one workgroup per tile otherwise -- are then made per batch instead of per crystal)."""
    import numpy as np
    from hamgnn_amd import ops, uni
    from hamgnn_amd.data import collate, synthetic as S
    from hamgnn_amd.graph_capture import CapturedForward
    from hamgnn_amd.models.model import Model
    irreps = IRREPS[args.irreps]
    models = {}
    for soc in (False, True):
        torch.manual_seed(666 + int(soc))
        rep, head = uni.build_hamgnn_components(uni_config(irreps, soc))
        models[soc] = Model(representation=rep, output=head).to(dev)
    pred = uni.HamiltonianPredictor(models[False], models[True], dev)
    rng = np.random.default_rng(2)
    pairs = []
    for k in range(8):
        base = S.random_cell(int(rng.integers(32, 129)), list(UNI_ZS), seed=60 + k, density=0.012)
        pairs.append((S.add_random_targets(type(base)(base), 26, seed=60 + k, soc=False), S.add_random_targets(type(base)(base), 26, seed=1060 + k, soc=True)))
    atoms, edges = [p[0].num_nodes for p in pairs], [p[0].num_edges for p in pairs]
    E_total = sum(edges)
    singles = [(a.to(dev), b.to(dev)) for a, b in pairs]
    batch_ns, batch_soc = collate([p[0] for p in pairs]).to(dev), collate([p[1] for p in pairs]).to(dev)

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps

    per_crystal = lambda: [pred.predict(a, b) for a, b in singles]
    batched = lambda: pred.predict(batch_ns, batch_soc)
    ops.PROFILE_EVENTS = None
    t_pc = timed(per_crystal, args.steps, args.warmup)
    t_b = timed(batched, args.steps, args.warmup)
    # consistency of the two issue orders (per-crystal rows == the batch's rows, crystal by crystal)
    with torch.no_grad():
        ob = batched()
        oc = per_crystal()
    hr = torch.cat([o["hamiltonian_real"] for o in oc], 0)
    batch_vs_single = float((ob["hamiltonian_real"] - hr).abs().max() / hr.abs().max())
    replays = [CapturedForward((lambda a=a, b=b: pred.predict(a, b))) for a, b in singles]
    t_gr = timed(lambda: [r() for r in replays], args.steps, args.warmup)
    lat = []
    for r in replays:                                          # per-crystal latency of one replayed chain
        lat.append(timed(r, max(3, args.steps), 1) * 1e3)
    replay_b = CapturedForward(batched)
    t_gb = timed(replay_b, args.steps, args.warmup)
    modes = {"per_crystal_eager": {"ms_per_step": t_pc * 1e3, "edges_per_s": E_total / t_pc},
             "per_crystal_graph_replay": {"ms_per_step": t_gr * 1e3, "edges_per_s": E_total / t_gr, "latency_ms_per_crystal": lat},
             "batched_eager": {"ms_per_step": t_b * 1e3, "edges_per_s": E_total / t_b},
             "batched_graph_replay": {"ms_per_step": t_gb * 1e3, "edges_per_s": E_total / t_gb}}
    best = max(modes, key=lambda m: modes[m]["edges_per_s"])
    # roofline of the dominant kernel at THIS size: the MessagePackBlock launches of the batched chain (HIP events on the launch stream, 12 per
    # step: two models x three layers x two blocks), SURVEY 8(d)'s reference-formulation flop count per edge and block
    ops.PROFILE_EVENTS = []
    for _ in range(3):
        batched()
    torch.cuda.synchronize()
    mp = [(s_.elapsed_time(e_) * 1e-3, rows) for (s_, e_, rows, tag) in ops.PROFILE_EVENTS if tag == "message_pack"]
    all_ev = sum(s_.elapsed_time(e_) * 1e-3 for (s_, e_, rows, tag) in ops.PROFILE_EVENTS)
    ops.PROFILE_EVENTS = None
    avg_s = sum(t for t, _ in mp) / max(1, len(mp))
    rows_l = sum(r for _, r in mp) / max(1, len(mp))
    ach = REF_FLOPS_PER_EDGE_BLOCK[args.irreps] * rows_l / avg_s / 1e12
    roofline = {"kernel": "tp_is_kernel (MessagePackBlock launches of the batched chain; split launches below 300 tiles)", "bound": "mfma", "achieved": ach,
                "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS, "traffic": None, "avg_launch_ms": avg_s * 1e3,
                "launches_timed": len(mp), "edges_per_launch": rows_l, "fused_program_launches_share_of_step": all_ev / 3 / t_b}
    res = {"metric": "edges/sec (equivariant MP forward)", "value": modes[best]["edges_per_s"], "unit": "edges/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": modes[best]["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"uni8: Uni-HamGNN two-model chain (non-SOC -> SOC/so3, add_H_nonsoc), 8 mixed-Z crystals of {min(atoms)}-{max(atoms)} atoms "
                                  f"({sum(atoms)} atoms, {E_total} directed edges), irreps set-{args.irreps}, nao_max 26, 3 layers per model",
                      "parallelism": f"single GPU, issue mode = {best}"},
           "roofline": roofline, "modes": modes, "batch_rows_vs_per_crystal_rows": batch_vs_single, "atoms": atoms, "edges": edges}
    print(json.dumps(res), flush=True)


def accuracy_vs_oracle(path):
    """Accuracy leg (child process, host cores): the fp64 oracle on the bounded sub-crystal the GPU run left in `path` (graph, both
    state_dicts, the GPU's Hamiltonian rows) -> max|H - H_oracle| / max|H_oracle| and the mean absolute error."""
    from oracle import hamgnn_ref as R
    blob = torch.load(path, weights_only=False)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        model = R.HamGNNConvE3(make_cfg(blob["irreps"], blob.get("lite", False)))
        head = R.HamGNNPlusPlusOut(blob["irreps"], blob["irreps"], nao_max=blob["nao"], ham_type="openmx", symmetrize=True, add_H0=True,
                                   **(dict(soc_switch=True, soc_basis="so3") if blob.get("soc") else {}))
    finally:
        torch.set_default_dtype(prev)
    for mod, sd in ((model, blob["backbone"]), (head, blob["head"])):
        res = mod.load_state_dict({k: v.double() for k, v in sd.items()}, strict=False)
        missing = set(res.missing_keys) & set(dict(mod.named_parameters()))
        assert not missing, sorted(missing)[:4]
    g = blob["graph"]
    g64 = type(g)({k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in g.items()})
    t0 = time.perf_counter()
    with torch.no_grad():
        Href = head(g64, model(g64))["hamiltonian"]
    H = blob["H"].double()
    return {"rel_max": float((H - Href).abs().max() / Href.abs().max()), "mae": float((H - Href).abs().mean()),
            "oracle_mean_abs": float(Href.abs().mean()), "tolerance": 1e-5,
            "sample": f"{blob['what']}: {g64.num_nodes} atoms / {g64.num_edges} directed edges, same weights as the timed model, "
                      f"fp32 HIP vs fp64 oracle ({time.perf_counter() - t0:.1f} s on the host)"}


class _LauncherExit(SystemExit):
    """exit of the process that only launched the ranks (not a rank's own failure)"""


def self_launch(n_gpus: int) -> int:
    """run `sys.argv` again as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py ...` and
    return the launcher's exit code"""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n_gpus) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sio2_10k")
    ap.add_argument("--irreps", default="A", choices=["A", "B"])
    ap.add_argument("--nao", type=int, default=19)
    ap.add_argument("--soc", action="store_true", help="SOC / so3 read-out (BASELINE config #3: MoS2 with spin-orbit coupling); the CPU baseline leg stays non-SOC")
    ap.add_argument("--no-mfma-probe", action="store_true", help="skip the 20 ms fp32-MFMA ceiling probe that follows the timed region (roofline.mfma_probe_tflops)")
    ap.add_argument("--lite", action="store_true", help="lite_mode MessagePackBlocks (message_passing.py:197-215: unweighted uvu products + o3.Linear + one combined radial scale); "
                    "runs on the lite instantiation of the input-stationary kernel (tp_is_kernel<., true>), the roofline then counts the planner's executed flops (SURVEY 8d's figures are for the default block)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-complete-pass", action="store_true", help="skip the extra pass with every path of the reference's op graph issued (value_complete_programs; N = 1 only, after the timed region)")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-full", action="store_true", help="with --cpu-baseline-only: the port on the WHOLE workload with a thread sweep (minutes; meant for --workload si512 --irreps B)")
    ap.add_argument("--accuracy-from", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--no-accuracy", action="store_true", help="skip the accuracy leg (fp64 oracle on a bounded sub-crystal, host cores)")
    args = ap.parse_args()
    if args.cpu_baseline_only:                       # child process of the N=1 run (hard wall-clock bound in the parent)
        if args.accuracy_from:
            print("ACCURACY " + json.dumps(accuracy_vs_oracle(args.accuracy_from)), flush=True)
            return
        print("CPU_BASELINE " + json.dumps(cpu_baseline(args.workload, args.irreps, args.nao, lite=args.lite, soc=args.soc, full=args.cpu_baseline_full)), flush=True)
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        # `python bench.py --gpus N` as ONE command (the reference's multi-GPU entry is one too: Lightning spawns the ranks, hamgnn/main.py:316-321): re-run this very
        # command line under torch.distributed.run, one process per GPU, static rendezvous on 127.0.0.1 (the container's hostname may not resolve); rank 0 prints the
        # JSON line on the inherited stdout, a failing rank makes the launcher -- and this process -- exit non-zero
        rc = self_launch(args.gpus)
        if rc:
            print("BENCH_LAUNCH_FAILURE " + json.dumps({"gpus": args.gpus, "launcher_rc": rc, "see": "the BENCH_RANK_FAILURE line(s) of the rank(s) that died first"}), file=sys.stderr, flush=True)
        raise _LauncherExit(rc)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): pass --gpus {world}, or run `python bench.py --gpus {args.gpus}` without a launcher")
    same_device = os.environ.get("HG_BENCH_SAME_DEVICE") == "1"
    if same_device:                                        # test hook: validate the N>1 script path on a 1-GPU box (gloo, shared device)
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} visible GPU(s)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HG_BENCH_BACKEND", "nccl")          # "nccl" == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        # every rank on its own GPU: gather (host, device index, PCI bus id) and refuse duplicates -- two ranks on one device would
        # silently halve the job's throughput (and RCCL refuses them much later, inside the first collective)
        props = torch.cuda.get_device_properties(dev)
        ident = (os.uname().nodename, local_rank, str(getattr(props, "pci_bus_id", "")) + ":" + str(getattr(props, "uuid", "")))
        idents = [None] * world
        dist.all_gather_object(idents, ident)
        if not same_device and len({(h, d) for h, d, _ in idents}) != world:
            raise SystemExit(f"rank {rank}: {world} ranks on {len({(h, d) for h, d, _ in idents})} distinct devices: {idents}")

    if args.workload == "uni8":
        if world > 1:
            raise SystemExit("uni8 runs as replicas (one batch of crystals per GPU, no collective): launch it per GPU")
        run_uni8(args, dev)
        return
    from hamgnn_amd import ops, parallel
    from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
    from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut

    irreps = IRREPS[args.irreps]
    torch.manual_seed(666)
    model = HamGNNConvE3(make_cfg(irreps, args.lite))
    head = HamGNNPlusPlusOut(irreps, irreps, nao_max=args.nao, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True,
                             soc_switch=args.soc, soc_basis="so3", calculate_sparsity=True, zero_point_shift=False)   # the reference's defaults
    # the reference runs the two as Model(representation, output): the output module is the representation's only reader (Model.py:459-465), which lets the
    # backbone's last PairInteractionBlock skip the irreps the head never reads (hamgnn_amd.models.model.Model does this call in its constructor)
    model.declare_consumer(head)
    g = make_graph(args.workload, args.nao, soc=args.soc)
    E_total, N_atoms = g.num_edges, g.num_nodes
    if world > 1:
        g = parallel.shard_graph(g, rank, world)
    g = g.to(dev)
    E_local = g.num_edges
    t_c = time.perf_counter()
    model.compile(dev)                                     # host planner (numpy) + upload of the packed weights and tables: once per model
    head.compile(dev)
    torch.cuda.synchronize()
    compile_s = time.perf_counter() - t_c
    if world > 1:                                          # first collective of the job (communicator set-up, buffer registration): untimed
        warm = torch.zeros(N_atoms, model.node_layout.dim if hasattr(model, "node_layout") else 880, device=dev)
        for _ in range(2):
            torch.distributed.all_reduce(warm)
        torch.cuda.synchronize()
        del warm

    def step():
        with torch.no_grad():
            rep = model(g)
            return head(g, rep)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    import gc
    gc.collect()
    gc.disable()                                           # (no cycle-collector pass inside the timed region; nothing in a step relies on it: tests/test_cpu_end_to_end.py::test_representation_is_released_by_reference_counting)
    for _ in range(args.warmup):
        step()
    barrier()
    ops.PROFILE_EVENTS = []                                  # HIP event pairs around every hg_tp_fused launch (launch stream)
    parallel.PROFILE_EVENTS = [] if world > 1 else None      # ... and around every node all-reduce
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # per-step boundaries on the launch stream (no host sync)
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        out = step()
        marks[k + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    events, ops.PROFILE_EVENTS = ops.PROFILE_EVENTS, None
    ar_events, parallel.PROFILE_EVENTS = parallel.PROFILE_EVENTS, None
    per_rank = None
    if world > 1:
        dt_local = dt
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
        # what every rank spent per step (HIP events on its launch stream): the edge kernel, the node all-reduces (incl. the wait for the
        # slowest rank), everything else (node-level Linears / gates, the read-out of its own edges, launch gaps)
        tp_ms = sum(s.elapsed_time(e) for (s, e, rows, tag) in events if tag == "message_pack") / args.steps
        tp_launches = sum(1 for (s, e, rows, tag) in events if tag == "message_pack") / args.steps
        ar_ms = sum(s.elapsed_time(e) for (s, e, nbytes) in ar_events) / args.steps
        # the node-level launches this rank repeats for ALL atoms (replicated work: does not shrink with the world size unless HG_NODE_SHARD=1):
        # every recorded launch whose row count is the atom count (Linears / row programs of the node chain; gates and lookups are not recorded)
        node_rows_ = {int(N_atoms), -(-int(N_atoms) // world)}
        node_ms = sum(s.elapsed_time(e) for (s, e, rows, tag) in events if tag != "message_pack" and int(rows) in node_rows_) / args.steps
        ideal_tp = REF_TP_MS_PER_MEDGE_LAUNCH[args.irreps] * E_local / 1e6 * tp_launches
        ar_bytes = (ar_events[0][2] if ar_events else 0)
        mine = {"rank": rank, "device": local_rank, "edges": int(E_local), "step_ms": dt_local / args.steps * 1e3, "tp_is_ms": tp_ms,
                "tp_is_launches_per_step": tp_launches, "tp_is_ms_at_the_1gpu_rate": ideal_tp, "tp_is_efficiency_vs_1gpu": (ideal_tp / tp_ms if tp_ms else None),
                "allreduce_ms": ar_ms, "allreduce_calls_per_step": len(ar_events) / args.steps,
                "allreduce_mbytes": ar_bytes / 1e6,
                # bus bandwidth of the collective as RCCL's tests define it: 2 (W - 1) / W of the message per rank over the measured time (incl. the wait
                # for the slowest rank: an upper bound on the wire time); the algorithm is RCCL's choice (NCCL_ALGO / NCCL_PROTO as set in the environment)
                "allreduce_busbw_GBs": (2.0 * (world - 1) / world * ar_bytes * (len(ar_events) / args.steps) / (ar_ms * 1e-3) / 1e9 if ar_ms else None),
                "collective": {"pattern": "reduce-scatter + all-gather of row blocks (HG_NODE_SHARD=1)" if os.environ.get("HG_NODE_SHARD") == "1" else "all-reduce of [N, Dp] per ConvBlock",
                               "NCCL_ALGO": os.environ.get("NCCL_ALGO", "default"), "NCCL_PROTO": os.environ.get("NCCL_PROTO", "default")},
                "node_level_replicated_ms": node_ms, "other_ms": dt_local / args.steps * 1e3 - tp_ms - ar_ms,
                "compile_s": compile_s}
        per_rank = [None] * world
        torch.distributed.all_gather_object(per_rank, mine)
    assert torch.isfinite(out["hamiltonian"]).all()
    sharded_check = None
    if world > 1 and not args.no_accuracy:
        # after the timed region: the SAME model on a bounded sub-crystal of the same generator, sharded over the ranks (the collective of the
        # job: RCCL on a multi-GPU node) against the unsharded forward on rank 0 -- so that a scaling record also says the sharded rows are right
        small = make_graph({"sio2_10k": "sio2_300", "si512": "si64", "mos2_1200": "mos2_48"}.get(args.workload, "sio2_300"), args.nao, soc=args.soc)
        with torch.no_grad():
            ss = parallel.shard_graph(small, rank, world).to(dev)
            Hs = head(ss, model(ss))["hamiltonian"]
        nn_ = small.num_nodes * (2 if args.soc else 1)
        parts_ = [None] * world
        torch.distributed.all_gather_object(parts_, (ss["_hg_edge_ids"].cpu(), Hs[nn_:].float().cpu(), Hs[:nn_].float().cpu()))
        if rank == 0 and not args.soc:
            with torch.no_grad():
                sf = small.to(dev)
                ref = head(sf, model(sf))["hamiltonian"].float().cpu()
            off = torch.zeros(small.num_edges, ref.shape[1])
            for ids, of, on in parts_:
                off[ids] = of
            full = torch.cat([parts_[0][2], off], 0)
            sharded_check = {"rel_err": float((full - ref).abs().max() / ref.abs().max()), "sample": f"{small.num_nodes} atoms / {small.num_edges} edges of the same generator, "
                             f"sharded x{world} vs unsharded on rank 0", "on_site_rows_agree": float(max((p[2] - parts_[0][2]).abs().max() for p in parts_))}
            assert sharded_check["rel_err"] < 1e-5, sharded_check

    # ---- roofline of the dominant kernel: the fused MessagePackBlock launches (hg_tp_is; hg_tp_fused on the fallback path)
    mp = [(s.elapsed_time(e) * 1e-3, rows, tag) for (s, e, rows, tag) in events if tag == "message_pack"]
    all_tp = sum(s.elapsed_time(e) * 1e-3 for (s, e, rows, tag) in events)
    n_launch = max(1, len(mp))
    avg_s = sum(t for t, _, _ in mp) / n_launch
    rows_per_launch = sum(r for _, r, _ in mp) / n_launch
    # the programs of a step in launch order (ConvBlock, PairInteractionBlock per layer).  r5: the FIRST layer's programs drop the super-paths that read
    # structurally zero input irreps (node rows out of the 0e embedding Linear, edge rows out of the 0e x Y^l pair embedding: hamgnn_conv._mark_structural_zeros),
    # so the launches of a step are not all the same program any more: flops are summed program by program
    # r5b: ... and the LAST PairInteractionBlock drops the output irreps the read-out head never reads (HamGNNConvE3.declare_consumer).
    dps, full_of = [], []
    for conv, pair in zip(model.convolutions, model.pair_interactions):
        dps.append(conv.conv_tp._dp_for(E_local, True))        # the program the launch ran (first layer / last pair block: the reduced one)
        full_of.append(conv.conv_tp._dp_for(E_local, False).prog)      # the complete program of the same block (every path of the reference)
        if pair.use_skip_connections or not pair.legacy_edge_update:
            dps.append(pair.conv_tp._dp_for(E_local, True))
            full_of.append(pair.conv_tp._dp_for(E_local, False).prog)
    issued_of = lambda dp: (dp.prog.mfma_per_wave - (dp.prog.mfma_odd_skipped if dp.sched is not None else 0)) * 2048.0 / 16.0
    t_tot = sum(t for t, _, _ in mp)
    useful_tot = sum(dps[k % len(dps)].prog.flops_per_row * r for k, (_, r, _) in enumerate(mp))
    issued_tot = sum(issued_of(dps[k % len(dps)]) * r for k, (_, r, _) in enumerate(mp))
    radial_tot = sum(getattr(dps[k % len(dps)].prog, "mfma_radial", 0) * 2048.0 / 16.0 * r for k, (_, r, _) in enumerate(mp))
    split_on = os.environ.get("HG_S_SPLIT", "1") != "0" and not args.lite
    share = [dp.prog.flops_per_row / fo.flops_per_row for dp, fo in zip(dps, full_of)]           # non-zero share of the reference formulation's flops, per launch of a step
    ref_tot = sum(REF_FLOPS_PER_EDGE_BLOCK[args.irreps] * r for _, r, _ in mp)
    ref_nonzero_tot = sum(REF_FLOPS_PER_EDGE_BLOCK[args.irreps] * share[k % len(dps)] * r for k, (_, r, _) in enumerate(mp))
    # r6: the headline `achieved` / `frac` count, per launch, only the share of the reference formulation's flops that the launched PROGRAM still holds
    # (a first-layer launch that drops the super-paths reading structurally zero irreps is credited with 27 % of 4.55 MFLOP per edge, not with all of it); the
    # figure that credits every launch with the reference op graph's full flops is reported as `frac_vs_reference_op_graph`, clearly named
    flops_tot = useful_tot if args.lite else ref_nonzero_tot
    ach = flops_tot / t_tot / 1e12
    kern = "is" if dps[-1].sched is not None else "seg"
    pmc_bytes = PMC_HBM_BYTES_PER_EDGE_BLOCK[(kern, args.irreps)]
    per_kind = {}
    for k, (t, r, _) in enumerate(mp):
        per_kind.setdefault(k % len(dps), []).append(t)
    full_t = [t for k, (t, r, _) in enumerate(mp) if share[k % len(dps)] > 0.999]
    roofline = {"kernel": ("tp_is_kernel (input-stationary" if kern == "is" else "tp_fused_kernel (segment-stationary") + " MessagePackBlock launches)", "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS, "traffic": pmc_bytes * rows_per_launch,
                "traffic_unit": "bytes per launch (rocprofv3 PMC FETCH_SIZE x 2 + WRITE_SIZE, measured offline on the benchmarked launches of this kernel: profiles/r06_tp_is_pmc.md; "
                                "set-B: r02b_tp_is_hbm_pmc.md, segment-stationary kernel: r01c_tp_fused_hbm_pmc.md), scaled to this launch's edge count", "avg_launch_ms": avg_s * 1e3,
                "launches_timed": len(mp), "edges_per_launch": rows_per_launch,
                # `achieved` / `frac`: reference-formulation flops (4.55 MFLOP per edge and block for set-A) of the paths each launched program HOLDS.
                # `frac_vs_reference_op_graph` credits every launch with the complete block -- incl. the flops the reference spends multiplying the structurally zero
                # input irreps of the first layer and computing output irreps the read-out head never reads, which this build does not issue:
                # a throughput-equivalent, NOT a utilisation.  `frac_full_program_launches`: the launches that run a complete program, alone.
                "frac_vs_reference_op_graph": (ref_tot / t_tot / 1e12 / PEAK_FP32_TFLOPS if not args.lite else None),
                "frac_full_program_launches": (REF_FLOPS_PER_EDGE_BLOCK[args.irreps] * rows_per_launch / (sum(full_t) / max(1, len(full_t))) / 1e12 / PEAK_FP32_TFLOPS
                                               if full_t and not args.lite else None),
                "launch_ms_by_position_in_step": [round(1e3 * sum(v) / len(v), 3) for _, v in sorted(per_kind.items())],
                "nonzero_flop_share_by_position": [round(x, 4) for x in share],
                "executed_useful_tflops": useful_tot / t_tot / 1e12, "issued_mfma_tflops": issued_tot / t_tot / 1e12,
                # r6: the radial scales (this share of the fp32-equivalent MFMAs the programs issue) run as 3 half-precision products of K = 32 on the f16 pipe; the floor of a
                # launch that spends exactly its fp32 MFMAs at the fp32 peak and those products at the f16 peak, over the measured time -- `frac` above prices ALL flops at fp32
                "radial_scale_share_of_issued_mfma": radial_tot / issued_tot if issued_tot else None,
                "frac_of_mixed_pipe_floor": ((issued_tot - radial_tot) / (PEAK_FP32_TFLOPS * 1e12) + 3.0 * radial_tot / (PEAK_F16_TFLOPS * 1e12)) / t_tot * (flops_tot / issued_tot)
                                            if (issued_tot and split_on) else None,
                "hbm_algorithmic_GBs": REF_BYTES_PER_EDGE_BLOCK[args.irreps] * rows_per_launch / avg_s / 1e9,
                "hbm_frac": REF_BYTES_PER_EDGE_BLOCK[args.irreps] * rows_per_launch / avg_s / 1e9 / PEAK_HBM_GBS,
                "hbm_measured_GBs": pmc_bytes * rows_per_launch / avg_s / 1e9,
                "fused_program_launches_share_of_step": all_tp / dt}
    issued, useful = issued_tot / n_launch, useful_tot / n_launch
    if not args.no_mfma_probe:
        # what the fp32 matrix pipe sustains on this device right now (random operands, two waves per SIMD, nothing but MFMAs; after the
        # timed region): the chip clocks to its power budget, so the nominal peak above is not attainable on non-trivial data.  Reported
        # beside `peak`; `frac` stays achieved / nominal peak.
        probe = ops.mfma_probe(dev)
        roofline["mfma_probe_tflops"] = probe
        roofline["issued_over_probe"] = issued / avg_s / 1e12 / probe

    res = {"metric": "edges/sec (equivariant MP forward)", "value": E_total * args.steps / dt, "unit": "edges/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "ms_per_step_median": median_ms,
           "edges_per_s_median_step": E_total / (median_ms * 1e-3), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{args.workload}: {N_atoms} atoms, {E_total} directed edges, irreps set-{args.irreps} (D={model.irreps_node_features.dim}), "
                                  f"sh lmax 5, 3 layers, nao_max {args.nao}, {'SOC (so3 head)' if args.soc else 'no SOC'}{', lite_mode' if args.lite else ''}, backbone+head forward",
                      "parallelism": "single GPU" if world == 1 else f"pair-sharded edges x{world} + RCCL all-reduce of node aggregates"},
           "roofline": roofline, "compile_s": compile_s,
           # arithmetic of the hot kernel (r6): everything in fp32 MFMAs EXCEPT the radial scale S = W3^T h (27.5 % of a block's fp32-equivalent MFMAs), which runs on the
           # half-precision matrix pipe with both operands split into two f16 terms (22 significant bits, fp32 accumulation): 6 x 16 pipe cycles per row tile instead of 16 x 32.
           # `roofline` counts its flops as the fp32 flops they replace and prices them against the fp32 peak.  Accuracy against the fp64 oracle: the `accuracy` field of this line
           # (2.28e-6 with, 2.24e-6 without); HG_S_SPLIT=0 = the all-fp32 form (229.5 ms per step on sio2_10k against 202.9: profiles/r06_tp_is.md)
           "radial_scale_arithmetic": ("fp32 MFMAs (HG_S_SPLIT=0)" if os.environ.get("HG_S_SPLIT", "1") == "0" or args.lite
                                       else "v_mfma_f32_16x16x32_f16 on split operands (hi + 2^-11 lo, 22 bits), fp32 accumulation"),
           # what this build does NOT compute that the reference's op graph does, with identical results (each has a test that compares against the
           # complete programs); same-call A/B on sio2_10k (profiles/r05_shortcuts_ab.md): 3.098 M edges/s without them (HG_STRUCT_ZEROS=0 HG_DEAD_OUT=0: round 4's programs), 3.613 M with the first only, 3.711 M with both
           "exact_shortcuts": {"structurally_zero_input_irreps_of_the_first_layer": os.environ.get("HG_STRUCT_ZEROS", "1") != "0" and not args.lite,
                               "output_irreps_the_declared_head_never_reads_in_the_last_pair_block": [str(model.irreps_node_features[k][0]) + "x" + str(model.irreps_node_features[k][1]) + ("e" if model.irreps_node_features[k][2] == 1 else "o")
                                                                                                    for k in getattr(model.pair_interactions[-1].conv_tp, "_zkw_compiled", {}).get("dead_out", ())],
                               "disable": "HG_STRUCT_ZEROS=0 HG_DEAD_OUT=0"}}
    if per_rank is not None:
        res["per_rank"] = per_rank
        res["sharded_check"] = sharded_check
        res["ranks_seen"] = {"backend": "RCCL (torch.distributed 'nccl')" if backend == "nccl" else backend, "world_size": torch.distributed.get_world_size(),
                             "distinct_devices": len({(h, d) for h, d, _ in idents}), "max_over_mean_edges": max(r["edges"] for r in per_rank) * world / E_total}
    if rank == 0:
        if world == 1 and not args.no_accuracy:
            # accuracy of THIS model (same weights) on a bounded sub-crystal of the same generator: the HIP forward here, the fp64 oracle in a
            # child process on the host cores -- reported next to the throughput, never inside the timed region
            import subprocess
            import tempfile
            try:
                small = make_graph({"sio2_10k": "sio2_60", "si512": "si64", "mos2_1200": "mos2_48", "si2": "si2"}.get(args.workload, "sio2_60"), args.nao, soc=args.soc)
                if small is not None:                          # (SOC / so3 head: [real rows | imaginary rows] of the (2 nao)^2 blocks, hamgnn_output.py:3923-3934)
                    with torch.no_grad():
                        sd = small.to(dev)
                        Hs = head(sd, model(sd))["hamiltonian"].float().cpu()
                    with tempfile.TemporaryDirectory() as td:
                        pth = os.path.join(td, "acc.pt")
                        torch.save({"graph": small, "backbone": {k: v.detach().cpu() for k, v in model.state_dict().items()},
                                    "head": {k: v.detach().cpu() for k, v in head.state_dict().items()}, "H": Hs, "irreps": irreps, "nao": args.nao, "lite": args.lite, "soc": bool(args.soc),
                                    "what": f"sub-crystal of the {args.workload} generator"}, pth)
                        cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--accuracy-from", pth],
                                            capture_output=True, text=True, timeout=240, env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
                    line = [l for l in cp.stdout.splitlines() if l.startswith("ACCURACY ")][-1]
                    res["accuracy"] = json.loads(line[len("ACCURACY "):])
            except Exception as exc:
                res["accuracy"] = {"rel_max": None, "mae": None, "sample": f"failed: {exc!r}"[:200]}
        if world == 1 and not args.lite and not args.no_complete_pass:
            # the like-for-like figure against the reference's op graph: the SAME model with every path of every block issued (HG_STRUCT_ZEROS=0 HG_DEAD_OUT=0:
            # no structural-zero shortcut, no unread-irreps shortcut), recompiled and timed AFTER the timed region -- `value` is never this pass
            prev_env = {k: os.environ.get(k) for k in ("HG_STRUCT_ZEROS", "HG_DEAD_OUT")}
            os.environ["HG_STRUCT_ZEROS"], os.environ["HG_DEAD_OUT"] = "0", "0"
            try:
                model.declare_consumer(head)                     # (HG_DEAD_OUT=0: drops the declaration)
                model.compile(dev)
                nst = max(1, min(args.steps, 5))
                for _ in range(3):                               # (the recompiled model builds its cached chains -- row programs, merged Linears -- on its first forwards)
                    step()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(nst):
                    step()
                torch.cuda.synchronize()
                dt_c = (time.perf_counter() - t1) / nst
                res["value_complete_programs"] = E_total / dt_c
                res["complete_programs"] = {"ms_per_step": dt_c * 1e3, "steps": nst, "what": "every path of the reference's op graph issued (HG_STRUCT_ZEROS=0 HG_DEAD_OUT=0), "
                                            "same model and crystal, timed after the timed region"}
            finally:
                for k, v in prev_env.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
        if world == 1 and not args.no_cpu_baseline:
            import subprocess
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload,
                                     "--irreps", args.irreps, "--nao", str(args.nao)] + (["--lite"] if args.lite else []) + (["--soc"] if args.soc else []), capture_output=True, text=True, timeout=300,
                                    env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
                line = [l for l in cp.stdout.splitlines() if l.startswith("CPU_BASELINE ")][-1]
                res["cpu_baseline"] = json.loads(line[len("CPU_BASELINE "):])
            except Exception as exc:                  # never lose the GPU line because the CPU leg misbehaved
                res["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": 0, "kind": "port", "sample": f"failed: {exc!r}"[:200]}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException as exc:                               # a rank that dies says who it was before the launcher tears the job down
        if isinstance(exc, _LauncherExit):
            raise
        if not isinstance(exc, SystemExit) or exc.code not in (0, None):
            print("BENCH_RANK_FAILURE " + json.dumps({"rank": os.environ.get("RANK", "0"), "local_rank": os.environ.get("LOCAL_RANK", "0"),
                                                      "world_size": os.environ.get("WORLD_SIZE", "1"), "error": repr(exc)[:500]}), file=sys.stderr, flush=True)
        raise
