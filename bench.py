#!/usr/bin/env python
"""bench.py -- edges/sec of the equivariant message-passing forward (HamGNNConvE3 + HamGNNPlusPlusOut) on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload sio2_10k|si512|mos2_1200|si2] [--irreps A|B]
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A "step" is ONE inference forward of the whole hot path over one synthetic periodic crystal already resident in HBM:
backbone (embedding + num_layers x (ConvBlockE3 + PairInteractionBlock)) + pair read-out head (ham_only, symmetrize,
add_H0, no SOC), random-init weights (seed 666), fp32.  value = directed edges of the crystal * steps / time, whole job.
For N > 1 the undirected pairs are sharded over the ranks (hamgnn_amd/parallel.py) with one RCCL all-reduce of the node
aggregates per layer; the crystal (total work) is fixed => "scaling": "strong".
Prints ONE JSON line (rank 0) incl. `roofline` for the dominant kernel (the fused MessagePackBlock launches: hg_tp_is, or
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
# HBM bytes per edge of one MessagePackBlock launch from the rocprofv3 PMC passes in profiles/r02_tp_is_hbm_pmc.md / r01c_tp_fused_hbm_pmc.md
# (separate --pmc FETCH_SIZE / WRITE_SIZE runs on tests/bench_tp.py, 131072 edges; FETCH_SIZE x 2: gfx950 correction for
# 16-B/lane reads, MI355X_MICROARCH.md "HBM"): measured offline for this kernel build, scaled to the launch's edge count.
# kernel "is" = input-stationary tp_is_kernel (profiles/r02_tp_is_hbm_pmc.md), "seg" = segment-stationary tp_fused_kernel
PMC_HBM_BYTES_PER_EDGE_BLOCK = {("is", "A"): 34.6e3, ("is", "B"): 14.2e3, ("seg", "A"): 173.2e3, ("seg", "B"): 59.9e3}
PEAK_FP32_TFLOPS = 157.3           # MI355X_MICROARCH.md: fp32 vector == fp32-input MFMA peak
PEAK_HBM_GBS = 8000.0


def make_cfg(irreps):
    return dict(num_types=96, irreps_edge_sh=SH, edge_sh_normalization="component", edge_sh_normalize=True,
                build_internal_graph=False, cutoff=26.0, rbf_func="bessel", num_radial=64, num_layers=3,
                irreps_node_features=irreps, use_kan=False, radial_MLP=[64, 64], correlation=2, num_hidden_features=16,
                radius_type="openmx", use_corr_prod=False, legacy_edge_update=False, lite_mode=False)


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
    elif workload.startswith("sio2_"):
        g = S.amorphous_sio2(int(workload.split("_")[1]), seed=1)
    else:
        raise SystemExit(f"unknown workload {workload}")
    return S.add_random_targets(g, nao, seed=0, soc=soc)


def cpu_baseline(workload, irreps_key, nao, budget_s=15.0):
    """Oracle (unfused torch port of the reference op graph) on the host cores, bounded sample of the same workload."""
    from oracle import hamgnn_ref as R
    from hamgnn_amd.data import synthetic as S
    ncpu = os.cpu_count() or 1
    cand = [int(os.environ["HG_CPU_THREADS"])] if "HG_CPU_THREADS" in os.environ else sorted({min(ncpu, c) for c in (8, 16, 32, 64)})
    cores = cand[0]
    torch.set_num_threads(cores)
    irreps = IRREPS[irreps_key]
    torch.manual_seed(666)
    model = R.HamGNNConvE3(make_cfg(irreps)).float()
    head = R.HamGNNPlusPlusOut(irreps, irreps, nao_max=nao, ham_type="openmx", symmetrize=True, add_H0=True).float()

    def run(n_atoms):
        if workload.startswith("sio2"):
            g = S.amorphous_sio2(n_atoms, seed=1)
        elif workload.startswith("mos2"):
            k = max(1, int(round((n_atoms / 3) ** 0.5)))
            g = S.mos2_monolayer(k, k)
        else:
            k = max(1, int(round((n_atoms / 8) ** (1 / 3))))
            g = S.si_diamond(k, k, k, jitter=0.05, seed=0)
        S.add_random_targets(g, nao, seed=0)
        t0 = time.perf_counter()
        with torch.no_grad():
            head(g, model(g))
        return g.num_edges, time.perf_counter() - t0, g.num_nodes

    e1, t1, n1 = run(12)                       # warm-up (first-touch, thread pool)
    best = None
    for c in cand:                             # these small ops do not scale with threads: pick the fastest count and report it
        torch.set_num_threads(c)
        r = run(12)                            # calibration: ~1e3 edges
        if best is None or r[1] < best[1][1]:
            best = (c, r)
    cores, (e1, t1, n1) = best
    torch.set_num_threads(cores)
    e2, t2, n2 = e1, t1, n1
    if t1 < budget_s / 3:                      # bounded sample: aim at ~budget_s seconds of CPU work
        n_atoms = max(12, min(int(n1 * budget_s / t1), 2000))
        e2, t2, n2 = run(n_atoms)
    return {"value": e2 / t2, "unit": "edges/s", "cores": cores, "kind": "port",
            "sample": f"{workload}-like crystal, {n2} atoms / {e2} directed edges, 1 forward in {t2:.1f}s, fp32, torch {torch.get_num_threads()} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="sio2_10k")
    ap.add_argument("--irreps", default="A", choices=["A", "B"])
    ap.add_argument("--nao", type=int, default=19)
    ap.add_argument("--soc", action="store_true", help="SOC / so3 read-out (BASELINE config #3: MoS2 with spin-orbit coupling); the CPU baseline leg stays non-SOC")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.cpu_baseline_only:                       # child process of the N=1 run (hard wall-clock bound in the parent)
        print("CPU_BASELINE " + json.dumps(cpu_baseline(args.workload, args.irreps, args.nao)), flush=True)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if os.environ.get("HG_BENCH_SAME_DEVICE") == "1":      # test hook: validate the N>1 script path on a 1-GPU box (gloo, shared device)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("HG_BENCH_BACKEND", "nccl")          # "nccl" == RCCL on ROCm
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from hamgnn_amd import ops, parallel
    from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
    from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut

    irreps = IRREPS[args.irreps]
    torch.manual_seed(666)
    model = HamGNNConvE3(make_cfg(irreps))
    head = HamGNNPlusPlusOut(irreps, irreps, nao_max=args.nao, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True,
                             soc_switch=args.soc, soc_basis="so3", calculate_sparsity=True, zero_point_shift=False)   # the reference's defaults
    g = make_graph(args.workload, args.nao, soc=args.soc)
    E_total, N_atoms = g.num_edges, g.num_nodes
    if world > 1:
        g = parallel.shard_graph(g, rank, world)
    g = g.to(dev)
    model.compile(dev)
    head.compile(dev)

    def step():
        with torch.no_grad():
            rep = model(g)
            return head(g, rep)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ops.PROFILE_EVENTS = []                                  # HIP event pairs around every hg_tp_fused launch (launch stream)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]      # per-step boundaries on the launch stream (no host sync)
    t0 = time.perf_counter()
    marks[0].record()
    for k in range(args.steps):
        out = step()
        marks[k + 1].record()
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    events, ops.PROFILE_EVENTS = ops.PROFILE_EVENTS, None
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    assert torch.isfinite(out["hamiltonian"]).all()

    # ---- roofline of the dominant kernel: the fused MessagePackBlock launches (hg_tp_is; hg_tp_fused on the fallback path)
    mp = [(s.elapsed_time(e) * 1e-3, rows, tag) for (s, e, rows, tag) in events if tag == "message_pack"]
    all_tp = sum(s.elapsed_time(e) * 1e-3 for (s, e, rows, tag) in events)
    n_launch = max(1, len(mp))
    avg_s = sum(t for t, _, _ in mp) / n_launch
    rows_per_launch = sum(r for _, r, _ in mp) / n_launch
    flops_launch = REF_FLOPS_PER_EDGE_BLOCK[args.irreps] * rows_per_launch
    useful = model.convolutions[0].conv_tp._dp.prog.flops_per_row * rows_per_launch
    dp0 = model.convolutions[0].conv_tp._dp
    issued = (dp0.prog.mfma_per_wave - (dp0.prog.mfma_odd_skipped if dp0.sched is not None else 0)) * 2048.0 / 16.0 * rows_per_launch
    ach = flops_launch / avg_s / 1e12
    kern = "is" if model.convolutions[0].conv_tp._dp.sched is not None else "seg"
    pmc_bytes = PMC_HBM_BYTES_PER_EDGE_BLOCK[(kern, args.irreps)]
    roofline = {"kernel": ("tp_is_kernel (input-stationary" if kern == "is" else "tp_fused_kernel (segment-stationary") + " MessagePackBlock launches)", "bound": "mfma", "achieved": ach, "peak": PEAK_FP32_TFLOPS,
                "unit": "TFLOP/s", "frac": ach / PEAK_FP32_TFLOPS, "traffic": pmc_bytes * rows_per_launch,
                "traffic_unit": "bytes per launch (PMC, measured offline: profiles/r02b_tp_is_hbm_pmc.md, r01c_tp_fused_hbm_pmc.md)", "avg_launch_ms": avg_s * 1e3,
                "launches_timed": len(mp), "edges_per_launch": rows_per_launch,
                "executed_useful_tflops": useful / avg_s / 1e12, "issued_mfma_tflops": issued / avg_s / 1e12,
                "hbm_algorithmic_GBs": REF_BYTES_PER_EDGE_BLOCK[args.irreps] * rows_per_launch / avg_s / 1e9,
                "hbm_frac": REF_BYTES_PER_EDGE_BLOCK[args.irreps] * rows_per_launch / avg_s / 1e9 / PEAK_HBM_GBS,
                "hbm_measured_GBs": pmc_bytes * rows_per_launch / avg_s / 1e9,
                "fused_program_launches_share_of_step": all_tp / dt}

    res = {"metric": "edges/sec (equivariant MP forward)", "value": E_total * args.steps / dt, "unit": "edges/s", "n_gpus": world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "ms_per_step_median": median_ms,
           "edges_per_s_median_step": E_total / (median_ms * 1e-3), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": f"{args.workload}: {N_atoms} atoms, {E_total} directed edges, irreps set-{args.irreps} (D={model.irreps_node_features.dim}), "
                                  f"sh lmax 5, 3 layers, nao_max {args.nao}, {'SOC (so3 head)' if args.soc else 'no SOC'}, backbone+head forward",
                      "parallelism": "single GPU" if world == 1 else f"pair-sharded edges x{world} + RCCL all-reduce of node aggregates"},
           "roofline": roofline}
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            import subprocess
            try:
                cp = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--workload", args.workload,
                                     "--irreps", args.irreps, "--nao", str(args.nao)], capture_output=True, text=True, timeout=150,
                                    env=dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
                line = [l for l in cp.stdout.splitlines() if l.startswith("CPU_BASELINE ")][-1]
                res["cpu_baseline"] = json.loads(line[len("CPU_BASELINE "):])
                res["speedup_vs_cpu"] = res["value"] / res["cpu_baseline"]["value"]
            except Exception as exc:                  # never lose the GPU line because the CPU leg misbehaved
                res["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": 0, "kind": "port", "sample": f"failed: {exc!r}"[:200]}
        print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
