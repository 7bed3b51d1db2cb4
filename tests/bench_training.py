"""Time one full training step (forward with saved layer inputs + loss + backward of head and backbone + host repack) of the non-SOC
model at the reference's default irreps on a synthetic crystal:  python tests/bench_training.py [--workload si64|si512|sio2_300] [--steps 3]
Prints per-phase milliseconds (HIP events are not needed: the phases are hundreds of milliseconds; torch.cuda.synchronize brackets)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="si64")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--irreps", default="A")
    ap.add_argument("--profile", action="store_true", help="cProfile of the last step (host time: the step is host-bound at small sizes)")
    a = ap.parse_args()
    from hamgnn_amd.data import synthetic as S
    from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
    from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
    from hamgnn_amd.models.model import Model
    from hamgnn_amd import training as T
    irr = B.IRREPS[a.irreps]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = Model(HamGNNConvE3(B.make_cfg(irr)), HamGNNPlusPlusOut(irr, irr, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True,
                                                                 add_H0=True, soc_switch=False, calculate_sparsity=False, zero_point_shift=False)).to(dev)
    if a.workload == "si64":
        g = S.add_random_targets(S.si_diamond(2, 2, 2, jitter=0.05, seed=0), 19, seed=0)
    else:
        g = B.make_graph(a.workload, 19)
    g = g.to(dev)
    sync = torch.cuda.synchronize
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    for step in range(a.steps):
        if a.profile and step == a.steps - 1:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
        sync(); t0 = time.time()
        with torch.no_grad():
            rep = model.representation(g, save_for_backward=True)
            out = model.output_module(g, rep)
            sync(); t1 = time.time()
            loss, gH = T._loss_and_grad(out["hamiltonian"], g["hamiltonian"].to(out["hamiltonian"].dtype), "mae")
            g_node, g_edge, gh = model.output_module.backward(g, rep, gH)
            sync(); t2 = time.time()
            gb = model.representation.backward(g, rep, g_node, g_edge)
            sync(); t3 = time.time()
        for mod, grads in ((model.output_module, gh), (model.representation, gb)):
            params = dict(mod.named_parameters())
            for k, v in grads.items():
                params[k].grad = v.reshape(params[k].shape)
        opt.step(); opt.zero_grad()
        T.weights_changed(model)
        sync(); t4 = time.time()
        print(f"step {step}: N {g.num_nodes} E {g.num_edges} loss {float(loss):.5f} | forward (incl. repack) {1e3 * (t1 - t0):.0f} ms, head backward "
              f"{1e3 * (t2 - t1):.0f} ms, backbone backward {1e3 * (t3 - t2):.0f} ms, optimiser {1e3 * (t4 - t3):.0f} ms", flush=True)
        if a.profile and step == a.steps - 1:
            pr.disable()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(45)


if __name__ == "__main__":
    main()
