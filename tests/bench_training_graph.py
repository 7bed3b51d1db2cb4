"""Experiment: the whole training step (forward + loss + backward of head and backbone + device-side refresh of the packed weights) as ONE
captured HIP graph, replayed per optimiser step:  python tests/bench_training_graph.py [--workload si512] [--steps 5]
Prints eager vs replayed milliseconds per step and the largest gradient difference between the two."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="si512")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--irreps", default="A")
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
    g = (S.add_random_targets(S.si_diamond(2, 2, 2, jitter=0.05, seed=0), 19, seed=0) if a.workload == "si64" else B.make_graph(a.workload, 19)).to(dev)
    params = list(model.parameters())

    def step():
        for p in params:
            p.grad = None
        return T.training_step(model, g, metric="mae")["loss"]

    sync = torch.cuda.synchronize
    for _ in range(3):
        step()
    sync(); t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    sync(); eager = (time.perf_counter() - t0) / a.steps * 1e3
    ref = [p.grad.clone() for p in params]
    cap = T.CapturedTrainingStep(model, g, metric="mae")
    sync(); t0 = time.perf_counter()
    for _ in range(a.steps):
        loss = cap()
    sync(); replay = (time.perf_counter() - t0) / a.steps * 1e3
    diff = max(float((p.grad - r).abs().max() / (r.abs().max() + 1e-30)) for p, r in zip(params, ref))
    print("TRAIN_GRAPH " + json.dumps({"workload": a.workload, "eager_ms": eager, "replay_ms": replay, "grad_rel_diff": diff, "loss": float(loss)}))


if __name__ == "__main__":
    main()
