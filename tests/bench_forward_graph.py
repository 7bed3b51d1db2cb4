"""The forward of a small crystal as a captured HIP graph (serving: the same crystal shape many times, e.g. MD frames with a fixed neighbour
list): python tests/bench_forward_graph.py [--workload si2|si512|mos2_1200] [--reps 50].  A 2-atom cell issues ~60 launches of a few
microseconds each per forward; replayed as one graph the launch gaps go away.  Prints eager and replay milliseconds and the max deviation."""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--workload", default="si2"); ap.add_argument("--reps", type=int, default=50); ap.add_argument("--irreps", default="A")
    a = ap.parse_args()
    from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
    from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
    dev = torch.device("cuda:0")
    irr = B.IRREPS[a.irreps]
    torch.manual_seed(666)
    model = HamGNNConvE3(B.make_cfg(irr))
    head = HamGNNPlusPlusOut(irr, irr, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True, add_H0=True, soc_switch=False, calculate_sparsity=False, zero_point_shift=False)
    g = B.make_graph(a.workload, 19).to(dev)
    model.compile(dev); head.compile(dev)

    def step():
        with torch.no_grad():
            return head(g, model(g))["hamiltonian"]
    for _ in range(3):
        ref = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / a.reps * 1e3
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        step()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = step()
    graph.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        graph.replay()
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t0) / a.reps * 1e3
    print(json.dumps({"workload": a.workload, "edges": int(g.num_edges), "eager_ms": eager, "graph_replay_ms": replay, "max_abs_dev": float((out - ref).abs().max()), "ref_max": float(ref.abs().max())}))


if __name__ == "__main__":
    main()
