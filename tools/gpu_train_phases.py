"""Where the GPU time of a training step goes, by REGION of the host code (HIP events around the regions, summed after one synchronisation): the device repack of
the weights, the forward, the head backward, and inside the backbone backward the fused weight-gradient launches, the adjoint launches, the radial-MLP GEMMs and
autograd, finish(), the o3.Linear gradients.  python tools/gpu_train_phases.py [--workload si512] [--steps 5]"""
import argparse
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B  # noqa: E402

EVENTS = []
DEPTH = [0]


def region(name, fn):
    def wrapped(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        DEPTH[0] += 1
        try:
            return fn(*a, **k)
        finally:
            DEPTH[0] -= 1
            e.record()
            EVENTS.append((name, DEPTH[0], s, e))
    return wrapped


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="si512")
    ap.add_argument("--steps", type=int, default=5)
    a = ap.parse_args()
    from hamgnn_amd import backward_mp as BM, nn as hnn, ops, repack as RP, training as T
    from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3
    from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut
    from hamgnn_amd.models.model import Model
    irr = B.IRREPS["A"]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = Model(HamGNNConvE3(B.make_cfg(irr)), HamGNNPlusPlusOut(irr, irr, nao_max=19, ham_type="openmx", ham_only=True, symmetrize=True,
                                                                 add_H0=True, soc_switch=False, calculate_sparsity=False, zero_point_shift=False)).to(dev)
    g = B.make_graph(a.workload, 19).to(dev)
    back, head = model.representation, model.output_module
    # regions (outer to inner)
    back.refresh_weights = region("repack: backbone.refresh_weights", back.refresh_weights)
    hnn.MessagePackBlock.refresh = region("  repack: MessagePackBlock.refresh", hnn.MessagePackBlock.refresh)
    RP.AffinePack.apply = region("    repack: AffinePack.apply", RP.AffinePack.apply)
    RP.mp_sources = region("    repack: mp_sources (L' products)", RP.mp_sources)
    back.pair_embedding.compile = region("  repack: pair_embedding.compile (host planner)", back.pair_embedding.compile)
    hnn.MessagePackBlock.backward_weights = region("bw: MessagePackBlock.backward_weights", hnn.MessagePackBlock.backward_weights)
    hnn.MessagePackBlock.backward_data = region("bw: MessagePackBlock.backward_data (adjoint)", hnn.MessagePackBlock.backward_data)
    ops.tp_wgrad = region("  bw: hg_tp_wgrad launch (+ acc / gs alloc)", ops.tp_wgrad)
    BM.TPWeightGrad.finish = region("  bw: finish()", BM.TPWeightGrad.finish)
    BM.radial_mlp = region("  bw: radial_mlp (autograd recompute)", BM.radial_mlp)
    hnn.HamLayer.backward = region("head: HamLayer.backward", hnn.HamLayer.backward)
    hnn.ResidualBlock.backward = region("bw: ResidualBlock.backward", hnn.ResidualBlock.backward)
    back.pair_embedding.backward = region("bw: pair_embedding.backward", back.pair_embedding.backward)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    fwd = region("forward (no repack)", lambda: (lambda rep: (rep, head(g, rep)))(back(g, save_for_backward=True)))
    hb = region("head backward", lambda rep, gH: head.backward(g, rep, gH))
    bb = region("backbone backward", lambda rep, gn, ge: back.backward(g, rep, gn, ge))
    optim = region("optimiser + weights_changed", lambda: (opt.step(), opt.zero_grad(), T.weights_changed(model)))
    for step in range(a.steps):
        if step == a.steps - 1:
            torch.cuda.synchronize()
            EVENTS.clear()
        with torch.no_grad():
            if getattr(back, "_pending_refresh", False):
                back._pending_refresh = False
                back.refresh_weights()
            rep, out = fwd()
            loss, gH = T._loss_and_grad(out["hamiltonian"], g["hamiltonian"].to(out["hamiltonian"].dtype), "mae")
            g_node, g_edge, gh = hb(rep, gH)
            gb = bb(rep, g_node, g_edge)
        for mod, grads in ((head, gh), (back, gb)):
            params = dict(mod.named_parameters())
            for k, v in grads.items():
                params[k].grad = v.reshape(params[k].shape)
        optim()
    torch.cuda.synchronize()
    tot = collections.OrderedDict()
    for name, depth, s, e in EVENTS:
        t = tot.setdefault(name, [0.0, 0])
        t[0] += s.elapsed_time(e)
        t[1] += 1
    for name, (ms, n) in sorted(tot.items(), key=lambda kv: (len(kv[0]) - len(kv[0].lstrip()), -kv[1][0])):
        print(f"{ms:8.2f} ms  {n:4d} x  {name}")


if __name__ == "__main__":
    main()
