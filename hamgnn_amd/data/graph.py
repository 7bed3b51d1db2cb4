"""Attribute- and key-addressable graph container standing in for torch_geometric.data.Data/Batch (the reference's input
object: DFT_interfaces/openmx/graph_data_gen.py:357-374; accessed as data.z, data['Hon0'], 'H0_u' in data, data.to(dev))."""
from __future__ import annotations

import torch


class Graph(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def to(self, device, non_blocking=False):
        # the topology cache (hamgnn_amd/topo.py) belongs to THIS object's tensors: never copied to a derived graph
        return Graph({k: (v.to(device, non_blocking=non_blocking) if torch.is_tensor(v) else v) for k, v in self.items() if k != "_hg_topology"})

    def to_dict(self):
        return dict(self)

    @property
    def num_nodes(self):
        return int(self["z"].shape[0])

    @property
    def num_edges(self):
        return int(self["edge_index"].shape[1])


_NODE_KEYS = ("z", "pos", "Hon", "Hon0", "Son", "iHon", "iHon0", "Lon", "Hon_nonsoc", "doping_charge")
_EDGE_KEYS = ("nbr_shift", "cell_shift", "Hoff", "Hoff0", "Soff", "iHoff", "iHoff0", "Loff", "Hoff_nonsoc")


def collate(graphs):
    """Batch graphs the way torch_geometric does for this data: node/edge tensors concatenated, edge_index offset by the
    node count, inv_edge_idx kept graph-local (hamgnn_output.py:2985-2990 adds the per-graph edge offset)."""
    out = Graph()
    n_off = 0
    ei, batch, inv = [], [], []
    for gi, g in enumerate(graphs):
        n = g["z"].shape[0]
        ei.append(g["edge_index"] + n_off)
        batch.append(torch.full((n,), gi, dtype=torch.long))
        inv.append(g["inv_edge_idx"])
        n_off += n
    out["edge_index"] = torch.cat(ei, 1)
    out["batch"] = torch.cat(batch)
    out["inv_edge_idx"] = torch.cat(inv)
    out["node_counts"] = torch.tensor([g["z"].shape[0] for g in graphs])
    out["cell"] = torch.cat([g["cell"].reshape(-1, 3, 3) for g in graphs], 0)
    for k in _NODE_KEYS + _EDGE_KEYS:
        if all(k in g for g in graphs):
            out[k] = torch.cat([g[k] for g in graphs], 0)
    return out
