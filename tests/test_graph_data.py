"""graph_data.npz container round trip, the torch_geometric-free unpickling shim, batching."""
import pickle
import sys
import types

import numpy as np
import pytest
import torch

from hamgnn_amd.data import Graph, collate
from hamgnn_amd.data import graph_data as GD
from hamgnn_amd.data import synthetic as S


def test_npz_roundtrip(tmp_path):
    g1 = S.add_random_targets(S.si_diamond(primitive=True), 19)
    g2 = S.add_random_targets(S.random_cell(5, [14, 8], seed=1, density=0.004), 19)
    p = str(tmp_path / "graph_data.npz")
    GD.save_graph_npz([g1, g2], p)
    ds = GD.NPZGraphDataset(p)
    assert len(ds) == 2
    for a, b in zip(ds.data_list, (g1, g2)):
        for k in ("z", "pos", "edge_index", "nbr_shift", "inv_edge_idx", "Hon0", "Hoff0"):
            assert torch.equal(a[k], b[k])
        assert a.z is a["z"]


def test_reads_pyg_style_pickles_without_torch_geometric(tmp_path):
    """records pickled as torch_geometric.data.data.Data (attribute dict / _store mapping) load as Graph."""
    mod = types.ModuleType("torch_geometric"); sub = types.ModuleType("torch_geometric.data"); sub2 = types.ModuleType("torch_geometric.data.data")
    Data = type("Data", (), {"__init__": lambda self, **k: self.__dict__.update(k)})   # what the producer side pickles
    Data.__module__, Data.__qualname__ = "torch_geometric.data.data", "Data"
    sub2.Data = Data
    sys.modules.update({"torch_geometric": mod, "torch_geometric.data": sub, "torch_geometric.data.data": sub2})
    try:
        g = S.si_diamond(primitive=True)
        rec = Data(**{k: v for k, v in g.items()})
        p = str(tmp_path / "graph_data.npz")
        np.savez(p, graph=np.array({0: rec}, dtype=object))
    finally:
        for k in ("torch_geometric", "torch_geometric.data", "torch_geometric.data.data"):
            sys.modules.pop(k, None)
    out = GD.load_graph_npz(p)
    assert isinstance(out[0], Graph) and torch.equal(out[0].edge_index, g.edge_index) and torch.equal(out[0]["pos"], g.pos)


def test_collate_offsets():
    g1, g2 = S.si_diamond(primitive=True), S.random_cell(4, [14, 8], seed=2, density=0.004)
    b = collate([g1, g2])
    assert b.z.shape[0] == 6 and b.edge_index.shape[1] == g1.num_edges + g2.num_edges
    assert torch.equal(b.edge_index[:, g1.num_edges:], g2.edge_index + 2)
    assert torch.equal(b.inv_edge_idx[g1.num_edges:], g2.inv_edge_idx)          # graph-local, as the reference expects
    assert b.node_counts.tolist() == [2, 4] and b.batch.tolist() == [0, 0, 1, 1, 1, 1]


def test_lmdb_store_roundtrip_without_the_lmdb_module(tmp_path):
    """npz -> LMDB (bulk writer) -> LMDBGraphDataset (pure-python B+tree reader): inline values, overflow runs (graphs are far larger than
    a page), several leaf pages and a branch level (200 extra keys), the reference's key scheme and dataset behaviour."""
    from hamgnn_amd.data import lmdb_lite as L
    gs = [S.add_random_targets(S.si_diamond(primitive=True), 19), S.add_random_targets(S.random_cell(5, [14, 8], seed=1, density=0.004), 19)]
    npz = str(tmp_path / "graph_data.npz")
    GD.save_graph_npz(gs, npz)
    db = GD.npz_to_lmdb(npz, str(tmp_path / "store"))
    ds = GD.LMDBGraphDataset(db, preload=1)
    assert len(ds) == 2 and ds.total_length == 2
    for a, b in zip([ds[0], ds[1]], gs):
        for k in ("z", "pos", "edge_index", "nbr_shift", "inv_edge_idx", "Hon0", "Hoff0"):
            assert torch.equal(a[k], b[k])
    assert len(GD.LMDBGraphDataset(db, indices=[1])) == 1
    with pytest.raises(IndexError):
        ds._load(7)
    ds.close()
    # a deeper tree: many small keys + a few big values
    rng = np.random.default_rng(0)
    items = {f"k{i:05d}".encode(): bytes(rng.integers(0, 256, size=int(rng.integers(1, 300)), dtype=np.uint8)) for i in range(12000)}
    items[b"big"] = bytes(rng.integers(0, 256, size=70000, dtype=np.uint8))
    items[b""] = b"empty key"
    L.write_lmdb(str(tmp_path / "deep"), items)
    with L.LMDBReader(str(tmp_path / "deep")) as r:
        assert len(r) == len(items) and r.depth >= 3
        for k, v in items.items():
            assert r.get(k) == v, k
        assert r.get(b"k99999") is None and r.get(b"a") is None and r.get(b"zzz") is None
        assert [k for k, _ in r.items()] == sorted(items)


class _Boom:
    """a record whose unpickling would run an arbitrary callable"""
    def __reduce__(self):
        return (eval, ("__import__('os').environ.__setitem__('HG_PWNED', '1')",))


def test_nested_torch_payload_cannot_run_code(tmp_path):
    """`torch.storage._load_from_bytes` is on the allow-list (tensors pickled by old torch versions go through it), but the stock helper is
    torch.load(weights_only=False): a record that reduces to _load_from_bytes(<torch.save of an object with __reduce__>) must be refused,
    while a genuine tensor payload still loads"""
    import io
    import os
    buf = io.BytesIO()
    torch.save(_Boom(), buf)
    evil = type("Evil", (), {"__reduce__": lambda self: (torch.storage._load_from_bytes, (buf.getvalue(),))})()
    raw = pickle.dumps({"z": torch.tensor([1]), "pos": torch.zeros(1, 3), "edge_index": torch.zeros(2, 0, dtype=torch.long), "x": evil})
    os.environ.pop("HG_PWNED", None)
    with pytest.raises(Exception):
        GD._Unpickler(io.BytesIO(raw)).load()
    assert "HG_PWNED" not in os.environ
    good = io.BytesIO()
    torch.save(torch.arange(4.0), good)
    ok = type("Ok", (), {"__reduce__": lambda self: (torch.storage._load_from_bytes, (good.getvalue(),))})()
    out = GD._Unpickler(io.BytesIO(pickle.dumps({"t": ok}))).load()
    assert torch.equal(out["t"], torch.arange(4.0))


def test_lmdb_dataset_pickles_for_dataloader_workers(tmp_path):
    """spawn-mode DataLoader workers receive a pickled dataset: the store is reopened lazily in the worker (the reference opens its env
    lazily too, hamgnn/data/graph_data.py:38-52)"""
    gs = [S.add_random_targets(S.si_diamond(primitive=True), 19)]
    npz = str(tmp_path / "graph_data.npz")
    GD.save_graph_npz(gs, npz)
    db = GD.npz_to_lmdb(npz, str(tmp_path / "store"))
    ds = GD.LMDBGraphDataset(db)
    _ = ds[0]
    clone = pickle.loads(pickle.dumps(ds))
    assert clone._reader_obj is None and torch.equal(clone[0]["pos"], gs[0].pos)
    ds.close()
