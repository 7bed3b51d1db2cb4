"""`graph_data.npz` reader / writer for the hot path (reference container: hamgnn/data/graph_data.py:96-185 NPZGraphDataset;
produced by DFT_interfaces/*/graph_data_gen.py:357-374 as ``np.savez(path, graph={idx: Data(...)})``).

The reference stores pickled torch_geometric ``Data`` objects.  This reader works without torch_geometric: an allow-list
unpickler maps ``torch_geometric.data.*`` classes onto the dependency-free ``Graph`` container (attribute + key access,
same field names), so files written by the reference tool-chain and by ``save_graph_npz`` load the same way.  Dict-of-
arrays graphs (the reference's second accepted form, graph_data.py:141-156) are converted as well.  LMDB
stores (graph_data.py:23-93, keys ``num_graphs`` / ``graph_{i}``) are read with the dependency-free parser in ``lmdb_lite``."""
from __future__ import annotations

import io
import pickle
import zipfile
from typing import Dict, List, Sequence

import numpy as np
import torch

from .graph import Graph, collate


class _PyGStub:
    """stand-in for torch_geometric.data.Data / storage classes while unpickling: keeps the attribute dict."""

    def __init__(self, *a, **k):
        self.__dict__.update(k)

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self.__dict__["_state"] = state


def _to_graph(obj) -> Graph:
    if isinstance(obj, Graph):
        return obj
    if isinstance(obj, dict):
        items = obj
    else:
        d = dict(getattr(obj, "__dict__", {}))
        store = d.get("_store", None)                       # PyG >= 2: Data.__dict__['_store'] is a GlobalStorage with a _mapping
        if store is not None:
            sd = getattr(store, "__dict__", {})
            items = dict(sd.get("_mapping", sd))
        else:
            items = {k: v for k, v in d.items() if not k.startswith("_")}
    g = Graph()
    for k, v in items.items():
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        g[k] = v
    for req in ("z", "pos", "edge_index"):
        if req not in g:
            raise ValueError(f"graph record lacks the field {req!r}")
    return g


# Everything a pickled graph record legitimately refers to: tensors / storages, numpy arrays and scalars, plain containers, and
# the torch_geometric container classes (mapped onto _PyGStub).  Any other global is refused: np.load(allow_pickle=True), which the
# reference uses (hamgnn/data/graph_data.py:110-119), would execute it.
_ALLOWED = {
    ("collections", "OrderedDict"), ("collections", "defaultdict"), ("builtins", "dict"), ("builtins", "list"), ("builtins", "tuple"),
    ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"),
    ("builtins", "bytes"), ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"), ("builtins", "bytearray"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch", "Size"), ("torch", "device"), ("torch", "dtype"), ("torch.storage", "_load_from_bytes"),
    ("torch.serialization", "_get_layout"), ("torch", "Tensor"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"), ("numpy.core.numeric", "_frombuffer"),
    ("numpy._core.numeric", "_frombuffer"), ("_codecs", "encode"), ("copyreg", "_reconstructor"), ("builtins", "object"),
}
_ALLOWED_TORCH_NAMES = {n for n in dir(torch) if n.endswith("Storage")} | {str(d).split(".")[-1] for d in (
    torch.float16, torch.float32, torch.float64, torch.bfloat16, torch.int8, torch.uint8, torch.int16, torch.int32, torch.int64, torch.bool,
    torch.complex64, torch.complex128)}


class _Unpickler(pickle.Unpickler):
    """allow-list unpickler (see _ALLOWED); torch_geometric classes become _PyGStub"""

    def find_class(self, module, name):
        if module.startswith("torch_geometric"):
            return _PyGStub
        if module.startswith("hamgnn_amd.data") and name == "Graph":
            return Graph
        if (module, name) == ("torch.storage", "_load_from_bytes"):
            # the stock helper is torch.load(BytesIO(b), weights_only=False): a nested payload would run an arbitrary __reduce__
            return lambda b: torch.load(io.BytesIO(b), weights_only=True)
        if (module, name) in _ALLOWED or (module == "torch" and name in _ALLOWED_TORCH_NAMES):
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"graph record refers to {module}.{name}, which is not on the loader's allow-list")


def load_graph_npz(path: str) -> List[Graph]:
    """All graphs of a ``graph_data.npz`` as Graph objects (order of the stored dict)."""
    with zipfile.ZipFile(path) as zf:
        names = zf.namelist()
        if "graph.npy" in names:
            with zf.open("graph.npy") as f:
                buf = io.BytesIO(f.read())
            major, minor = np.lib.format.read_magic(buf)
            (np.lib.format.read_array_header_1_0 if major == 1 else np.lib.format.read_array_header_2_0)(buf)
            obj = _Unpickler(buf).load()                    # object array of shape () holding the dict
            payload = obj.item() if isinstance(obj, np.ndarray) else obj
            records = list(payload.values()) if isinstance(payload, dict) else list(payload)
            return [_to_graph(r) for r in records]
    with np.load(path, allow_pickle=False) as data:         # architecture 2: every key is one graph stored as arrays is not poolable
        raise ValueError(f"{path}: no 'graph' entry (found {list(data.keys())[:5]})")


def save_graph_npz(graphs: Sequence[Graph] | Dict[int, Graph], path: str):
    """Write graphs in the reference container format: key 'graph' -> dict {index: graph object}."""
    d = dict(graphs) if isinstance(graphs, dict) else {i: g for i, g in enumerate(graphs)}
    np.savez(path, graph=np.array(d, dtype=object))


class NPZGraphDataset:
    """Minimal mirror of the reference dataset class (len / getitem / optional transform)."""

    def __init__(self, npz_path: str, indices=None, transform=None, preload: int = 0):
        self.data_list = load_graph_npz(npz_path)
        self.indices = list(indices) if indices is not None else list(range(len(self.data_list)))
        self.transform = transform

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, idx):
        if isinstance(idx, list):
            return [self[i] for i in idx]
        g = self.data_list[self.indices[idx]]
        return self.transform(g) if self.transform is not None else g


class LMDBGraphDataset:
    """Mirror of the reference's LMDB dataset (hamgnn/data/graph_data.py:23-93): keys ``num_graphs`` and ``graph_{i}``, values =
    pickled graph records; same constructor / len / getitem / transform / preload behaviour.  Reads the store with the dependency-free
    parser in ``lmdb_lite`` (the ``lmdb`` module is not needed) and unpickles through the allow-list ``_Unpickler`` above, so records
    written by the reference (torch_geometric ``Data``) come back as ``Graph`` objects."""

    def __init__(self, lmdb_path: str, indices=None, transform=None, preload: int = 0):
        self.lmdb_path, self.transform, self.preload = lmdb_path, transform, preload
        self._reader_obj = None                                # opened lazily (and again in every DataLoader worker: the mmap does not pickle)
        n = self._reader.get(b"num_graphs")
        if n is None:
            raise ValueError(f"{lmdb_path}: key 'num_graphs' is missing")
        self.total_length = int(n.decode())
        self.indices = list(indices) if indices is not None else list(range(self.total_length))
        self.preloaded_data = {i: self._load(i) for i in self.indices[:max(0, preload)]}

    @property
    def _reader(self):
        if self._reader_obj is None:
            from .lmdb_lite import LMDBReader
            self._reader_obj = LMDBReader(self.lmdb_path)
        return self._reader_obj

    def __getstate__(self):                                    # spawn-mode DataLoader workers: reopen the store on first use, as the reference does
        d = dict(self.__dict__)
        d["_reader_obj"] = None
        return d

    def _load(self, real_idx: int) -> Graph:
        raw = self._reader.get(f"graph_{real_idx}".encode())
        if raw is None:
            raise IndexError(f"Index {real_idx} out of bounds for LMDB dataset")
        return _to_graph(_Unpickler(io.BytesIO(raw)).load())

    def __len__(self):
        return len(self.indices)

    def __getitem__(self, idx):
        if isinstance(idx, list):
            return [self[i] for i in idx]
        real = self.indices[idx]
        g = self.preloaded_data.get(real)
        if g is None:
            g = self._load(real)
        return self.transform(g) if self.transform is not None else g

    def close(self):
        if self._reader_obj is not None:
            self._reader_obj.close()
            self._reader_obj = None


def npz_to_lmdb(npz_path: str, lmdb_path: str) -> str:
    """tools/npz_to_lmdb.py:27-109 without the lmdb module: ``num_graphs`` + ``graph_{i}`` (i = rank of the sorted npz key)."""
    from .lmdb_lite import write_lmdb
    graphs = load_graph_npz(npz_path)
    items = {b"num_graphs": str(len(graphs)).encode()}
    for i, g in enumerate(graphs):
        items[f"graph_{i}".encode()] = pickle.dumps(g)
    return write_lmdb(lmdb_path, items)


def batches(graphs: Sequence[Graph], batch_size: int = 1):
    for i in range(0, len(graphs), batch_size):
        yield collate(list(graphs[i:i + batch_size]))
