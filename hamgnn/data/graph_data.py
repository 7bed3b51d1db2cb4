"""drop-in import path of the reference's dataset classes (hamgnn/data/graph_data.py:23-185) -> hamgnn_amd.data.graph_data"""
from hamgnn_amd.data.graph_data import LMDBGraphDataset, NPZGraphDataset, load_graph_npz, save_graph_npz  # noqa: F401
