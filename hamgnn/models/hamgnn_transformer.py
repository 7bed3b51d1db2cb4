"""drop-in import path of the reference's attention backbone (hamgnn/models/hamgnn_transformer.py:36-250) -> hamgnn_amd.models.hamgnn_transformer"""
from hamgnn_amd.models.hamgnn_transformer import HamGNNTransformer  # noqa: F401
