"""drop-in import path of the reference backbone (hamgnn/models/hamgnn_conv.py:88-284) -> hamgnn_amd.models.hamgnn_conv"""
from hamgnn_amd.models.hamgnn_conv import HamGNNConvE3, Representation  # noqa: F401
