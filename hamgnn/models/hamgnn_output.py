"""drop-in import path of the reference read-out head (hamgnn/models/hamgnn_output.py:96-123, :2916-4021) -> hamgnn_amd.models.hamgnn_output"""
from hamgnn_amd.models.hamgnn_output import HamGNNPlusPlusOut  # noqa: F401
