from .hamgnn_conv import HamGNNConvE3  # noqa: F401
from .hamgnn_output import HamGNNPlusPlusOut  # noqa: F401
from .Model import Model  # noqa: F401
