"""drop-in import path of the reference's model wrapper (hamgnn/models/Model.py:63-82, :359-376) -> hamgnn_amd.models.model.Model"""
from hamgnn_amd.models.model import Model, load_reference_state_dict, read_checkpoint_state_dict  # noqa: F401
