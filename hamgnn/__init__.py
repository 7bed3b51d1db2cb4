"""`hamgnn` import-path shim: the reference's module API for the hot path, resolved to the MI355X modules.

    from hamgnn.models.hamgnn_conv import HamGNNConvE3          # reference: hamgnn/models/hamgnn_conv.py:88
    from hamgnn.models.hamgnn_output import HamGNNPlusPlusOut    # reference: hamgnn/models/hamgnn_output.py:96
    from hamgnn.models.Model import Model                        # reference: hamgnn/models/Model.py:63 (Lightning-free here)
    from hamgnn.main import Model, build_hamgnn_model            # reference: hamgnn/main.py:36, :178-263
    from hamgnn.data.graph_data import NPZGraphDataset, LMDBGraphDataset

Only the names on the hot path exist; the training harness, CLI, config parser and post-processing of the reference are not
provided ("OUT OF SCOPE").  Put this repository root on PYTHONPATH *instead of* the reference package."""
__backend__ = "hamgnn_amd (MI355X / gfx950 HIP kernels)"
