"""Package name used by the reference's example.ipynb (cell 0): NODE_WAN_model.training / NODE_WAN_model.dataset."""
