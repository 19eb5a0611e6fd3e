"""Import-path shim: ``from src.models.simclr import ...`` as in the reference."""
from ss25_hierarchical_multiscale_image_classification_amd.simclr import (  # noqa: F401
    SimCLRModel, get_simclr_transform, nt_xent_loss, pretrain_simclr)
