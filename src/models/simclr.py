"""Same import path and names as the reference's src/models/simclr.py."""
from ss25_hierarchical_multiscale_image_classification_amd.simclr import SimCLRModel, nt_xent_loss  # noqa: F401
