"""``src.datasets`` as a real package (the reference's is a namespace package that an
installed ``datasets`` distribution shadows)."""
from ss25_hierarchical_multiscale_image_classification_amd.patch_dataset import PatchDataset  # noqa: F401
from ss25_hierarchical_multiscale_image_classification_amd.simclr_dataset import SimCLRDataset  # noqa: F401
