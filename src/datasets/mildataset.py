"""Reference module path ``datasets.mildataset`` (src/datasets/mildataset.py) -> the MI355X build."""
from ss25_hierarchical_multiscale_image_classification_amd.mil import WSIMILDDataset  # noqa: F401
