from ss25_hierarchical_multiscale_image_classification_amd.patch_dataset import PatchDataset  # noqa: F401
