"""Same import path and class names as the reference's src/models/resnet.py."""
from ss25_hierarchical_multiscale_image_classification_amd.resnet import (  # noqa: F401
    ResNet18Classifier,
    ResNet18ClassifierSIMCLR,
    ResNet18FeatureExtractor,
    UnifiedResNet,
)
