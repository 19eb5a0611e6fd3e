"""Reference module path ``models.mil_classifier`` (src/models/mil_classifier.py) -> the MI355X build."""
from ss25_hierarchical_multiscale_image_classification_amd.mil import MILAttentionPooling, MILClassifier  # noqa: F401
