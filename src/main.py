"""Drop-in entry point with the reference's path: ``python src/main.py --patch ...``.
Forwards to the package CLI (same flags as the reference's src/main.py:1073-1166 hot-path
subset)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ss25_hierarchical_multiscale_image_classification_amd.main import main  # noqa: E402

if __name__ == "__main__":
    sys.exit(main())
