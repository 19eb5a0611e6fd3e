"""CPU oracle for the HiPAC patch-inference hot path.  TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement of the reference's algorithm for the path
named in BASELINE.json (sliding-window extractor -> Resize/ToTensor/Normalize
-> ResNet18 forward -> gather; NT-Xent for the SimCLR step).  It exists so the
HIP path can be checked; it is never the thing shipped or measured.

Who may import it: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``.  Nothing under
``ss25_hierarchical_multiscale_image_classification_amd/`` imports it, and the
product path raises when the HIP library is missing instead of falling back.

Pinning status (see DESIGN.md "Oracle"):
  * resize    - pinned on Pillow itself (the third-party code the reference
                calls through torchvision.transforms.Resize); the numpy
                restatement in ``transform_ref`` is checked bit-for-bit
                against ``PIL.Image.resize`` in tests/test_oracle_transform.py.
  * extractor - grid arithmetic pinned on the two known answers the
                reference's notebooks record (6642 windows / 1.39 % for
                97792x221184 at P=1792; 64x56 grid for 14336x12544 at L3... see
                tests/test_oracle_extractor.py).  Per-window pixels: the
                reference's goldens need the real CAMELYON16 slides (absent),
                so beyond the grid the extractor is PARITY UNPINNED.
  * ResNet18  - the graph lives in torchvision (==0.16.0+cu121 in the
                reference's pip freeze), which is absent from this image and
                from /root/reference, and the reference holds no fixture for
                it: PARITY UNPINNED by reference fixtures.  Pinned only on the
                published architecture's known answers (11,176,512 conv/bn
                parameters + 1,026 fc; output shapes) and on agreement
                between an nn.Module restatement and the functional one.
  * NT-Xent   - restated from src/models/simclr.py:31-54 (pure torch); that
                file cannot be imported whole (top-level torchvision import),
                but its nt_xent_loss definition compiles on its own:
                tests/golden/make_golden_ntxent.py ran the reference's OWN
                function in the build container and stored inputs, values and
                gradients (tests/golden/ntxent_golden.npz) -- PINNED on those.
  * dataset   - src/datasets/patch_dataset.py imports by file location in the
                build container; tests/golden/patch_dataset_ref.json was
                generated from it (script committed beside it).
"""
