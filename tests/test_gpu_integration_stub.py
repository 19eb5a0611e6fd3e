"""GPU: the ctypes stub printed in INTEGRATION.md (section B) is executed as written and must give the same
features / logits as the package's own binding -- the document a maintainer would copy from stays correct."""
import os
import re

import pytest
import torch

from ss25_hierarchical_multiscale_image_classification_amd import capi, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_integration_md_stub_runs_and_matches():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## B. Bind the C ABI directly"):]
    code = re.search(r"```python\n(.*?)```", section, re.S).group(1)
    lib_path = os.path.join(ROOT, "ss25_hierarchical_multiscale_image_classification_amd", "libhipac_hip.so")
    code = code.replace('C.CDLL("libhipac_hip.so")', f'C.CDLL("{lib_path}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    sd = synth.seeded_resnet18_state_dict(0, num_classes=2)
    handle = ns["pack"](sd, 0)  # precision 0 = bf16
    x = capi.patches_normalize(synth.synth_patches_u8(5, seed=8, device="cuda"), "nchw_f32")
    feats, logits = ns["forward"](handle, x)
    torch.cuda.synchronize()
    f_ref, l_ref, _ = capi.PackedResNet18(sd, precision="bf16").forward(x, want_logits=True)
    assert torch.equal(feats, f_ref) and torch.equal(logits, l_ref)
