"""Build container only: the reference's own harness code (utils/core_utils_mtl_concat.py, utils/eval_utils_mtl_concat.py, utils/utils.py,
imported UNMODIFIED from /root/reference) bound to this repository's model through a `models.model_toad` pre-registration - INTEGRATION.md
Option B, executed (host side: everything up to relocate() / forward, which need a HIP device). Skipped where /root/reference is absent."""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(not os.path.isdir("/root/reference/utils"), reason="the reference tree only exists in the build container")
@pytest.mark.timeout(300)
def test_reference_harness_drives_the_dropin_host_side():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_option_b_probe.py")], capture_output=True, text=True, timeout=280, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    for marker in ("BOUND core_utils and eval_utils to toad_amd.model_toad", "Total number of trainable parameters: 1192490", "OPTIM ok",
                   "CHECKPOINT keys ok: 14", "INITIATE_MODEL ok", "OPTION_B_OK"):
        assert marker in r.stdout, (marker, r.stdout[-2000:])
    assert "Validation loss decreased (inf --> 1.250000)" in r.stdout        # the reference's EarlyStopping printed it, saving OUR state_dict
