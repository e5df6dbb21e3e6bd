"""INTEGRATION.md §1 executed, not described: the reference's unchanged main.py / network_run.py construct, load and switch the package's
networks (oracle/tools/check_binding.py, a child process because it aliases `networks` in sys.modules and installs the import shims).
Runs in the build container only -- the reference does not travel to the GPU box."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/root/reference/main.py"), reason="the reference is only present in the build container")
def test_reference_harness_binds_the_hip_networks():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "tools", "check_binding.py")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["ok"] and out["reference_main"] == "/root/reference/main.py"
    assert out["sn_keys"] == 759 and out["dc_keys"] == 2031            # SURVEY 8b [probed]
    assert out["adam_parameters"] == 1026 and out["checkpoint_files_loaded"]
    assert set(out["refusals"]) == {"cpu_forward_dc", "cpu_forward_sn", "data_parallel", "train_mode_forward"}
