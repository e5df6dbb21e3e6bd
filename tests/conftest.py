import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def seeded_weights():
    """Seeded SN + DC state_dicts (seed 1234) on CPU, built from the committed key/shape manifest."""
    import numpy as np
    import torch
    from vi_depth_completion_amd import synthetic as S

    man = np.load(os.path.join(GOLDEN, "state_dict_manifest.npz"))

    def build(prefix):
        shapes = {str(k): torch.empty(eval(s), device="meta") for k, s in zip(man[prefix + "_keys"], man[prefix + "_shapes"])}     # (plain str keys: a
        #                                                  state_dict saved to a file must not carry numpy.str_ objects -- tests/test_checkpoints.py)
        return S.seeded_state_dict(shapes, 1234)

    return {"sn": build("sn"), "dc": build("dc")}


@pytest.fixture(scope="session")
def detector_weights(golden_dir):
    """Seeded plane-mask detector state_dict (seed 1234) on CPU, from the committed key/shape/anchor manifest."""
    import numpy as np
    import torch
    from vi_depth_completion_amd import synthetic as S
    man = np.load(os.path.join(golden_dir, "plane_mask_manifest.npz"))
    shapes, a = {}, 0
    for k, s in zip(man["keys"], man["shapes"]):
        shp = eval(s)
        if "anchor_generator" in k:
            shapes[str(k)] = torch.from_numpy(man["anchors"][a:a + shp[0]].copy())
            a += shp[0]
        else:
            shapes[str(k)] = torch.empty(shp, device="meta")
    return S.seeded_detector_state_dict(shapes, 1234)


