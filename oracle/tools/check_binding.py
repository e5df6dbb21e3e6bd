"""Executes INTEGRATION.md §1's recipe against THE REFERENCE'S OWN HARNESS, in the build container (CPU): alias the package
`networks` to vi_depth_completion_amd.networks, import the reference's unchanged main.py, and let its code -- not ours -- construct,
load and switch the networks.  TEST INFRASTRUCTURE (tests/test_binding.py runs it as a child process; never on the GPU box: the
reference is not there).

What the reference does with the classes, and what is checked here:
  * `RunDepthCompletion(args, None, None, ModifiedFPN, use_gravity=True)` (main.py:236-246): `DefaultImageNetwork.__init__` calls
    `net_cls()` with no arguments, `.cuda()`, and `torch.optim.Adam(cnn.parameters(), lr)` (network_run.py:86-101, 228-229, 413-424);
    `SurfaceNormalPrediction(fc_img=np.array([202., 202.])).cuda()` (main.py:243).
  * `load_network_from_file` / `load_surface_normal_network_from_file` (network_run.py:319-323, main.py:256-259): the strict
    `state = m.state_dict(); state.update(torch.load(f)); m.load_state_dict(state)` on checkpoint FILES.
  * `eval_mode()` (main.py:248-250).
  * `state_dict()` keys and shapes == the manifest recorded from the reference's own modules (tests/golden/state_dict_manifest.npz).
  * the documented refusals: a forward on CPU tensors, a forward in train mode, and DataParallel replication raise.
ref_shims supplies what this container lacks for `import main` (torchvision / cv2 / skimage / yacs stubs, `.cuda()` as identity).
Prints one JSON line; exit code 0 = every check held.
"""
import argparse
import importlib
import json
import os
import sys
import tempfile
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")


def main():
    from oracle.tools import ref_shims
    from vi_depth_completion_amd import synthetic as S
    ref_shims.install()

    # ---- INTEGRATION.md §1, verbatim --------------------------------------------------------------------------------------------------
    import vi_depth_completion_amd.networks as hip_networks
    mods = ("depth_completion", "surface_normal", "surface_normal_dorn", "warping_2dof_alignment")
    for m in mods:
        importlib.import_module("vi_depth_completion_amd.networks." + m)
    sys.modules["networks"] = hip_networks
    for m in mods:
        sys.modules["networks." + m] = getattr(hip_networks, m)
    import main as ref_main                                         # the reference, unchanged
    # ---------------------------------------------------------------------------------------------------------------------------------------
    assert os.path.realpath(ref_main.__file__) == os.path.realpath(os.path.join(ref_shims.REFERENCE_ROOT, "main.py")), ref_main.__file__
    import network_run as ref_network_run
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.networks.surface_normal import SurfaceNormalPrediction
    from vi_depth_completion_amd.networks.surface_normal_dorn import SurfaceNormalDORN
    out = {"reference_main": ref_main.__file__}
    assert ref_main.ModifiedFPN is ModifiedFPN and ref_main.SurfaceNormalPrediction is SurfaceNormalPrediction
    assert ref_main.SurfaceNormalDORN is SurfaceNormalDORN

    args = argparse.Namespace(save="", enable_multi_gpu=0, learning_rate=1e-4, batch_size=1, enriched_samples=200, dataset_type="demo")
    run = ref_main.RunDepthCompletion(args, None, None, ref_main.ModifiedFPN, use_gravity=True)      # main.py:347's call
    assert isinstance(run, ref_network_run.DefaultImageNetwork)
    assert type(run.cnn) is ModifiedFPN and type(run.surface_normal_cnn) is SurfaceNormalPrediction
    assert isinstance(run.optimizer, torch.optim.Adam)
    n_opt = sum(len(g["params"]) for g in run.optimizer.param_groups)

    man = np.load(os.path.join(ROOT, "tests", "golden", "state_dict_manifest.npz"))
    for prefix, mod in (("sn", run.surface_normal_cnn), ("dc", run.cnn)):
        sd = mod.state_dict()
        keys, shapes = [str(k) for k in man[prefix + "_keys"]], [str(s) for s in man[prefix + "_shapes"]]
        assert list(sd.keys()) == keys, "%s: state_dict keys differ from the reference's" % prefix
        assert [str(tuple(v.shape)) for v in sd.values()] == shapes, "%s: state_dict shapes differ from the reference's" % prefix
        out[prefix + "_keys"] = len(keys)
    n_param = sum(1 for k in man["dc_keys"] if not str(k).endswith(("running_mean", "running_var", "num_batches_tracked")))
    assert n_opt == n_param == len(list(run.cnn.parameters())), (n_opt, n_param)
    out["adam_parameters"] = n_opt

    # ---- checkpoint files through the reference's two loaders ------------------------------------------------------------------------------
    sn_sd = S.seeded_state_dict(run.surface_normal_cnn.state_dict(), 1234)
    dc_sd = S.seeded_state_dict(run.cnn.state_dict(), 1234)
    with tempfile.TemporaryDirectory() as tmp:
        f_sn, f_dc = os.path.join(tmp, "sn.ckpt"), os.path.join(tmp, "dc.ckpt")
        torch.save(sn_sd, f_sn)
        torch.save(dc_sd, f_dc)
        v_sn, v_dc = run.surface_normal_cnn._version, run.cnn._version
        run.load_network_from_file(f_dc)                            # network_run.py:319-323
        run.load_surface_normal_network_from_file(f_sn)             # main.py:256-259
        # a partial checkpoint (the reference's update-then-load pattern accepts it) and a checkpoint with a foreign key (strict: refused)
        part = {k: v for k, v in list(dc_sd.items())[:10]}
        torch.save(part, f_dc)
        run.load_network_from_file(f_dc)
        bad = dict(part)
        bad["not.a.key"] = torch.zeros(1)
        torch.save(bad, f_dc)
        try:
            run.load_network_from_file(f_dc)
            raise AssertionError("a checkpoint with an unknown key must be refused (strict load_state_dict)")
        except RuntimeError as e:
            assert "not.a.key" in str(e)
    for mod, sd in ((run.surface_normal_cnn, sn_sd), (run.cnn, dc_sd)):
        got = mod.state_dict()
        assert all(torch.equal(got[k], v) for k, v in sd.items()), "loaded values differ from the checkpoint file"
    assert run.surface_normal_cnn._version > v_sn and run.cnn._version > v_dc          # derived (packed / folded) copies were dropped
    out["checkpoint_files_loaded"] = True

    run.eval_mode()                                                 # main.py:248-250
    assert not run.cnn.training and not run.surface_normal_cnn.training

    # ---- documented refusals ------------------------------------------------------------------------------------------------------------
    x = torch.zeros(1, 3, 240, 320)
    refusals = {}
    for name, fn in (("cpu_forward_dc", lambda: run.cnn(x, x, x[:, :1])),
                     ("cpu_forward_sn", lambda: run.surface_normal_cnn(x, torch.tensor([[0., 1., 0.]]), torch.tensor([[0., 1., 0.]]))),
                     ("data_parallel", lambda: run.cnn._replicate_for_data_parallel())):
        try:
            fn()
            raise AssertionError(name + " did not raise")
        except RuntimeError as e:
            refusals[name] = str(e)[:80]
    run.cnn.train()
    try:
        run.cnn(x, x, x[:, :1])
        raise AssertionError("forward in train mode did not raise")
    except RuntimeError as e:
        refusals["train_mode_forward"] = str(e)[:80]
    run.eval_mode()
    out["refusals"] = refusals
    # DORN variant of the same constructor (main.py:244-245)
    run2 = ref_main.RunDepthCompletion(args, None, None, ref_main.ModifiedFPN, use_gravity=False)
    assert type(run2.surface_normal_cnn) is SurfaceNormalDORN
    out["ok"] = True
    print(json.dumps(out))


if __name__ == "__main__":
    main()
