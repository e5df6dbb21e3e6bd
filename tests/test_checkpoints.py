"""Weights travel in and out of the reference as FILES: `torch.save(self.cnn.state_dict(), path)` (network_run.py:335-337) and
`state = m.state_dict(); state.update(torch.load(path)); m.load_state_dict(state)` (network_run.py:319-323, main.py:256-259).  The
drop-in keeps both contracts: `DepthCompletionPipeline.load_network_from_file` / `load_surface_normal_network_from_file` mirror the
reference's loaders (strict), the modules' `state_dict()` has the reference's keys, and everything derived from the parameters on the
device (packed weights, folded BatchNorm affines, recorded programs, captured graphs) is rebuilt after a load, a `.cuda()` / `.to()`
and a training step.  (missing 2.)"""
import numpy as np
import pytest
import torch

from vi_depth_completion_amd import synthetic as S

pytestmark = pytest.mark.gpu
torch.set_grad_enabled(False)
DEV = "cuda"


def _pipe(rng_seed=0):
    from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
    p = DepthCompletionPipeline(enriched_samples=200, rng=np.random.RandomState(rng_seed))
    p.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
    return p


def _batch(frame0=11):
    return {k: (v.to(DEV) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, 240, 320, 1234, frame0=frame0).items()}


@pytest.fixture(scope="module")
def checkpoint_files(seeded_weights, tmp_path_factory):
    d = tmp_path_factory.mktemp("ckpt")
    sn_path, dc_path = str(d / "surface_normal.ckpt"), str(d / "depth_completion.ckpt")
    torch.save(seeded_weights["sn"], sn_path)            # what a training run of the reference leaves behind (network_run.py:335-337)
    torch.save(seeded_weights["dc"], dc_path)
    return sn_path, dc_path


def test_checkpoint_files_load_like_the_in_memory_state(seeded_weights, checkpoint_files):
    sn_path, dc_path = checkpoint_files
    a = _pipe()
    a.load_state_dicts(seeded_weights["sn"], seeded_weights["dc"])
    b = _pipe()
    b.load_surface_normal_network_from_file(sn_path)     # main.py:256-259
    b.load_network_from_file(dc_path)                    # network_run.py:319-323
    for ma, mb in ((a.surface_normal_cnn, b.surface_normal_cnn), (a.cnn, b.cnn)):
        sa, sb = ma.state_dict(), mb.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        for k in sa:
            assert torch.equal(sa[k], sb[k]), k
    batch = _batch()
    for p in (a, b):
        p.rng = np.random.RandomState(5)
    ya, yb = a._call_cnn(batch).cpu(), b._call_cnn(batch).cpu()
    assert torch.equal(ya, yb) and float(ya.max()) > 0.0
    # the stream mode bench.py times records other programs (joint 4-group frame program) from the same stores
    for p in (a, b):
        p.rng = np.random.RandomState(5)
    sa = [o.cpu() for o in a.run_interleaved(iter([batch, _batch(12), _batch(13)]), lanes=2, frames_per_launch=2)]
    sb = [o.cpu() for o in b.run_interleaved(iter([batch, _batch(12), _batch(13)]), lanes=2, frames_per_launch=2)]
    assert all(torch.equal(x, y) for x, y in zip(sa, sb))


def test_loaders_are_strict_and_partial_files_update(seeded_weights, checkpoint_files, tmp_path):
    """`state.update(torch.load(..))` + strict `load_state_dict`: a key the module does not know raises; a file holding only some keys
    replaces those and keeps the rest (that is how the reference's own checkpoints without `num_batches_tracked` load)."""
    sn_path, dc_path = checkpoint_files
    p = _pipe()
    p.load_surface_normal_network_from_file(sn_path)
    p.load_network_from_file(dc_path)
    extra = dict(list(seeded_weights["dc"].items())[:4])
    extra["not_a_parameter.weight"] = torch.zeros(3)
    bad = str(tmp_path / "extra_key.ckpt")
    torch.save(extra, bad)
    with pytest.raises(RuntimeError, match="Unexpected key"):
        p.load_network_from_file(bad)
    wrong = {k: v for k, v in list(seeded_weights["sn"].items())[:2]}
    k0 = next(iter(wrong))
    wrong[k0] = torch.zeros(tuple(s + 1 for s in wrong[k0].shape))
    bad2 = str(tmp_path / "wrong_shape.ckpt")
    torch.save(wrong, bad2)
    with pytest.raises(RuntimeError, match="size mismatch"):
        p.load_surface_normal_network_from_file(bad2)
    # a partial file: only the last conv's bias changes
    batch = _batch()
    p.rng = np.random.RandomState(3)
    y0 = p._call_cnn(batch).cpu()
    part = str(tmp_path / "partial.ckpt")
    torch.save({"feature_concat.2.bias": seeded_weights["dc"]["feature_concat.2.bias"] + 0.5}, part)
    p.load_network_from_file(part)
    p.rng = np.random.RandomState(3)
    y1 = p._call_cnn(batch).cpu()
    assert not torch.equal(y0, y1)
    assert abs(float((y1 - y0)[:, :, 1:-1, 1:-1].mean()) - 0.5) < 0.05      # the 1x1 head conv's bias moves the (interior of the) map by 0.5
    sd = p.cnn.state_dict()
    for k, v in seeded_weights["dc"].items():
        if k != "feature_concat.2.bias":
            assert torch.equal(sd[k].cpu(), v), k


def test_reload_and_device_moves_invalidate_the_packed_weights(seeded_weights, checkpoint_files):
    """Programs, packed weights and captured graphs derive from the parameters: a second load (other weights) must not leave any of
    them behind, neither for `_call_cnn`'s programs nor for the stream mode's joint program, and `.cpu()` -> `.cuda()` (what
    `network_run.py:100` does to a freshly built module) must rebuild them from the moved tensors."""
    sn_path, dc_path = checkpoint_files
    p = _pipe()
    p.load_surface_normal_network_from_file(sn_path)
    p.load_network_from_file(dc_path)
    batch = _batch(21)
    p.rng = np.random.RandomState(9)
    y_seed = p._call_cnn(batch).cpu()
    p.rng = np.random.RandomState(9)
    s_seed = [o.cpu() for o in p.run_interleaved(iter([batch, batch]), lanes=1, frames_per_launch=2)]
    other_dc = S.seeded_state_dict(p.cnn.state_dict(), 4321)
    p.load_state_dicts({}, other_dc)
    p.rng = np.random.RandomState(9)
    y_other = p._call_cnn(batch).cpu()
    p.rng = np.random.RandomState(9)
    s_other = [o.cpu() for o in p.run_interleaved(iter([batch, batch]), lanes=1, frames_per_launch=2)]
    assert not torch.equal(y_seed, y_other) and not torch.equal(s_seed[0], s_other[0])
    fresh = _pipe()
    fresh.load_surface_normal_network_from_file(sn_path)
    fresh.load_state_dicts({}, other_dc)
    fresh.rng = np.random.RandomState(9)
    assert torch.equal(fresh._call_cnn(batch).cpu(), y_other)
    # back to the file, through a round trip over the host
    p.load_network_from_file(dc_path)
    v0 = p.cnn._version
    p.cnn.cpu()
    p.cnn.cuda()
    assert p.cnn._version > v0
    p.rng = np.random.RandomState(9)
    assert torch.equal(p._call_cnn(batch).cpu(), y_seed)
    p.rng = np.random.RandomState(9)
    s_again = [o.cpu() for o in p.run_interleaved(iter([batch, batch]), lanes=1, frames_per_launch=2)]
    assert all(torch.equal(a, b) for a, b in zip(s_seed, s_again))


def test_trained_weights_survive_save_and_load(seeded_weights, tmp_path):
    """network_run.py:231-254 then :335-337: one training iteration (the trainer turns the parameters into views of its flat buffers),
    `torch.save(cnn.state_dict())`, a FRESH ModifiedFPN loads the file through the reference's loader: same state, and in eval mode the
    same depth map bit for bit -- from the trained module itself (whose inference programs must have been re-recorded from the
    stepped parameters and the updated running statistics) and from the fresh one."""
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    cnn = ModifiedFPN().to(DEV)
    cnn.load_state_dict({k: v.to(DEV) for k, v in seeded_weights["dc"].items()})
    b = S.synthetic_batch(2, 240, 320, 1234, frame0=30)
    image = b["image"].to(DEV)
    normal = torch.nn.functional.normalize(image - 0.5, dim=1)
    depth_in = b["sparse_depth"].to(DEV)
    gt = S.synthetic_ground_truth_depth(b["image"], 1234).to(DEV)
    cnn.eval()
    before = cnn(image, normal, depth_in).cpu()
    cnn.train()
    tr = DepthCompletionTrainer(cnn, 1e-3)
    with torch.enable_grad():
        loss = tr.step(image, normal, depth_in, gt)
    assert np.isfinite(float(loss))
    path = str(tmp_path / "trained.ckpt")
    torch.save(cnn.state_dict(), path)                   # network_run.py:335-337
    cnn.eval()
    after = cnn(image, normal, depth_in).cpu()
    assert not torch.equal(before, after), "the training step changed nothing"
    loaded = torch.load(path)
    assert list(loaded.keys()) == list(seeded_weights["dc"].keys())
    moved = sum(int(not torch.equal(loaded[k].cpu(), seeded_weights["dc"][k])) for k in loaded)
    assert moved > 1000, moved                            # parameters, running statistics, num_batches_tracked all stepped
    assert int(loaded["resnet_rgb.bn1.num_batches_tracked"]) == int(seeded_weights["dc"]["resnet_rgb.bn1.num_batches_tracked"]) + 1

    from vi_depth_completion_amd.pipeline import DepthCompletionPipeline
    fresh = DepthCompletionPipeline(enriched_samples=0)
    fresh.load_network_from_file(path)                    # network_run.py:319-323
    for k, v in cnn.state_dict().items():
        assert torch.equal(fresh.cnn.state_dict()[k], v), k
    assert torch.equal(fresh.cnn(image, normal, depth_in).cpu(), after)
    # the trainer keeps going on the same module after the evaluation pass (network_run.py alternates train and eval)
    cnn.train()
    with torch.enable_grad():
        loss2 = tr.step(image, normal, depth_in, gt)
    assert np.isfinite(float(loss2)) and float(loss2) != float(loss)
