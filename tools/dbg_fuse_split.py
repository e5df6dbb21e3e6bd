"""Debug: the same frame with VIDC_FUSE_SPLIT=1 / 0 (split images written by the producing kernel / by split launches of their own), per mode."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from vi_depth_completion_amd import synthetic as S
    from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
    torch.set_grad_enabled(False)
    pipe = DepthCompletionPipeline(enriched_samples=200, rng=np.random.RandomState(3))
    pipe.load_state_dicts(S.seeded_state_dict(pipe.surface_normal_cnn.state_dict(), 1234, device="cuda"), S.seeded_state_dict(pipe.cnn.state_dict(), 1234, device="cuda"))
    pipe.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(240, 320))
    b = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, 240, 320, 1234, frame0=60).items()}
    out = pipe._call_cnn(b).cpu()
    torch.save(out, sys.argv[1])
    print(os.environ.get("VIDC_PRECISION"), os.environ.get("VIDC_FUSE_SPLIT"), os.environ.get("VIDC_WINOGRAD", "auto"), "mean %.4f max %.3e" % (float(out.mean()), float(out.abs().max())))
    sys.exit(0)
import torch
for mode in ("mixed", "fp32"):
    for wino in ("auto", "0"):
        outs = []
        for fs in ("1", "0"):
            f = "/tmp/fs_%s_%s_%s.pt" % (mode, wino, fs)
            subprocess.run([sys.executable, __file__, f], env=dict(os.environ, VIDC_PRECISION=mode, VIDC_FUSE_SPLIT=fs, VIDC_WINOGRAD=wino), check=True)
            outs.append(torch.load(f))
        d = (outs[0] - outs[1]).abs()
        print("== %s winograd=%s: fused vs separate split: max|diff| %.3e, equal %s" % (mode, wino, float(d.max()), bool(torch.equal(outs[0], outs[1]))))
