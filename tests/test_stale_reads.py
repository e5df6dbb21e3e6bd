"""Stress test of the stream mode against wrong reads (the whole investigation is in profiles/EXPERIMENTS.md, round 6).

What round 5 saw -- one frame in ten off by 1e-3 with the opt-in fused warp + stem kernel, read as "stale cache lines" -- is reproduced at will in round 6
and traced to ONE instruction: `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[1,0]`, which hipcc's SLP vectoriser had emitted for the two cross bilinear weights
in the kernel's plain-load form.  On MI355X a packed-fp32 op whose low half takes the high dword of src1 returns a wrong low half in lanes 48-63 while another
wave of the SIMD issues v_mfma_f32_32x32x16_bf16 (tools/stale_read/pkmul.hip: 100 lines, no memory) -- i.e. in the mixed mode, beside another lane's bf16x3
conv, on any number of streams but never in fp32, never alone.  The library no longer contains the encoding (csrc/Makefile, tests/test_isa_audit.py); the
debug form `VIDC_DBG_STEM_LOADS=3` puts it back ON PURPOSE as an identity multiply and is the positive control here: the same stream, the same comparison,
and it MUST differ -- so a green run of the default path means the comparison can see the defect and did not.

Every case is a process of its own (tools/stale_read/stress_pipeline.py): the kernel form and the hardware-queue count are read once per process.
Reference of every case: the same items through ONE lane executed eagerly by a second pipeline object (no graphs, no overlap).  Each item has its own image
and gravity (per-frame homography: networks/warping_2dof_alignment.py:35-58,108-156).
"""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = os.path.join(ROOT, "tools", "stale_read", "stress_pipeline.py")
_STRIP = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "VIDC_LANES", "VIDC_FRAMES_PER_LAUNCH", "VIDC_EXEC", "VIDC_FUSE_WARP",
          "VIDC_DBG_STEM_LOADS", "GPU_MAX_HW_QUEUES", "VIDC_PRECISION")


def _stress(env, items, runs=2, lanes=3, F=4, timeout=900):
    e = {k: v for k, v in os.environ.items() if k not in _STRIP}
    e.update(env)
    p = subprocess.run([sys.executable, SCRIPT, "--items", str(items), "--runs", str(runs), "--lanes", str(lanes), "--F", str(F)],
                       env=e, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    m = re.search(r"STRESS total differing items: (\d+)", p.stdout)
    assert m is not None, "stress run did not finish (rc %d):\n%s\n%s" % (p.returncode, p.stdout[-2000:], p.stderr[-3000:])
    return int(m.group(1)), p.stdout


@pytest.mark.parametrize("precision", ["fp32", "mixed"])
def test_default_path_2000_items_three_lanes_graphs_vs_one_lane_eager(precision):
    """The timed configuration (3 lanes, 4 items per launch, captured graphs): 2000 items x 2 runs, every depth map bit-equal to lanes=1 eager."""
    bad, out = _stress({"VIDC_PRECISION": precision}, items=2000, runs=2)
    assert bad == 0, out


@pytest.mark.parametrize("precision", ["fp32", "mixed"])
def test_default_path_with_eight_hardware_queues(precision):
    bad, out = _stress({"VIDC_PRECISION": precision, "GPU_MAX_HW_QUEUES": "8"}, items=1000, runs=2)
    assert bad == 0, out


def test_default_path_one_item_per_launch_and_two_lanes():
    """Other phase relations between the lanes: F = 1 on three lanes, F = 2 on two."""
    bad, out = _stress({"VIDC_PRECISION": "mixed"}, items=800, runs=2, lanes=3, F=1)
    assert bad == 0, out
    bad, out = _stress({"VIDC_PRECISION": "mixed"}, items=800, runs=2, lanes=2, F=2)
    assert bad == 0, out


def test_positive_control_round5_kernel_form_is_caught():
    """The opt-in fused stem with the defective packed multiply put back (VIDC_DBG_STEM_LOADS=3): the stream MUST differ from the reference (measured:
    8-15 % of the items per run, i.e. P(no difference in 2 x 800 items) is nil) -- and the form that ships behind VIDC_FUSE_WARP=1 must not."""
    bad, out = _stress({"VIDC_PRECISION": "mixed", "VIDC_FUSE_WARP": "1", "VIDC_DBG_STEM_LOADS": "3"}, items=800, runs=2)
    if bad == 0:          # (never seen on the MI355X boxes of rounds 5 and 6; a longer second look before concluding anything)
        bad, out = _stress({"VIDC_PRECISION": "mixed", "VIDC_FUSE_WARP": "1", "VIDC_DBG_STEM_LOADS": "3"}, items=2000, runs=3)
    if bad == 0:
        # The control provokes a HARDWARE defect; a part that does not have it (another stepping, a microcode fix) would make this assertion fail for a good
        # reason.  The suite then says so loudly instead of going red: on such a part the default-path tests above lose their proof of sensitivity.
        pytest.skip("the defective packed-fp32 instruction computed correctly in 7 600 item-runs beside bf16x3 convs: this GPU does not show the MI355X defect "
                    "(DESIGN 4.5); the positive control is inconclusive here")
    bad, out = _stress({"VIDC_PRECISION": "mixed", "VIDC_FUSE_WARP": "1", "VIDC_DBG_STEM_LOADS": "0"}, items=800, runs=2)
    assert bad == 0, out
