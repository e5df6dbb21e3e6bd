"""CPU: the BUILT libvidc.so carries no instance of the instruction encoding that round 6 isolated as defective on MI355X -- packed-fp32 VALU arithmetic
whose LOW result half selects the HIGH dword of src1 (`v_pk_mul_f32 ... op_sel:[0,1]`): wrong low halves in lanes 48-63 beside a wave issuing the 16-k
bf16 / f16 MFMAs (tools/stale_read/pkmul.hip; profiles/EXPERIMENTS.md).  hipcc's SLP vectoriser had put it into the fused stem's plain-load form (the
round-5 "stale read") and into warp_params_kernel (live lanes 48-63 from program batch 49 on); those files (and, since, wfused.hip, where this test caught it) are built with -fno-slp-vectorize
(csrc/Makefile: everywhere else packed fp32 stays, it is worth 2.6 % of the conv kernels) and this test disassembles what was built.  The one allowed instance is the debug kernel that carries it on purpose (the stress test's positive control)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_built_library_has_no_defective_packed_fp32_encoding():
    import audit_isa
    from vi_depth_completion_amd import _lib as L
    if not os.path.exists(os.path.join(audit_isa.LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump of the ROCm toolchain not found")
    L.build()
    objs, n_inst, n_pk, findings, allowed = audit_isa.audit(L.LIB_PATH)
    assert len(objs) >= 10 and n_inst > 100000, (objs, n_inst)            # the audit really saw the device code
    assert not findings, findings[:5]
    assert len(allowed) == 1 and "stem_conv_kernelILi3ELb1ELb1EEE" in allowed[0][1]      # the positive control is still there


def test_audit_recognises_the_encodings():
    import audit_isa
    hit = lambda s: (lambda m: m is not None and ((len(m.group(2).split(",")) >= 2 and m.group(2).split(",")[1] == "1") or      # noqa: E731
                                                  (len(m.group(2).split(",")) >= 3 and m.group(2).split(",")[2] == "1")))(audit_isa.PAT.search(s))
    assert hit("\tv_pk_mul_f32 v[28:29], v[36:37], v[28:29] op_sel:[0,1] op_sel_hi:[1,0]")
    assert hit("\tv_pk_add_f32 v[4:5], v[12:13], v[4:5] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]")
    assert hit("\tv_pk_fma_f32 v[2:3], v[14:15], v[2:3], v[6:7] op_sel:[0,0,1] op_sel_hi:[1,1,0]")
    assert not hit("\tv_pk_mul_f32 v[22:23], v[20:21], 0.5 op_sel_hi:[1,0]")            # high-half select only: measured clean
    assert not hit("\tv_pk_fma_f32 v[2:3], v[2:3], s[0:1], v[6:7] op_sel:[1,0,0]")     # src0 select: measured clean
    assert not hit("\tv_pk_mov_b32 v[34:35], v[28:29], v[20:21] op_sel:[1,0]")           # not arithmetic: measured clean
