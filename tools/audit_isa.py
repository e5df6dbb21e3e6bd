#!/usr/bin/env python3
"""ISA audit of the BUILT library: extracts every gfx950 code object bundled in libvidc.so and fails if any kernel contains the instruction encoding that
round 6 isolated as defective on MI355X (tools/stale_read/pkmul.hip, profiles/EXPERIMENTS.md):

    packed-fp32 VALU arithmetic (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 / v_pk_min_f32 / v_pk_max_f32) whose op_sel selects the HIGH dword of src1
    (or src2: measured clean -- tools/stale_read/pkmul.hip form 10 -- refused as well) for the LOW result half

-- wrong low halves in lanes 48-63 while another wave of the SIMD issues v_mfma_f32_32x32x16_bf16 / _f16.  src0 selects, op_sel_hi, v_pk_mov_b32 and
packed ops without operand selects are measured clean and only counted.

    python tools/audit_isa.py [vi_depth_completion_amd/libvidc.so]        (exit code 1 on a finding)
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
# the one kernel that carries the encoding on purpose: the debug form behind VIDC_DBG_STEM_LOADS=3, the positive control of tests/test_stale_reads.py
ALLOWED = ("stem_conv_kernelILi3ELb1ELb1EEE",)
PAT = re.compile(r"\b(v_pk_(?:mul|add|fma|min|max)_f32)\b[^\n]*?op_sel:\[([01,]+)\]")


def audit(lib):
    tmp = tempfile.mkdtemp(prefix="vidc_audit_")
    try:
        local = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, local)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f)
        findings, allowed, n_pk, n_inst = [], [], 0, 0
        for f in objs:
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", os.path.join(tmp, f)], capture_output=True, text=True, check=True).stdout
            cur = "?"
            for ln in dis.splitlines():
                m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
                if m:
                    cur = m.group(1)
                    continue
                if "\t" in ln:
                    n_inst += 1
                if "v_pk_" in ln and "_f32" in ln:
                    n_pk += 1
                m = PAT.search(ln)
                if m:
                    sel = m.group(2).split(",")
                    if (len(sel) >= 2 and sel[1] == "1") or (len(sel) >= 3 and sel[2] == "1"):
                        (allowed if any(a in cur for a in ALLOWED) else findings).append((f, cur, ln.split("\t", 2)[-1].strip()))
        return objs, n_inst, n_pk, findings, allowed
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "vi_depth_completion_amd", "libvidc.so")
    objs, n_inst, n_pk, findings, allowed = audit(lib)
    print("audit_isa: %d gfx950 code objects, %d instructions, %d packed-fp32 instructions, %d with a src1 / src2 low-half select (+ %d in the debug control kernel)"
          % (len(objs), n_inst, n_pk, len(findings), len(allowed)))
    for f, k, ins in findings[:40]:
        print("  %s  %s:  %s" % (f.split(".")[-2] if "." in f else f, k[:80], ins))
    return 1 if findings or not objs else 0


if __name__ == "__main__":
    sys.exit(main())
