"""Variants of base.s: edits inside the patch loop of stem_conv_kernel<3, true, 2> (the failing form), one .s per named edit.
    python patch.py <name> [<name> ...]     (see EDITS)"""
import re
import sys

KER = "_ZN12_GLOBAL__N_116stem_conv_kernelILi3ELb1ELi2EEEvPKfS2_PfiiiiiiiPtiS2_ffi:"
lines = open("base.s").read().split("\n")
k0 = next(i for i, l in enumerate(lines) if l.startswith(KER))
k1 = next(i for i in range(k0, len(lines)) if "s_endpgm" in lines[i])
l0 = next(i for i in range(k0, k1) if lines[i].startswith(".LBB8_5:"))        # patch loop header
l1 = next(i for i in range(l0, k1) if lines[i].startswith(".LBB8_17:"))
body = lambda: list(range(l0, l1))


def code(l):
    return l.split(";")[0].strip()


def ed_vm0(out):           # every counted vmcnt wait of the loop -> vmcnt(0)
    for i in body():
        if re.match(r"s_waitcnt vmcnt\(\d+\)", code(out[i])):
            out[i] = "\ts_waitcnt vmcnt(0)"


def ed_lgkm_after_ds(out):  # an lgkmcnt(0) behind every ds_write of the loop
    for i in body():
        if code(out[i]).startswith("ds_write"):
            out[i] += "\n\ts_waitcnt lgkmcnt(0)"


def ed_nop_exec(out):      # idle cycles behind every write of EXEC in the loop
    for i in body():
        c = code(out[i])
        if re.match(r"s_(or|xor|and|andn2)_b64 exec,", c) or "saveexec" in c:
            out[i] += "\n\ts_nop 15\n\ts_nop 15"


def ed_nop_wait(out):      # idle cycles behind every vmcnt wait
    for i in body():
        if re.match(r"s_waitcnt vmcnt", code(out[i])):
            out[i] += "\n\ts_nop 15"


def ed_vm0_first8(out):    # only the waits of the first two channels' eight loads
    n = 0
    for i in body():
        if re.match(r"s_waitcnt vmcnt\(\d+\)", code(out[i])):
            n += 1
            if n <= 8:
                out[i] = "\ts_waitcnt vmcnt(0)"


def ed_vm0_last4(out):     # only the third channel's waits
    idx = [i for i in body() if re.match(r"s_waitcnt vmcnt\(\d+\)", code(out[i]))]
    for i in idx[8:]:
        out[i] = "\ts_waitcnt vmcnt(0)"


def ed_wait_before_ds(out):   # vmcnt(0) in front of every ds_write
    for i in body():
        if code(out[i]).startswith("ds_write"):
            out[i] = "\ts_waitcnt vmcnt(0)\n" + out[i]


def ed_wait_loop_end(out):    # drain loads + LDS at the loop's back edge (.LBB8_4)
    i = next(i for i in range(k0, k1) if lines[i].startswith(".LBB8_4:"))
    out[i] += "\n\ts_waitcnt vmcnt(0) lgkmcnt(0)"


def ed_barriers(out):      # every s_barrier of the kernel: drain everything first, idle cycles behind
    for i in range(k0, k1):
        if code(out[i]) == "s_barrier":
            out[i] = "\ts_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier\n\ts_nop 15"


def ed_vgpr72(out):        # a larger register allocation for the wave (61 -> 72: another allocation granule)
    for i in range(k1, len(out)):
        if ".amdhsa_kernel " + KER[:-1] in out[i]:
            j = i
            while ".end_amdhsa_kernel" not in out[j]:
                if ".amdhsa_next_free_vgpr" in out[j]:
                    out[j] = "\t\t.amdhsa_next_free_vgpr 72"
                j += 1
            return
    raise SystemExit("kernel descriptor not found")


def ed_phase2_waits(out):  # compute phase: drain loads and LDS behind every LDS read / global load
    for i in range(l1, k1):
        c = code(out[i])
        if c.startswith("ds_read") or c.startswith("global_load"):
            out[i] += "\n\ts_waitcnt vmcnt(0) lgkmcnt(0)"


def ed_sleep_loop(out):    # the patch loop slowed down
    out[l0] += "\n\ts_sleep 4"


def ed_stores_wait(out):   # drain behind every global store of the kernel
    for i in range(k0, k1):
        if code(out[i]).startswith("global_store"):
            out[i] += "\n\ts_waitcnt vmcnt(0)"


PK = "v_pk_mul_f32 v[28:29], v[36:37], v[28:29] op_sel:[0,1] op_sel_hi:[1,0]"


def _pk_line(out):
    idx = [i for i in body() if code(out[i]) == PK]
    assert len(idx) == 1, idx
    return idx[0]


def ed_pk_split(out):      # the in-place cross-half packed multiply as two scalar multiplies through a spare register (v61: allocation 61 -> 64)
    i = _pk_line(out)
    out[i] = "\tv_mul_f32_e32 v61, v36, v29\n\tv_mul_f32_e32 v29, v37, v28\n\tv_mov_b32_e32 v28, v61"
    ed_vgpr72(out)


def ed_pk_nop_before(out):
    i = _pk_line(out)
    out[i] = "\ts_nop 7\n" + out[i]


def ed_pk_nop_after(out):
    i = _pk_line(out)
    out[i] += "\n\ts_nop 7"


def ed_pk_noninplace(out):  # the same packed multiply into a fresh register pair, then two moves (v[62:63])
    i = _pk_line(out)
    out[i] = "\tv_pk_mul_f32 v[62:63], v[36:37], v[28:29] op_sel:[0,1] op_sel_hi:[1,0]\n\ts_nop 1\n\tv_mov_b32_e32 v28, v62\n\tv_mov_b32_e32 v29, v63"
    ed_vgpr72(out)


EDITS = {k[3:]: v for k, v in list(globals().items()) if k.startswith("ed_")}
for name in sys.argv[1:]:
    out = list(lines)
    for part in name.split("+"):
        EDITS[part](out)
    open(name.replace("+", "_") + ".s", "w").write("\n".join(out))
    print("wrote", name)
