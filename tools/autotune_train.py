#!/usr/bin/env python3
"""Measures every conv launch of one training step (forward convs, dgrad convs and the wgrad GEMMs of training.DepthCompletionTrainer)
over the tilings / split-K factors of the conv kernel and writes `vi_depth_completion_amd/train_tuning.json`
(signature -> [tile_fp32, splitk_fp32, tile_bf16x3, splitk_bf16x3]) that the trainer consults.

    python tools/autotune_train.py --batch 8

The trainer calls `tune_hook(desc, role)` before it plans a launch; the hook times the candidates on the launch's own tensors
(captured hipGraph of back-to-back launches, 512 MB of junk written before each replay so that nothing is cache-warm across replays)
and the step then proceeds with the cost-model plan -- its results are not used.  All tilings compute the same fp32 sums in a different
association order; tests/test_training.py runs with whatever table is committed.
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from autotune import TILE_DIMS  # noqa: E402
from vi_depth_completion_amd import _lib as L, engine, synthetic as S, training  # noqa: E402
from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN  # noqa: E402

OUT = os.path.join(ROOT, "vi_depth_completion_amd", "train_tuning.json")


def time_launches(lib, d, st, pool, junk, copies):
    """Average GPU time (us) of `copies` back-to-back launches of `d` in one captured graph; weights of small layers rotate through
    `pool` (cold per launch, as in a step where 1.2 GB of parameters are touched between two uses)."""
    wbytes = d.groups * d.Cout * d.KH * d.KW * d.Cin * 4
    stride = (wbytes + 255) // 256 * 256
    rotate = pool is not None and 2 * stride <= pool.numel() * 4
    if rotate:
        copies = int(max(2, min(copies, pool.numel() * 4 // stride)))
    if lib.vidc_conv2d_bn_act(C.byref(d), st) != 0:
        return None
    ops = (L.Op * copies)()
    for i, o in enumerate(ops):
        o.kind = L.OP_CONV
        C.memmove(C.byref(o.u.conv), C.byref(d), C.sizeof(L.ConvDesc))
        if rotate:
            o.u.conv.w = pool.data_ptr() + i * stride
    h = C.c_void_p()
    if lib.vidc_program_create(ops, copies, C.byref(h)) != 0:
        return None
    best = None
    try:
        if lib.vidc_program_capture(h, st) == 0:
            ms = (C.c_float * 1)()
            for rep in range(2):
                junk.fill_(rep)
                torch.cuda.synchronize()
                if lib.vidc_program_time(h, st, 1, 1, ms, None) != 0:
                    return None
                us = ms[0] * 1e3 / copies
                best = us if best is None else min(best, us)
    finally:
        lib.vidc_program_destroy(h)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--only-missing", action="store_true")
    ap.add_argument("--precisions", default="0,1")
    ap.add_argument("--bf16", action="store_true", help="run the step in the plain-bf16 mode (VIDC_TRAIN_PRECISION=bf16) and measure its launches\n"
                                                        "(K counted in 64-channel units, so they are signatures of their own): fills slots 4, 5 of the entries")
    ap.add_argument("--remeasure", action="store_true", help="with --bf16: measure every bf16 launch again (e.g. after new tilings were added) instead of "
                                                             "only the signatures without a bf16 entry; the fp32 / bf16x3 slots are kept")
    ap.add_argument("--out", default=OUT)
    a = ap.parse_args()
    if a.bf16:
        os.environ["VIDC_TRAIN_PRECISION"] = "bf16"
        a.only_missing = True
    dev = torch.device("cuda")
    torch.set_grad_enabled(False)
    lib = L.lib()
    table = json.load(open(OUT)) if (a.only_missing and os.path.exists(OUT)) else {}
    training._TRAIN_TUNING = {}
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    st = side.cuda_stream
    ws = torch.zeros(96 << 20, dtype=torch.float32, device=dev)       # 384 MB split-K scratch
    pool = torch.randn(64 << 20, dtype=torch.float32, device=dev) * 0.05
    junk = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    precs = [int(v) for v in a.precisions.split(",")]
    seen, total, remeasured = {}, {"plan": 0.0, "best": 0.0}, set()

    def hook(d0, role):
        sig = engine.conv_signature(d0)
        seen[sig] = seen.get(sig, 0) + 1
        if d0.precision == 2:
            return hook_bf16(d0, role, sig)
        if sig in table:
            return
        d = L.ConvDesc.from_buffer_copy(d0)
        d.flags &= ~L.ACCUM
        d.workspace = ws.data_ptr()
        M, units = d.B * d.Ho * d.Wo, d.KH * d.KW * d.Cin // 32
        big_w = d.groups * d.Cout * d.KH * d.KW * d.Cin * 4 > (32 << 20)
        ent, line = [], []
        for prec in (0, 1):
            if prec not in precs or (prec == 1 and d0.precision == 0 and role == "gemm" and False):
                ent += [0, 0]
                continue
            d.precision = prec
            # the cost-model plan, for the record
            p = L.ConvDesc.from_buffer_copy(d)
            p.tile = 0
            lib.vidc_conv2d_plan(C.byref(p))
            if role == "conv":
                p.splitk = 1
            us_plan = time_launches(lib, p, st, None if big_w else pool, junk, 6 if big_w else 16)
            cands = []
            for t, (bm, bn) in TILE_DIMS.items():
                if bn > max(64, d.Cout) or bm >= 4 * max(32, M):
                    continue
                wgs = -(-M // bm) * -(-d.Cout // bn) * d.groups
                for sk in (1, 2, 4, 8, 16, 32, 64):
                    if sk > 1 and (units // sk < 4 or wgs * (sk // 2) >= 2048):
                        continue
                    if sk > 1 and L.SPLITK_COUNTERS + sk * d.groups * M * d.Cout > ws.numel():
                        continue
                    d.tile, d.splitk = t, sk
                    us = time_launches(lib, d, st, None if big_w else pool, junk, 6 if big_w else 16)
                    if us is not None:
                        cands.append((us, t, sk))
            cands.sort()
            if not cands:
                raise RuntimeError("no tiling ran for %s: %s" % (sig, lib.vidc_last_error().decode()))
            us, t, sk = cands[0]
            ent += [t, sk]
            line.append("%s plan %-12s sk%-2d %8.1f us -> %-12s sk%-2d %8.1f us" % ("bf16x3" if prec else "fp32  ", L.TILE_NAMES[p.tile], p.splitk,
                                                                                 us_plan or -1, L.TILE_NAMES[t], sk, us))
            if prec == 0:
                table[sig + "#us"] = [us_plan, us]
        table[sig] = ent
        print("%-5s %-36s %s" % (role, sig, " | ".join(line)), flush=True)

    def hook_bf16(d0, role, sig):
        ent = table.get(sig, [0, 0, 0, 0])
        if len(ent) >= 6 and ent[4] and not (a.remeasure and sig not in remeasured):
            return
        remeasured.add(sig)
        d = L.ConvDesc.from_buffer_copy(d0)
        d.flags &= ~L.ACCUM
        d.workspace = ws.data_ptr()
        M, units = d.B * d.Ho * d.Wo, d.KH * d.KW * d.Cin // 32
        big_w = d.groups * d.Cout * d.KH * d.KW * d.Cin * 4 > (32 << 20)
        p = L.ConvDesc.from_buffer_copy(d)
        p.tile = 0
        lib.vidc_conv2d_plan(C.byref(p))
        if role == "conv":
            p.splitk = 1
        us_plan = time_launches(lib, p, st, None if big_w else pool, junk, 6 if big_w else 16)
        cands = []
        for t, (bm, bn) in TILE_DIMS.items():
            if bn > max(64, d.Cout) or bm >= 4 * max(32, M):
                continue
            wgs = -(-M // bm) * -(-d.Cout // bn) * d.groups
            for sk in (1, 2, 4, 8, 16, 32, 64):
                if sk > 1 and (units // sk < 4 or wgs * (sk // 2) >= 2048):
                    continue
                if sk > 1 and L.SPLITK_COUNTERS + sk * d.groups * M * d.Cout > ws.numel():
                    continue
                d.tile, d.splitk = t, sk
                us = time_launches(lib, d, st, None if big_w else pool, junk, 6 if big_w else 16)
                if us is not None:
                    cands.append((us, t, sk))
        cands.sort()
        if not cands:
            raise RuntimeError("no tiling ran for %s: %s" % (sig, lib.vidc_last_error().decode()))
        us, t, sk = cands[0]
        table[sig] = (list(ent) + [0, 0, 0, 0])[:4] + [t, sk]
        table[sig + "#us"] = [us_plan, us]
        was = ""
        if len(ent) >= 6 and ent[4]:
            us_was = dict(((tt, ss), u) for u, tt, ss in cands).get((ent[4], ent[5]))
            was = "   (table had %s sk%d: %s us)" % (L.TILE_NAMES[ent[4]], ent[5], "%.1f" % us_was if us_was is not None else "-")
        print("%-5s %-36s bf16   plan %-12s sk%-2d %8.1f us -> %-12s sk%-2d %8.1f us%s" % (role, sig, L.TILE_NAMES[p.tile], p.splitk, us_plan or -1,
                                                                                       L.TILE_NAMES[t], sk, us, was), flush=True)

    cnn = ModifiedFPN().to(dev)
    cnn.load_state_dict(S.seeded_state_dict(cnn.state_dict(), 1234, device=dev))
    cnn.train()
    tr = training.DepthCompletionTrainer(cnn, 1e-4)
    tr.tune_hook = hook
    b = S.synthetic_batch(a.batch, 240, 320, 1234)
    image = b["image"].to(dev)
    normal = torch.nn.functional.normalize(image - 0.5, dim=1)
    gt = S.synthetic_ground_truth_depth(b["image"], 1234).to(dev)
    tr.step(image, normal, b["sparse_depth"].to(dev), gt)
    torch.cuda.synchronize()
    for sig, n in seen.items():
        u = table.get(sig + "#us")
        if u and u[0]:
            total["plan"] += n * u[0]
            total["best"] += n * u[1]
    print("conv launches per step: %d over %d signatures; plan %.1f ms -> tuned %.1f ms" % (sum(seen.values()), len(seen), total["plan"] / 1e3, total["best"] / 1e3))
    out = {k: v for k, v in sorted(table.items()) if not k.endswith("#us")}
    with open(a.out, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote %d signatures to %s" % (len(out), a.out))


if __name__ == "__main__":
    main()
