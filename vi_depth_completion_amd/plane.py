"""Plane block + sparse-depth enrichment: host orchestration over the libvidc.so plane kernels.

Replaces `extract_plane_images_from_normal_image` (main.py:130-190) and the enrichment loop of
`RunDepthCompletion._call_cnn` (main.py:277-297).  The host keeps exactly what the reference keeps on the host --
the numpy legacy RNG draws (np.random.permutation / np.random.randint, in the reference's order) and the plane-id
maps, which arrive as numpy arrays from the plane detector (predictor.py:143-150) -- and hands index arrays to the
device.  Everything else (RANSAC, offset, projection, override, scatter) is one launch per stage for the whole batch.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib as L

NUM_HYPOTHESES = 300   # main.py:38,68


class _LegacyStream:
    """numpy's legacy MT19937 stream of `rng` (np.random itself or a RandomState) opened for native replay: permutation prefixes are
    drawn by vidc_host_mt19937_permutation_prefix on a copy of the state, `commit()` writes the advanced state back.  None when `rng`
    is something else (a Generator, a stub): the caller then uses rng.permutation."""

    @staticmethod
    def open(rng):
        try:
            st = rng.get_state()
        except (AttributeError, TypeError):
            return None
        if not isinstance(st, tuple) or st[0] != "MT19937":
            return None
        s = _LegacyStream()
        s.rng, s.key, s.pos, s.tail = rng, np.ascontiguousarray(st[1], dtype=np.uint32).copy(), C.c_int32(int(st[2])), st[3:]
        s.scratch = None
        return s

    def permutation_prefix(self, n, k):
        k = min(k, n)
        out = np.empty(k, dtype=np.int32)
        if self.scratch is None or self.scratch.shape[0] < n:
            self.scratch = np.empty(max(n, 1), dtype=np.int32)
        L.check(L.lib().vidc_host_mt19937_permutation_prefix(self.key.ctypes.data, C.addressof(self.pos), n, k, out.ctypes.data, self.scratch.ctypes.data),
                "permutation_prefix")
        return out

    def commit(self):
        self.rng.set_state(("MT19937", self.key, int(self.pos.value)) + tuple(self.tail))


def plane_groups(id_map):
    """Pixels of an id map grouped by plane: (classes > 0 ascending, like torch.unique; for each the flat pixel indices ascending, like
    np.flatnonzero(flat == cls)).  One stable sort instead of one pass over the map per plane."""
    flat = np.asarray(id_map).reshape(-1)
    if flat.dtype != np.uint8:
        flat = flat.astype(np.int64)
    counts = np.bincount(flat)
    if counts.shape[0] == 1:
        return []                          # only background: the reference returns its inputs unchanged (main.py:135-137)
    order = np.argsort(flat, kind="stable")
    starts = np.concatenate(([0], np.cumsum(counts)))
    return [(int(cls), order[starts[cls]:starts[cls + 1]]) for cls in np.flatnonzero(counts) if cls != 0]


def draw_normal_hypotheses(id_maps, rng=np.random, dense=None, groups=None):
    """For every image and every plane id > 0 (ascending, like torch.unique), draw the hypothesis rows exactly as
    mean_normal_ranasc does (main.py:43) and convert them to flat pixel indices.
    `dense`: {slot index: n_pts} -- planes known (from a first device pass) to be accepted with n_pts > 300 sparse points on
    them: plane_offset_ransac then draws np.random.permutation(np.r_[0:n_pts])[0:300] (main.py:78) right after the plane's normal
    hypotheses and before the next plane's, which is where it is drawn here.
    `groups`: plane_groups() of every map, if the caller has them already (unchanged id maps: PlaneBlock keeps them).
    The permutations run natively on the generator's MT19937 state when `rng` is numpy's legacy generator (same draws, same state
    afterwards: tests/test_abi.py), through rng.permutation otherwise.
    Returns (slots int32 [n,4] = (b, cls, hyp_offset, n_hyp), hyp_pix int32, {slot index: offset hypothesis ranks int32 [300]})."""
    slots, hyp, dense_hyp = [], [], {}
    off = 0
    stream = _LegacyStream.open(rng)

    def prefix(n):
        if stream is not None:
            return stream.permutation_prefix(n, NUM_HYPOTHESES)
        return rng.permutation(np.r_[0:n])[0:min(NUM_HYPOTHESES, n)]

    for b, m in enumerate(id_maps):
        for cls, pix in (groups[b] if groups is not None else plane_groups(m)):
            idx = prefix(pix.shape[0])
            hyp.append(pix[idx].astype(np.int32))
            k = len(slots)
            slots.append((b, cls, off, len(idx)))
            off += len(idx)
            if dense and k in dense:
                dense_hyp[k] = prefix(dense[k]).astype(np.int32)
    if stream is not None:
        stream.commit()
    slots = np.asarray(slots, dtype=np.int32).reshape(-1, 4)
    hyp = np.concatenate(hyp).astype(np.int32) if hyp else np.zeros(0, dtype=np.int32)
    return slots, hyp, dense_hyp


def draw_enrichment(nnz, goal, rng=np.random):
    """main.py:290-292 for every image: np.unique(np.random.randint(0, nnz, size=min(goal, nnz)))."""
    subs, offs = [], [0]
    for n in nnz:
        n = int(n)
        k = min(goal, n)
        sub = np.unique(rng.randint(0, n, size=k)) if n > 0 else np.zeros(0, dtype=np.int64)
        subs.append(sub.astype(np.int32))
        offs.append(offs[-1] + len(sub))
    sub = np.concatenate(subs) if subs else np.zeros(0, dtype=np.int32)
    return sub, np.asarray(offs, dtype=np.int32)


class _Upload:
    """One host->device transfer per call through a reused pinned staging buffer (a pageable source would make the runtime stage and
    synchronise on every call).  The staging buffer is rewritten only after the previous transfer out of it has completed."""

    def __init__(self):
        self.host = self.dev = self.event = None

    def __call__(self, arrays, device):
        n = sum(a.size for a in arrays)
        if self.host is None or self.host.numel() < n or self.dev.device != device:
            cap = max(4096, 2 * n)
            self.host = torch.empty(cap, dtype=torch.int32, pin_memory=True)
            self.dev = torch.empty(cap, dtype=torch.int32, device=device)
            self.event = torch.cuda.Event()
        else:
            self.event.synchronize()
        hv = self.host.numpy()
        offs, o = [], 0
        for a in arrays:
            hv[o:o + a.size] = a.reshape(-1)
            offs.append(self.dev.data_ptr() + 4 * o)
            o += a.size
        self.dev[:n].copy_(self.host[:n], non_blocking=True)
        self.event.record()
        return offs


class PlaneBlock:
    """Device buffers persist per (device, B, HW) and grow with the number of plane slots; id maps are re-uploaded only when they
    change.  `plane_depth()` and `enrich()` of one batch go together: `enrich` is where the host reads the device's counts (the one
    synchronisation of the path) and where a plane with more than 300 sparse points is resolved (main.py:75-78)."""

    def __init__(self):
        self._ids_dev = self._ids_host = self._ids_event = None
        self._groups = self._groups_src = None
        self.last_records = None
        self._bufs = {}
        self._up1, self._up2 = _Upload(), _Upload()
        self._ctx = None

    def _stacked_ids(self, id_maps):
        """The batch's id maps as ONE (B,H,W) uint8 array, and whether its content differs from the previous batch's (one stack + one
        comparison of 77 KB per image and batch; `_groups_of` and `_upload_ids` both key on it).  None when a map is not uint8."""
        if not all(np.asarray(m).dtype == np.uint8 for m in id_maps):
            return None, True
        arr = np.stack([np.asarray(m, dtype=np.uint8) for m in id_maps])
        prev = self._groups_src
        same = prev is not None and prev.shape == arr.shape and np.array_equal(prev, arr)
        return arr, not same

    def _groups_of(self, id_maps, stacked=None):
        """plane_groups() of the batch's id maps, recomputed only when their content changed (a fixed mask or a static scene)."""
        arr, changed = stacked if stacked is not None else self._stacked_ids(id_maps)
        if arr is None:
            return [plane_groups(m) for m in id_maps]
        if self._groups is None or changed:
            self._groups_src, self._groups = arr, [plane_groups(m) for m in arr]
        return self._groups

    def _upload_ids(self, id_maps, device, stacked=None):
        """The (B,H,W) uint8 id maps on the device.  Uploaded only when their CONTENT changed since the last batch (a fixed plane mask, or
        a static scene, costs one 77 KB comparison per batch instead of a host->device copy that waits for the stream), through a pinned
        staging buffer otherwise."""
        arr = stacked[0] if (stacked is not None and stacked[0] is not None) else np.stack([np.asarray(m, dtype=np.uint8) for m in id_maps])
        if self._ids_dev is None or self._ids_dev.device != device or self._ids_host.shape != arr.shape or \
                (arr is not self._ids_host and not np.array_equal(self._ids_host, arr)):
            if self._ids_dev is None or self._ids_dev.device != device or self._ids_host.shape != arr.shape:
                self._ids_pinned = torch.empty(arr.shape, dtype=torch.uint8, pin_memory=True)
                self._ids_dev = torch.empty(arr.shape, dtype=torch.uint8, device=device)
                self._ids_event = None
            if self._ids_event is not None:
                self._ids_event.synchronize()               # the previous upload out of the staging buffer has completed
            self._ids_host = arr
            self._ids_pinned.numpy()[...] = arr
            self._ids_dev.copy_(self._ids_pinned, non_blocking=True)
            self._ids_event = torch.cuda.Event()
            self._ids_event.record()
        return self._ids_dev

    def _buffers(self, dev, B, HW, n_slots):
        key = (dev, B, HW)
        b = self._bufs.get(key)
        if b is None or b["cap"] < n_slots:
            cap = max(8, n_slots, 2 * (b["cap"] if b else 0))
            lib = L.lib()
            b = {"cap": cap,
                 "mask": torch.empty((cap, HW), dtype=torch.uint8, device=dev),
                 "counts": torch.empty((cap, L.MAX_HYP), dtype=torch.int32, device=dev),
                 "rec": torch.zeros((cap, L.PLANE_RECORD), dtype=torch.float32, device=dev),
                 "scratch": torch.empty(lib.vidc_plane_scratch_bytes(cap, B, HW), dtype=torch.uint8, device=dev),
                 "di": torch.empty((B, HW), dtype=torch.float32, device=dev),
                 "info": torch.empty(lib.vidc_plane_info_count(B, HW), dtype=torch.int32, device=dev),
                 "enriched": torch.empty((B, HW), dtype=torch.float32, device=dev)}
            self._bufs[key] = b
        return b

    def plane_depth(self, normals, id_maps, sparse_depth, homo, rng=np.random):
        """normals (B,3,H,W), sparse_depth (B,1,H,W), homo (B,H,W,3): GPU fp32.  id_maps: B numpy (H,W) integer maps.
        Returns (di (B,1,H,W), info): `info` is the device int32 buffer vidc_plane_finalize fills (per-chunk candidate
        counts + the number of flagged slots); `enrich()` reads it back -- the single device->host sync of the path.  `di` is
        a buffer of this object: valid until the next `plane_depth` call."""
        if not normals.is_cuda:
            raise RuntimeError("PlaneBlock runs on the GPU only (no CPU fallback)")
        self._ctx = {"normals": normals.contiguous(), "ids": id_maps, "ds": sparse_depth.contiguous(), "homo": homo.contiguous(),
                     "rng": rng, "state0": (rng.get_state() if hasattr(rng, "get_state") else None), "dense": {}}      # (np.random.Generator
        #                                                 and stub generators have no legacy state: fine until a plane needs the rewind below)
        return self._launch(self._ctx)

    def _launch(self, ctx):
        lib, st = L.lib(), L.current_stream()
        normals, homo, rng = ctx["normals"], ctx["homo"], ctx["rng"]
        B, _, H, W = normals.shape
        HW = H * W
        dev = normals.device
        ds = ctx["ds"].view(B, HW)
        stacked = self._stacked_ids(ctx["ids"])
        if stacked[0] is not None and not stacked[1]:
            stacked = (self._groups_src, False)      # unchanged content: the array object both caches already hold
        slots, hyp, dense_hyp = draw_normal_hypotheses(ctx["ids"], rng, ctx["dense"], groups=self._groups_of(ctx["ids"], stacked))
        n_slots = slots.shape[0]
        bufs = self._buffers(dev, B, HW, n_slots)
        di = bufs["di"]
        rec = None
        info = bufs["info"]
        ids = slots_p = hyp_p = dh_p = dn_p = None
        dots = None
        if n_slots > 0:
            ids = self._upload_ids(ctx["ids"], dev, stacked)
            arrays = [slots, hyp]
            if dense_hyp:
                dh = np.zeros((n_slots, L.MAX_HYP), dtype=np.int32)
                dn = np.zeros(n_slots, dtype=np.int32)
                for k, v in dense_hyp.items():
                    dh[k], dn[k] = v, ctx["dense"][k]
                arrays += [dh, dn]
            ptrs = self._up1(arrays, dev)
            slots_p, hyp_p = ptrs[0], ptrs[1]
            rec = bufs["rec"][:n_slots]
            if dense_hyp:
                if bufs.get("dense_dots") is None or bufs["dense_dots"].shape[0] < n_slots:
                    bufs["dense_dots"] = torch.empty((bufs["cap"], HW), dtype=torch.float32, device=dev)
                dh_p, dn_p, dots = ptrs[2], ptrs[3], bufs["dense_dots"]
        # ONE native call for the whole block (copy of the sparse depth, RANSAC, offset, projection, finalize): its launch-bound kernels
        # go out back to back instead of one trip through Python apiece (the stream modes enqueue them between two graph segments)
        L.check(lib.vidc_plane_block(L.ptr(normals), L.ptr(ids), slots_p, n_slots, hyp_p, B, HW, L.ptr(homo), L.ptr(ds), L.ptr(bufs["mask"]),
                                     L.ptr(bufs["counts"]), L.ptr(bufs["scratch"]), L.ptr(rec), dh_p, dn_p, L.ptr(dots), L.ptr(di), L.ptr(info), st),
                "plane_block")
        self.last_records, self.last_slots, self.last_mask = rec, slots, (bufs["mask"][:n_slots] if n_slots > 0 else None)
        return di.view(B, 1, H, W), info

    def read_info_async(self, info):
        """Starts the device->host read of `info` on the current stream; returns (pinned host tensor, event).  Work enqueued
        after this call does not delay the read: `enrich(..., info_host=...)` waits for the event only."""
        if getattr(self, "_info_pinned", None) is None or self._info_pinned.numel() != info.numel():
            self._info_pinned = torch.empty(info.numel(), dtype=torch.int32, pin_memory=True)
            self._info_event = torch.cuda.Event()
        self._info_pinned.copy_(info, non_blocking=True)
        self._info_event.record()
        return self._info_pinned, self._info_event

    def _resolve_dense(self, info_h):
        """Slow path, taken only when a plane carries more than 300 sparse-depth points (main.py:75-78).  The reference draws the
        plane-offset hypotheses of such a plane with np.random.permutation between the normal-hypothesis draws of this plane and of
        the next one, so every later draw moves: rewind the generator to where this batch started, replay the draws with the extra
        permutation(s) in place and run the plane kernels again -- one plane (in draw order) and one extra host sync per pass,
        until no slot is flagged.  The draws before the first flagged plane are reproduced bit for bit."""
        ctx = self._ctx
        di = info = None
        if ctx["state0"] is None:
            raise RuntimeError("plane block: a plane carries more than 300 sparse-depth points (main.py:75-78) and the draws of this batch have "
                               "to be replayed, which needs a generator with get_state()/set_state() (np.random or np.random.RandomState)")
        while info_h[-1] != 0:
            rec = self.last_records.cpu().numpy()
            flagged = [k for k in np.flatnonzero(rec[:, 10] < 0) if k not in ctx["dense"]]
            if not flagged:
                raise RuntimeError("plane block: a slot stays flagged after its offset hypotheses were supplied")
            k = int(flagged[0])
            ctx["dense"] = {j: n for j, n in ctx["dense"].items() if j < k}       # later planes are re-drawn: their counts may change
            ctx["dense"][k] = int(rec[k, 7])
            ctx["rng"].set_state(ctx["state0"])
            di, info = self._launch(ctx)
            info_h = info.cpu().numpy()
        return di, info_h

    def enrich(self, sparse_depth, di, info, goal, rng=np.random, info_host=None, out=None):
        """main.py:285-294.  One device->host read (the reference syncs on torch.nonzero here): per-chunk candidate counts
        + the number of planes that need the >300-point host permutation (main.py:78; resolved by `_resolve_dense`).
        `out`: (B,1,H,W) tensor to write the enriched depth into (every pixel is written); default: a buffer of this object,
        valid until the next `enrich` call."""
        lib, st = L.lib(), L.current_stream()
        B, _, H, W = sparse_depth.shape
        dev = sparse_depth.device
        if info_host is not None:
            pinned, ev = info_host
            ev.synchronize()
            info_h = pinned.numpy().copy()
        else:
            info_h = info.cpu().numpy()
        if info_h[-1] != 0:
            di, info_h = self._resolve_dense(info_h)
        chunks = info_h[:-1].reshape(B, -1)
        nnz_h = chunks.sum(axis=1)
        sub, offs = draw_enrichment(nnz_h, goal, rng)
        if out is None:
            out = self._buffers(dev, B, H * W, 0)["enriched"].view(B, 1, H, W)
        self.last_sub, self.last_nnz = (sub, offs), nnz_h
        base = (np.cumsum(chunks, axis=1) - chunks).astype(np.int32)
        p = self._up2([sub if len(sub) else np.zeros(1, dtype=np.int32), offs, base], dev)
        L.check(lib.vidc_enrich_scatter_from(L.ptr(di), L.ptr(sparse_depth.contiguous()), p[0], p[1], p[2], B, H * W, L.ptr(out), st),
                "enrich_scatter")
        return out
