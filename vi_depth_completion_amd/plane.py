"""Plane block + sparse-depth enrichment: host orchestration over the libvidc.so plane kernels.

Replaces `extract_plane_images_from_normal_image` (main.py:130-190) and the enrichment loop of
`RunDepthCompletion._call_cnn` (main.py:277-297).  The host keeps exactly what the reference keeps on the host --
the numpy legacy RNG draws (np.random.permutation / np.random.randint, in the reference's order) and the plane-id
maps, which arrive as numpy arrays from the plane detector (predictor.py:143-150) -- and hands index arrays to the
device.  Everything else (RANSAC, offset, projection, override, scatter) is one launch per stage for the whole batch.
"""
import numpy as np
import torch

from . import _lib as L

NUM_HYPOTHESES = 300   # main.py:38,68


def draw_normal_hypotheses(id_maps, rng=np.random):
    """For every image and every plane id > 0 (ascending, like torch.unique), draw the hypothesis rows exactly as
    mean_normal_ranasc does (main.py:43) and convert them to flat pixel indices.
    Returns (slots int32 [n,4] = (b, cls, hyp_offset, n_hyp), hyp_pix int32)."""
    slots, hyp = [], []
    off = 0
    for b, m in enumerate(id_maps):
        flat = np.asarray(m).reshape(-1)
        if flat.dtype != np.uint8:
            flat = flat.astype(np.int64)
        counts = np.bincount(flat)
        if counts.shape[0] == 1:
            continue                      # only background: the reference returns its inputs unchanged (main.py:135-137)
        # one stable sort groups the pixels by plane id with ascending pixel index inside every plane (= np.flatnonzero(flat == cls)
        # for every cls, without one pass over the map per plane: the detector finds 5-9 planes per image)
        order = np.argsort(flat, kind="stable")
        starts = np.concatenate(([0], np.cumsum(counts)))
        for cls in np.flatnonzero(counts):
            if cls == 0:
                continue
            pix = order[starts[cls]:starts[cls + 1]]
            n = pix.shape[0]
            idx = rng.permutation(np.r_[0:n])[0:min(NUM_HYPOTHESES, n)]
            hyp.append(pix[idx].astype(np.int32))
            slots.append((b, int(cls), off, len(idx)))
            off += len(idx)
    slots = np.asarray(slots, dtype=np.int32).reshape(-1, 4)
    hyp = np.concatenate(hyp).astype(np.int32) if hyp else np.zeros(0, dtype=np.int32)
    return slots, hyp


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


class PlaneBlock:
    """Device buffers are cached per (B, HW, n_slots); id maps are re-uploaded only when they change."""

    def __init__(self):
        self._ids_key = None
        self._ids_dev = None
        self.last_records = None

    def _upload_ids(self, id_maps, device):
        key = tuple(id(m) for m in id_maps)
        arr = np.stack([np.asarray(m, dtype=np.uint8) for m in id_maps])
        if self._ids_key != key or self._ids_dev is None or self._ids_dev.device != device or not np.array_equal(self._ids_host, arr):
            self._ids_host = arr
            self._ids_dev = torch.from_numpy(arr).to(device)
            self._ids_key = key
        return self._ids_dev

    def plane_depth(self, normals, id_maps, sparse_depth, homo, rng=np.random):
        """normals (B,3,H,W), sparse_depth (B,1,H,W), homo (B,H,W,3): GPU fp32.  id_maps: B numpy (H,W) integer maps.
        Returns (di (B,1,H,W), info): `info` is the device int32 buffer vidc_plane_finalize fills (per-chunk candidate
        counts + the error flag); `enrich()` reads it back -- the single device->host sync of the path."""
        if not normals.is_cuda:
            raise RuntimeError("PlaneBlock runs on the GPU only (no CPU fallback)")
        lib, st = L.lib(), L.current_stream()
        B, _, H, W = normals.shape
        HW = H * W
        dev = normals.device
        normals, homo = normals.contiguous(), homo.contiguous()
        ds = sparse_depth.contiguous().view(B, HW)
        slots, hyp = draw_normal_hypotheses(id_maps, rng)
        di = ds.clone()
        n_slots = slots.shape[0]
        rec = None
        if n_slots > 0:
            ids = self._upload_ids(id_maps, dev)
            host = np.concatenate([slots.reshape(-1), hyp]).astype(np.int32)          # one upload: slots then hypotheses
            host_d = torch.from_numpy(host).to(dev, non_blocking=True)
            slots_p, hyp_p = host_d.data_ptr(), host_d.data_ptr() + 4 * slots.size
            mask = torch.empty((n_slots, HW), dtype=torch.uint8, device=dev)
            counts = torch.empty((n_slots, L.MAX_HYP), dtype=torch.int32, device=dev)
            rec = torch.empty((n_slots, L.PLANE_RECORD), dtype=torch.float32, device=dev)
            scratch = torch.empty(lib.vidc_plane_scratch_bytes(n_slots, B, HW), dtype=torch.uint8, device=dev)
            L.check(lib.vidc_plane_ransac_normal(L.ptr(normals), L.ptr(ids), slots_p, n_slots, hyp_p, HW, L.ptr(mask), L.ptr(counts),
                                                 L.ptr(scratch), st), "plane_ransac_normal")
            L.check(lib.vidc_plane_offset(L.ptr(homo), L.ptr(ds), slots_p, n_slots, B, L.ptr(mask), L.ptr(counts), HW, L.ptr(scratch),
                                          L.ptr(rec), st), "plane_offset")
            L.check(lib.vidc_plane_project_depth(L.ptr(homo), slots_p, n_slots, L.ptr(mask), HW, L.ptr(scratch), L.ptr(rec), L.ptr(di), st),
                    "plane_project_depth")
            self._keep = (host_d, scratch, counts)
        self.last_records, self.last_slots, self.last_mask = rec, slots, (mask if n_slots > 0 else None)
        info = torch.empty(lib.vidc_plane_info_count(B, HW), dtype=torch.int32, device=dev)
        L.check(lib.vidc_plane_finalize(L.ptr(ds), L.ptr(di), B, HW, L.ptr(rec), n_slots, L.ptr(info), st), "plane_finalize")
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

    def enrich(self, sparse_depth, di, info, goal, rng=np.random, info_host=None):
        """main.py:285-294.  One device->host read (the reference syncs on torch.nonzero here): per-chunk candidate counts
        + the flag for planes that would need the >300-point host permutation (main.py:78), which is not done on device."""
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
            raise NotImplementedError("a plane has more than %d sparse depth points; the subsampled plane-offset RANSAC "
                                      "(main.py:78) is not implemented on device" % L.MAX_HYP)
        chunks = info_h[:-1].reshape(B, -1)
        nnz_h = chunks.sum(axis=1)
        sub, offs = draw_enrichment(nnz_h, goal, rng)
        out = sparse_depth.clone()
        self.last_sub, self.last_nnz = (sub, offs), nnz_h
        if len(sub):
            base = (np.cumsum(chunks, axis=1) - chunks).astype(np.int32)
            host = np.concatenate([sub, offs, base.reshape(-1)]).astype(np.int32)      # one upload
            host_d = torch.from_numpy(host).to(dev, non_blocking=True)
            p0 = host_d.data_ptr()
            L.check(lib.vidc_enrich_scatter(L.ptr(di), p0, p0 + 4 * len(sub), p0 + 4 * (len(sub) + len(offs)), B, H * W, L.ptr(out), st),
                    "enrich_scatter")
            self._keep2 = host_d
        return out
