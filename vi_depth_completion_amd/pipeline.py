"""The per-batch hot path behind the reference's operator API.

`DepthCompletionPipeline._call_cnn(input_batch)` is a drop-in for `RunDepthCompletion._call_cnn` (main.py:261-298):
same input-batch dictionary (dataset.py:515-520), same output (B,1,H,W) depth tensor, same attribute names
(`cnn`, `surface_normal_cnn`, `plane_masks_extraction`, `args.enriched_samples`) and the same checkpoint loaders
(network_run.py:319-323, main.py:256-259).  See INTEGRATION.md for how the unchanged main.py/network_run.py bind to it.
"""
import argparse
import os
import weakref

import numpy as np
import torch

from .networks.depth_completion import ModifiedFPN
from .networks.surface_normal import SurfaceNormalPrediction
from .plane import PlaneBlock


class _Stager:
    """Host-resident batch tensors (the reference's DataLoader hands out CPU batches) -> the device, without blocking the host: a
    copy out of pageable memory makes the host wait until the device has caught up with everything queued before it (a whole tick in
    the stream modes), and pinned memory from torch's caching allocator is not reusable while a copy out of it is still queued, so
    every frame would allocate fresh pinned memory (milliseconds per call; measured 38 frames/s).  Hence a small ring of persistent
    pinned buffers per tensor name, each rewritten only after the copy out of it has completed."""

    def __init__(self, depth=2):
        self.depth, self.rings, self.turn = depth, {}, {}

    def __call__(self, name, t, dev):
        if t.is_cuda:
            return t
        key = (name, tuple(t.shape), t.dtype)
        ring = self.rings.setdefault(key, [])
        i = self.turn.get(key, 0)
        self.turn[key] = i + 1
        if len(ring) < self.depth:
            ring.append([torch.empty(t.shape, dtype=t.dtype, pin_memory=True), None])
        slot = ring[i % len(ring)] if len(ring) == self.depth else ring[-1]
        if slot[1] is not None:
            slot[1].synchronize()
        # (numpy, not Tensor.copy_: a torch CPU copy of a megabyte goes through the OpenMP pool, whose workers then spin on every core of
        #  the host and slow the launching thread down tenfold -- measured 47 frames/s)
        np.copyto(slot[0].numpy(), t.detach().numpy())
        d = slot[0].to(dev, non_blocking=True)
        slot[1] = torch.cuda.Event()
        slot[1].record()
        return d


class FixedPlaneMask:
    """Plane-mask provider for the perf configuration ("plane mask fixed", BASELINE.json configs[1]); has the
    `run_on_tensor(image) -> uint8 (H,W) id map` interface of COCODemo (plane_mask_detection/demo/predictor.py:143-150)."""

    def __init__(self, id_map):
        self.id_map = np.ascontiguousarray(id_map, dtype=np.uint8)

    def run_on_tensor(self, image):
        return self.id_map


def build_frame_program(sn, dc, B, H, W, device, dry_run=False, weights=None):
    """ONE planned program for a pipeline tick: the surface-normal network of frame i+1 and the depth-completion
    network of frame i.  Their four ResNet-101 pyramids have identical layer shapes, so each pyramid layer is one
    grouped launch over (sn, dc.rgb, dc.normal, dc.depth) instead of a 1-group and a 3-group launch -- at batch 1
    these ~210 launches are latency-bound, not throughput-bound, so the 1-group ones are almost free riders.
    Segment 0 = warp + pyramids + surface-normal decoder + inverse warp; segment 1 = depth-completion decoder (the
    host enqueues the plane block of frame i+1 between them and reads its counts back while segment 1 runs)."""
    from . import engine
    from .engine import T
    from .networks.fpn_decoder import emit_decoder
    wp = sn.warp_2dof_alignment
    assert (wp.H, wp.W) == (H, W), "frame size %dx%d does not match the warp intrinsics (%dx%d)" % (W, H, wp.W, wp.H)
    ws = weights if weights is not None else engine.JointWeightStore({"sn": sn, "dc": dc})     # `weights`: reuse packed weights (tools)
    prog = engine.Program(ws, device, B)
    x = prog.input_nchw("sn_image", 3, H, W)
    g = prog.input_raw("gravity", B * 3)
    a = prog.input_raw("aligned", B * 3)
    kinv = prog.input_raw("kinv", 9)
    img = prog.input_nchw("dc_image", 3, H, W)
    nrm = prog.input_nchw("dc_normal", 3, H, W)
    dep = prog.input_nchw("dc_depth", 1, H, W)
    params = prog.warp_params(g, a, wp, kinv)
    xw = prog.warp_fwd(x, params, wp, wp.align_corners)
    levels = sn.resnet_pyramids.emit(prog, [xw, img, nrm, dep],
                                     engine.K(("sn/resnet_pyramids.", "dc/resnet_rgb.", "dc/resnet_normal.", "dc/resnet_depth.")))
    if prog.mode == "mixed":
        for t in levels:             # ONE split image per level, written by the producing conv's epilogue; the next pyramid
            prog.split(t)            # stage and both decoders read (channel slices of) it
    sn_levels = [T(t.buf, t.B, t.H, t.W, t.C, 1, t.ld, t.ch_off) for t in levels]
    dc_levels = [T(t.buf, t.B, t.H, t.W, 3 * t.C, 1, t.ld, t.ch_off + t.C) for t in levels]
    if sn.use_mask:                  # surface_normal.py:150-162
        zs = prog.mask_scale(emit_decoder(prog, sn, [prog.mask_scale(t, xw) for t in sn_levels], "sn/"), xw)
    else:
        zs = emit_decoder(prog, sn, sn_levels, "sn/")
    h = prog.conv(zs, "sn/feature_concat.0", relu=True, padding=1)
    y, _low = prog.head(h, "sn/feature_concat.2", 0, (H, W), relu=False)
    # The normals of frame t are written straight into the buffer the normal pyramid of the depth network reads them from one
    # tick later: its stem (earlier in this very segment, same stream) has consumed frame t-1's normals by then, so the
    # hand-over costs no copy.
    z = prog.warp_inv(y, params, wp, wp.align_corners, normalize=True, out=nrm)
    prog.mark_output("normals", z)
    prog.cut()
    zd = emit_decoder(prog, dc, dc_levels, "dc/")
    h = prog.conv(zd, "dc/feature_concat.0", relu=True, padding=1)
    yd, _low = prog.head(h, "dc/feature_concat.2", 1, (H, W), relu=True)
    prog.mark_output("depth", yd)
    prog.finalize(dry_run)
    if not dry_run:
        prog.storage[kinv.buf][:9].copy_(wp.kinv(device))
        import os
        if os.environ.get("VIDC_TICK_VARIANTS", "1") == "1":
            # first tick of a stream: only the surface-normal side of segment 0 (pyramid group 0 + decoder + warps); drain tick: only the
            # depth-completion pyramids (groups 1..3) -- engine.Program.group_variant: same buffers, weights, tiles, bit-identical results
            prog.group_variant("head", 0, 0, 1, 4, keep_ungrouped=True)
            prog.group_variant("tail", 0, 1, 4, 4, keep_ungrouped=False)
    return prog


class DepthCompletionPipeline:
    def __init__(self, enriched_samples=200, fc_img=(202.0, 202.0), cc_img=(0.5 * 319.87654, 0.5 * 239.87603),
                 align_corners=False, device="cuda", network_class_creator=ModifiedFPN, rng=np.random, use_gravity=True):
        if not torch.cuda.is_available():
            raise RuntimeError("DepthCompletionPipeline needs a GPU: the HIP path has no CPU fallback")
        self.args = argparse.Namespace(enriched_samples=enriched_samples)
        self.device = torch.device(device)
        reserve_lane_streams(self.device, 3)         # (process-wide, idempotent: the stream mode's lanes take their hardware queues as early as possible)
        self.cnn = network_class_creator().to(self.device)                                   # network_run.py:422-424
        self.use_gravity = bool(use_gravity)
        if self.use_gravity:
            self.surface_normal_cnn = SurfaceNormalPrediction(fc_img=np.asarray(fc_img, dtype=np.float64),
                                                              cc_img=np.asarray(cc_img, dtype=np.float64),
                                                              align_corners=align_corners).to(self.device)   # main.py:243
        else:
            from .networks.surface_normal_dorn import SurfaceNormalDORN
            self.surface_normal_cnn = SurfaceNormalDORN().to(self.device)                                    # main.py:245
        self.plane_masks_extraction = None
        self.planes = PlaneBlock()
        self._stage = _Stager()
        self.rng = rng
        self.eval_mode()

    # ---- the reference's harness methods ------------------------------------------------------------------------
    def eval_mode(self):
        self.surface_normal_cnn.eval()
        self.cnn.eval()

    def load_network_from_file(self, filename):
        state = self.cnn.state_dict()
        state.update(torch.load(filename, map_location=self.device))
        self.cnn.load_state_dict(state)

    def load_surface_normal_network_from_file(self, checkpoint):
        state = self.surface_normal_cnn.state_dict()
        state.update(torch.load(checkpoint, map_location=self.device))
        self.surface_normal_cnn.load_state_dict(state)

    def load_state_dicts(self, sn_state, dc_state):
        for m, sd in ((self.surface_normal_cnn, sn_state), (self.cnn, dc_state)):
            state = m.state_dict()
            state.update(sd)
            m.load_state_dict(state)

    # ---- plane masks ---------------------------------------------------------------------------------------------------
    def _masks_begin(self, rgb, key=None):
        """Starts the plane-instance maps of a batch.  A device-side extractor (plane_mask.PlaneMaskDetector: `run_on_batch`) is
        enqueued on the current stream together with ONE asynchronous device->host copy of the (B,H,W) uint8 ids; anything else is
        the reference's per-sample `run_on_tensor` call (main.py:273), resolved in `_masks_end`.
        `key`: names the pinned host buffer the ids land in (one per lane and batch slot in the stream modes, where the next
        extraction is enqueued before the host has read this one).  Extractions on different streams are ordered one after the other
        on the device: the detector owns ONE set of buffers and one captured graph."""
        ex = self.plane_masks_extraction
        if not hasattr(ex, "run_on_batch"):
            return None
        # (on the current stream: running the detector on a side stream next to the networks measured 224 vs 235 frames/s in round 1 and,
        #  with two lanes and the deferred enrichment wait of round 2, 272 vs 302 at batch 1 and 444 vs 568 at batch 8)
        prev = getattr(self, "_masks_done", None)
        if prev is not None:
            torch.cuda.current_stream().wait_event(prev)
        ids = ex.run_on_batch(rgb)
        hosts = self.__dict__.setdefault("_ids_hosts", {})
        host = hosts.get(key)
        if host is None or host.shape != ids.shape:
            host = hosts[key] = torch.empty(ids.shape, dtype=torch.uint8, pin_memory=True)
        host.copy_(ids, non_blocking=True)
        self._ids_host = host
        ev = torch.cuda.Event()
        ev.record()
        self._masks_done = ev
        return ev, host

    def _masks_end(self, handle, images, H, W):
        if handle is None:
            return [np.asarray(self.plane_masks_extraction.run_on_tensor(images[i])).reshape(H, W) for i in range(images.shape[0])]
        ev, host = handle
        ev.synchronize()
        return [m.copy() for m in host.numpy()]

    # ---- the hot path ------------------------------------------------------------------------------------------------
    def _stage1(self, input_batch, slot=0, planes=None, rng=None):
        """warp + surface-normal net + plane block, enqueued on the current stream (main.py:262-283)."""
        dev = self.device
        planes = planes or self.planes
        ds = self._stage("sparse_depth", input_batch["sparse_depth"], dev)
        rgb = self._stage("image", input_batch["image"], dev)
        mh = self._masks_begin(rgb) if self.args.enriched_samples != 0 else None      # the id maps travel to the host under the normal net
        if self.use_gravity:
            normals = self.surface_normal_cnn.enqueue(rgb, self._stage("gravity", input_batch["gravity"], dev), self._stage("aligned_direction", input_batch["aligned_direction"], dev), slot)
        else:
            normals = self.surface_normal_cnn(rgb)                                                           # main.py:270-271
        rng = rng if rng is not None else self.rng
        st = {"ds": ds, "rgb": rgb, "normals": normals, "di": None, "nnz": None, "rng": rng}
        if self.args.enriched_samples != 0:
            homo = self._stage("homogeneous_coordinates", input_batch["homogeneous_coordinates"], dev)
            masks = self._masks_end(mh, input_batch["image"], ds.shape[-2], ds.shape[-1])
            st["di"], st["nnz"] = planes.plane_depth(normals, masks, ds, homo, rng=rng)
        return st

    def _stage2(self, st, slot=0, planes=None):
        """enrichment (one small device->host read of the candidate counts) + depth-completion net (main.py:285-297)."""
        planes = planes or self.planes
        depth_in = st["ds"]
        if st["di"] is not None:
            depth_in = planes.enrich(st["ds"], st["di"], st["nnz"], self.args.enriched_samples, rng=st["rng"])
            st["enriched"] = depth_in          # (buffers of `planes`: valid until its next batch)
        return self.cnn.enqueue(st["rgb"], st["normals"], depth_in, slot)

    @torch.no_grad()
    def network_inputs(self, input_batch):
        """(image, predicted normals, enriched sparse depth): exactly what `_call_cnn` hands to `self.cnn` (main.py:262-296), as fresh
        tensors -- the inputs of a training iteration (`training.DepthCompletionTrainer.step`; network_run.py:236 calls `_call_cnn`,
        whose only trained part is that last network).  The surface-normal network runs in eval mode here."""
        st = self._stage1(input_batch)
        depth_in = st["ds"]
        if st["di"] is not None:
            depth_in = self.planes.enrich(st["ds"], st["di"], st["nnz"], self.args.enriched_samples, rng=st["rng"])
        return st["rgb"], st["normals"].clone(), depth_in.clone()

    @torch.no_grad()
    def _call_cnn(self, input_batch, taps=None):
        st = self._stage1(input_batch)
        out = self._stage2(st).clone()
        if taps is not None:
            taps["normals"] = st["normals"].clone()
            if st["di"] is not None:
                taps.update(plane_depth=st["di"].clone(), enriched=st["enriched"].clone(), records=self.planes.last_records.clone()
                            if self.planes.last_records is not None else None)
        return out

    # ---- software-pipelined throughput mode ------------------------------------------------------------------------
    def frame_program(self, B, H, W):
        key = (B, H, W, self.surface_normal_cnn._version, self.cnn._version, self.surface_normal_cnn.warp_2dof_alignment.align_corners)
        progs = self.__dict__.setdefault("_frame_progs", {})
        if key not in progs:
            for k in [k for k in progs if k[3:] != key[3:]]:      # programs of parameters / a warp convention that are gone
                del progs[k]
            progs[key] = build_frame_program(self.surface_normal_cnn, self.cnn, B, H, W, self.device)
        return progs[key]

    @torch.no_grad()
    def run_interleaved(self, batches, copy_outputs=True, lanes=None, frames_per_launch=None, frame_rng=None):
        """Throughput mode, software-pipelined over the items of a stream: a tick runs the surface-normal network + plane block of one
        group of items and the depth-completion network of the previous group as ONE program (build_frame_program).  Yields the depth
        map of every batch, in order.  Per item the arithmetic is that of `_call_cnn` (same kernels; the 4-group launches may use
        another tile than the 1- / 3-group ones, i.e. fp32 sums in a different order), and the RANSAC / enrichment draws come off
        `self.rng` in the order of back-to-back `_call_cnn` calls.  Scheduler: `_run_grouped`.

        lanes = L (default: VIDC_LANES, else 1): L such pipelines on L HIP streams, group p on lane p mod L, each with its own program
        buffers and plane-block scratch: the fixed cost of a small-layer launch (dispatch, first weight stage from HBM, split-K
        epilogue, drain) and the launch-bound plane kernels between a lane's segments are filled by the other lanes' launches.  The
        draws are issued strictly in item order from this one host thread, so every item's result is bit-identical for every L.

        frames_per_launch = F (default: VIDC_FRAMES_PER_LAUNCH, else 1): F consecutive items share every launch of a tick: the lane's
        frame program is recorded for batch F x B and items F*p .. F*p+F-1 occupy its batch slots, so a batch-1 stream runs its
        ResNet-101 layer-3 convolutions at M = 640 (F = 2) instead of 320 -- half the launches per frame, each above the per-launch
        floor that bounds them at batch 1.  The items stay what the API hands in (main.py:261-298 semantics per item: own
        gravity, own plane block, own draws); a stream whose length is not a multiple of F runs its tail through the SAME program with
        the unused slots holding stale frames, so an item's bits do not depend on whether it had a partner, on which slot it took, or
        on the shard it was part of (rows of an implicit GEMM do not interact; tile, split-K and K order are the launch's).

        frame_rng(i) -> generator: item i draws from its own generator instead of `self.rng`.

        Data movement: an item is copied into its batch slot when it is pulled from `batches` (image, gravity, alignment, sparse depth,
        homogeneous grid: the caller may refill its tensors as soon as the next item is requested); the normals and the enriched
        sparse depth are WRITTEN where the lane's next tick reads them (no copies).  Outputs come out in item order, a group at a
        time, one lane round after their inputs went in; the caller's current stream waits (on the device) for the tick that produced
        them.  copy_outputs=False hands out views of the lane's own output buffer: valid ONLY until the next item is requested."""
        import os
        if not self.use_gravity:
            yield from self._run_interleaved_two_programs(batches, copy_outputs)
            return
        n = int(lanes if lanes is not None else os.environ.get("VIDC_LANES", "1"))
        if n < 1:
            raise ValueError("run_interleaved: lanes must be >= 1")
        fpl = int(frames_per_launch if frames_per_launch is not None else os.environ.get("VIDC_FRAMES_PER_LAUNCH", "1"))
        if fpl < 1:
            raise ValueError("run_interleaved: frames_per_launch must be >= 1")
        yield from self._run_grouped(batches, copy_outputs, n, fpl, frame_rng)

    @torch.no_grad()
    def prepare_interleaved(self, sample_batch, lanes=None, frames_per_launch=None):
        """Builds, captures and uploads the frame programs of EVERY lane of `run_interleaved(lanes, frames_per_launch)` for batches shaped
        like `sample_batch` (normally that happens when a lane sees its first item), and -- if `sample_batch` is a whole batch -- runs it
        once through every batch slot of every lane (the slots' plane blocks allocate on first use; `self.rng` is not touched).  Set-up, not
        a step: a stream shorter than lanes x frames_per_launch items would otherwise leave a lane's program to be recorded, and its
        slots' buffers to be allocated, inside a later stream -- e.g. a timed one."""
        import os
        n = int(lanes if lanes is not None else os.environ.get("VIDC_LANES", "1"))
        fpl = int(frames_per_launch if frames_per_launch is not None else os.environ.get("VIDC_FRAMES_PER_LAUNCH", "1"))
        if not self.use_gravity:
            return
        rgb = sample_batch["image"]
        if not rgb.is_cuda:
            rgb = rgb.to(self.device)
        for k in range(n):
            lane = _GroupLane(self, k, fpl)
            with torch.cuda.stream(lane.stream):
                lane._prepare(rgb)
        torch.cuda.synchronize(self.device)
        if all(k in sample_batch for k in ("sparse_depth", "gravity", "aligned_direction", "homogeneous_coordinates")):
            # ... and one item through every batch slot of every lane, with a generator of its own: a slot's plane block allocates its device
            # buffers and its pinned host buffers when it sees its first item -- for the slots a short warm-up does not reach (5 items touch 5
            # of the 12 slots of 3 lanes x 4 items) that was inside the caller's timed region, several host allocations of ~0.5 ms each
            throwaway = np.random.RandomState(0)
            for _ in self._run_grouped(iter([sample_batch] * (n * fpl)), True, n, fpl, lambda i: throwaway):
                pass
            torch.cuda.synchronize(self.device)

    def _run_grouped(self, batches, copy_outputs, n_lanes, F, frame_rng):
        """run_interleaved with F items per launch (see there).  Group p = items F*p .. F*p+F-1 runs on lane p mod L.  Per lane and group
        the device work is: segment 0 [surface-normal side of group p + depth pyramids of the lane's previous group], the plane kernels
        of every item, segment 1 [depth decoder of the previous group], the enrichment scatters.  The host side is two interleaved
        sequences from this one thread:
          * the DRAW sequence, strictly in item order -- hypotheses(i), [wait for item i's candidate counts], enrichment(i),
            hypotheses(i+1), ... -- which is the order back-to-back `_call_cnn` calls consume the generator in;
          * the LAUNCH sequence, which runs ahead of it: segment 0 of group p+L is launched as soon as group p is enriched (it needs
            nothing else), i.e. before the draw sequence turns to group p+1 on the next lane -- so whenever the host waits for an item's
            counts, the other lane has a whole segment 0 queued.  Segment 1 of a visit (the previous group's decoder) is launched
            after the LAST item's plane kernels and before the wait for their counts: launched earlier (after the first item's, as until
            the end of round 4) the later items' plane kernels queued behind ~1.5 ms of decoder on the lane's stream and the host sat in
            their count waits -- mixed leg 856-874 -> 917-918 frames/s at 200 steps, fp32 395-398 -> 400-401 (VIDC_DECODER_AFTER_LAST=0:
            the old order; same programs, same bits)."""
        import os
        lanes = [_GroupLane(self, k, F) for k in range(n_lanes)]
        it = iter(batches)
        state = {"taken": 0}

        def start(lane):
            """Pulls up to F items, each one copied into its batch slot before the next is requested (a caller that refills ONE set of
            input tensors sees them consumed item by item), and launches segment 0 for them.  Returns the items' stream indices."""
            idx = []
            for j in range(F):
                b = next(it, None)
                if b is None:
                    break
                lane.put(j, b)
                idx.append(state["taken"])
                state["taken"] += 1
            if idx:
                lane.begin(len(idx))
            return idx

        groups = {}                              # group index -> stream indices of its items; filled as segment 0 of the group is launched
        # (the first segments of the L lanes start side by side: chaining them one behind the other so that lane 0's counts arrive
        #  earlier measured 350 (chain of 1) / 360 (chain of 2) against 359 frames/s at 20 steps, fp32, 3 lanes -- DESIGN 4.4)
        for p in range(n_lanes):
            idx = start(lanes[p])
            if not idx:
                break
            groups[p] = idx
        ready, nxt, p = {}, 0, 0                 # group index -> (outputs, event, n items, lane)

        def flush():
            nonlocal nxt
            while nxt in ready:
                out, ev, n, lane = ready.pop(nxt)
                nxt += 1
                torch.cuda.current_stream().wait_event(ev)        # device-side: readers on the caller's stream find the items complete
                if copy_outputs and out.is_cuda:
                    # the clone was allocated on the LANE's stream and is read on the caller's: without this the caching allocator hands the
                    # block back to the lane's pool when the caller drops the tensor and a later visit of the lane (hypothesis temporaries,
                    # the next clone) may overwrite it under a read the caller enqueued asynchronously
                    out.record_stream(torch.cuda.current_stream())
                B = out.shape[0] // F
                for j in range(n):
                    yield out[j * B:(j + 1) * B]
                if not copy_outputs:              # the lane's own buffer went out: its next decoder waits for the caller's reads of it --
                    lane.consumed = torch.cuda.Event()            # for THOSE only (a wait for the caller's whole stream would also wait for
                    lane.consumed.record()                        # the other lanes' outputs that stream has been told to wait for)

        while p in groups:
            lane = lanes[p % n_lanes]
            items = groups.pop(p)
            dec_at = (len(items) - 1) if os.environ.get("VIDC_DECODER_AFTER_LAST", "1") == "1" else 0
            for j, i in enumerate(items):
                rng = frame_rng(i) if frame_rng is not None else self.rng
                lane.hypotheses(j, rng)
                if j == dec_at and lane.have_prev:
                    ready[p - n_lanes] = lane.decoder(copy_outputs) + (lane,)
                lane.enrich(j, rng)
            lane.have_prev, lane.prev_n = True, len(items)
            nxt_items = start(lane)              # segment 0 of the lane's next group first (it does not touch the depth output), then
            if nxt_items:                        # the finished items go to the caller
                groups[p + n_lanes] = nxt_items
                yield from flush()
            else:                                # the lane's last group: the depth pyramids + decoder of it and nothing new -- into the
                yield from flush()               # output buffer, so only after the caller has been handed what is in there
                ready[p] = lane.drain(copy_outputs) + (lane,)
            p += 1
        yield from flush()
        assert not ready and not groups

    def _run_interleaved_two_programs(self, batches, copy_outputs):
        """run_interleaved for `use_gravity=False` (main.py:244-245, 270-271: SurfaceNormalDORN, no warp).  The DORN backbone shares no
        layer shapes with the depth network's pyramids, so there is no joint 4-group program; the two networks of consecutive frames
        overlap on two HIP streams instead: stream A runs the normals network + plane block of frame t, stream B the enrichment and
        the depth network of frame t-1.  Host order per visit: launch normals(t) [no random numbers], wait for frame t-1's candidate
        counts, draw enrichment(t-1), launch depth(t-1), draw hypotheses(t), launch planes(t) -- the generator is consumed in the order
        hypotheses(0), enrichment(0), hypotheses(1), ... of back-to-back `_call_cnn` calls, and the programs are `_call_cnn`'s own, so
        every frame's depth map is bit-identical to the sequential path's (tests/test_dorn.py).
        Measured (round 4, tools/dorn_stream_rate.py, after depth(t-1) was made to wait for frame t-1's own work only instead of for all
        of stream A): 243 frames/s against 249 for back-to-back `_call_cnn` -- NO gain.  Two different batch-1 programs side by side
        alternate on the CUs rather than share them (a conv workgroup reserves 48-128 KB of LDS), and what the joint 4-group program of
        the gravity path gains comes from sharing LAUNCHES, which these two networks cannot.  The mode is kept for the API (a stream
        of batches in, depth maps out, draw order preserved); it is not a throughput mode."""
        dev = self.device
        if getattr(self, "_two", None) is None:
            self._two = {"a": torch.cuda.Stream(device=dev), "b": torch.cuda.Stream(device=dev), "planes": [PlaneBlock(), PlaneBlock()]}
        sa, sb, blocks = self._two["a"], self._two["b"], self._two["planes"]
        main = torch.cuda.current_stream(dev)
        prev, t = None, 0

        def finish(st):
            """enrichment + depth network of a frame on stream B; returns its output (the caller's stream waits for it)."""
            sb.wait_event(st["done_a"])                           # normals + plane kernels + copies of THIS frame (not the next frame's
            #                                                       normals network, which stream A has already been given: that one runs beside
            #                                                       this depth network)
            with torch.cuda.stream(sb):
                depth_in = st["ds"]
                if st["di"] is not None:
                    depth_in = st["planes"].enrich(st["ds"], st["di"], st["nnz"], self.args.enriched_samples, rng=self.rng, info_host=st["info_host"])
                out = self.cnn.enqueue(st["rgb"], st["normals"], depth_in, 0)
                out = out.clone() if copy_outputs else out
            main.wait_stream(sb)
            if copy_outputs and out.is_cuda:
                out.record_stream(main)          # allocated on stream B, consumed on the caller's (see _run_grouped's flush)
            return out

        for batch in batches:
            sa.wait_stream(main)                                  # the batch comes from the caller's stream
            sa.wait_stream(sb)                                    # ... and depth(t-2) has consumed what this visit overwrites
            with torch.cuda.stream(sa):
                ds = self._stage("sparse_depth", batch["sparse_depth"], dev)
                rgb = self._stage("image", batch["image"], dev)
                ds, rgb = ds.clone(), rgb.clone()                 # live until frame t's depth network has run, one visit later
                mh = self._masks_begin(rgb) if self.args.enriched_samples != 0 else None
                normals = self.surface_normal_cnn(rgb)            # (a fresh tensor per call: surface_normal_dorn.forward clones)
            out = finish(prev) if prev is not None else None
            st = {"ds": ds, "rgb": rgb, "normals": normals, "di": None, "nnz": None, "info_host": None, "planes": blocks[t % 2]}
            if self.args.enriched_samples != 0:
                with torch.cuda.stream(sa):
                    homo = self._stage("homogeneous_coordinates", batch["homogeneous_coordinates"], dev)
                    masks = self._masks_end(mh, batch["image"], ds.shape[-2], ds.shape[-1])
                    st["di"], st["nnz"] = st["planes"].plane_depth(normals, masks, ds, homo, rng=self.rng)
                    st["info_host"] = st["planes"].read_info_async(st["nnz"])
            st["done_a"] = torch.cuda.Event()
            st["done_a"].record(sa)
            prev, t = st, t + 1
            if out is not None:
                yield out
        if prev is not None:
            yield finish(prev)

    @torch.no_grad()
    def run_stream(self, batches, in_flight=2, frame_rng=None):
        """Throughput mode: yields the depth map of every batch, in order, with up to `in_flight` frames executing
        concurrently on separate HIP streams (frame i's depth network overlaps frame i+1's surface-normal network --
        at batch 1 the 134 small layer-3 convolutions per network cannot fill 256 CUs on their own).

        Same kernels and arithmetic as `_call_cnn`.  The RANSAC / enrichment draws of different frames interleave
        differently on the shared `self.rng` than in back-to-back `_call_cnn` calls (frame i+1's hypotheses are drawn
        before frame i's enrichment indices); pass `frame_rng(i) -> np.random.RandomState` to give every frame its own
        generator, in which case the outputs are identical to sequential calls using the same per-frame generators."""
        import collections
        if not hasattr(self, "_slots") or len(self._slots) != in_flight:
            self._slots = [{"stream": torch.cuda.Stream(device=self.device), "planes": PlaneBlock(), "state": None, "out": None,
                            "done": torch.cuda.Event()} for _ in range(in_flight)]
        main = torch.cuda.current_stream(self.device)
        order = collections.deque()

        def finish_stage2(sl, k):
            with torch.cuda.stream(sl["stream"]):
                sl["out"] = self._stage2(sl["state"], slot=k, planes=sl["planes"]).clone()
                sl["done"].record(sl["stream"])
            sl["state"] = None

        for i, batch in enumerate(batches):
            k = i % in_flight
            sl = self._slots[k]
            if sl["state"] is not None:
                finish_stage2(sl, k)             # host waits here for frame i-in_flight's counts; the other slots keep the GPU busy
            if sl["out"] is not None and len(order) >= in_flight:
                j = order.popleft()
                sj = self._slots[j]
                sj["done"].synchronize()
                out, sj["out"] = sj["out"], None
                yield out
            sl["stream"].wait_stream(main)
            with torch.cuda.stream(sl["stream"]):
                sl["state"] = self._stage1(batch, slot=k, planes=sl["planes"], rng=frame_rng(i) if frame_rng else None)
            order.append(k)
        # drain: RNG order must stay frame order, so finish the pending second stages oldest first
        pending = list(order)
        for j in pending:
            if self._slots[j]["state"] is not None:
                finish_stage2(self._slots[j], j)
        while order:
            j = order.popleft()
            sj = self._slots[j]
            sj["done"].synchronize()
            out, sj["out"] = sj["out"], None
            main.wait_stream(sj["stream"])
            yield out


_LANE_STREAMS = {}


def _lane_stream(device, index):
    """The HIP stream of lane `index` on `device`, ONE per process: every pipeline object's lane k runs on the same stream.  The runtime
    multiplexes streams onto a few hardware queues (4 by default); a second pipeline that created three fresh streams after the first one's
    (bench.py's mixed leg after its fp32 leg) got two of its lanes onto one queue, where their kernels serialise: 588 frames/s against 734
    for the same leg in a process of its own.  Streams created once, early, keep the mapping they were measured with."""
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), index)
    st = _LANE_STREAMS.get(key)
    if st is None:
        # (HIP stream priorities per lane -- a favoured first lane to shorten the fill of a short stream -- measured: the 20-step rate is
        #  unchanged and the steady-state rate drops 658 -> 594-629 frames/s, profiles/r5_lane_priorities.txt; all lanes at the default priority)
        st = _LANE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def reserve_lane_streams(device, lanes=3):
    """Creates the process-wide lane streams of `device` and runs one launch on each, so that they take their hardware queues NOW.  The
    runtime binds a stream to the least-used of its 4 hardware queues at the stream's first launch; streams that came first keep a queue
    of their own.  Call this before anything else makes streams on the device -- in particular before `init_process_group("nccl")` / the
    first RCCL collective: with RCCL's streams in place first, the third lane shares the caller's queue and the stream mode measured
    324 instead of 363 frames/s (fp32) and 577 instead of 733 (mixed) on one MI355X (profiles/r4_lane_streams_hw_queues.txt)."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("reserve_lane_streams: a GPU device is required (no CPU fallback)")
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    fresh = [i for i in range(int(lanes)) if (idx, i) not in _LANE_STREAMS]
    if not fresh:
        return
    probe = torch.zeros(64, device=dev)
    torch.cuda.current_stream(dev).synchronize()
    for i in fresh:
        with torch.cuda.stream(_lane_stream(dev, i)):
            probe.add_(1.0)
    torch.cuda.synchronize(dev)


class _GroupLane:
    """One lane of `DepthCompletionPipeline._run_grouped`: a frame program recorded for batch F x B whose batch slots hold F consecutive
    items of the stream, a plane block per slot (an item's plane buffers live from its hypothesis draws to its enrichment, which for the
    last slot of a group is after the next lane's visit has begun), its own HIP stream, and the group whose depth network is pending."""

    def __init__(self, pipe, index, F):
        self.pipe, self.index, self.F = pipe, index, F
        ent = pipe.__dict__.setdefault("_group_lane_cache", {}).setdefault((index, F), {})
        if "stream" not in ent:
            ent["stream"] = _lane_stream(pipe.device, index)
            ent["planes"] = [PlaneBlock() for _ in range(F)]
            ent["stagers"] = [_Stager() for _ in range(F)]
        self.cache, self.stream, self.planes, self.stagers = ent, ent["stream"], ent["planes"], ent["stagers"]
        self.prog, self.shape0, self.have_prev, self.prev_n, self.consumed = None, None, False, 0, None
        self.cur = [None] * F                     # per slot: the item whose plane block / enrichment is still to come
        self.pending = [None] * F

    # ---- program ---------------------------------------------------------------------------------------------------------------
    def _prepare(self, rgb):
        import os
        pipe = self.pipe
        B, _, H, W = rgb.shape
        if self.prog is not None:
            if (B, H, W) != self.shape0:
                raise ValueError("run_interleaved: all batches of a stream must have the same shape (got %s after %s); "
                                 "start a new stream for the remainder" % ((B, H, W), self.shape0))
            return
        self.shape0 = (B, H, W)
        FB = self.F * B
        key = (FB, H, W, pipe.surface_normal_cnn._version, pipe.cnn._version, pipe.surface_normal_cnn.warp_2dof_alignment.align_corners)
        if self.cache.get("prog_key") != key:
            # (lane 0 runs the pipeline's own cached program of that batch: `frame_program(F * B, H, W)` is what tools and bench.py inspect)
            self.cache["prog"] = pipe.frame_program(FB, H, W) if self.index == 0 else build_frame_program(pipe.surface_normal_cnn, pipe.cnn, FB, H, W, pipe.device)
            self.cache["prog_key"] = key
            self.cache["ds"] = torch.zeros((FB, 1, H, W), dtype=torch.float32, device=pipe.device)
            self.cache["homo"] = [None] * self.F
        prog = self.prog = self.cache["prog"]
        pipe.surface_normal_cnn._check(rgb)
        pipe.cnn._check(rgb)
        if not prog.captured and os.environ.get("VIDC_EXEC", "graph") == "graph":
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                prog.run()            # warm-up outside capture (sets kernel attributes)
                prog.capture_segments()
                for vname in ("head", "tail"):
                    if prog.has_variant(vname):
                        prog.capture_variant(vname)
                # every captured graph once, on whatever the buffers hold: the first launch of an executable graph uploads it to the device
                # (hundreds of microseconds for a 100-node graph), and a short stream would otherwise pay that inside its first full tick
                for k in (0, 1):
                    prog.launch_segment(k)
                for vname in ("head", "tail"):
                    if prog.has_variant(vname):
                        prog.launch_variant(vname)
            torch.cuda.current_stream().wait_stream(side)
        self.sn_image, self.dc_image = prog.tensor(prog.inputs["sn_image"]), prog.tensor(prog.inputs["dc_image"])
        self.dc_depth = prog.tensor(prog.inputs["dc_depth"])
        self.normals = prog.tensor(prog.outputs["normals"])
        self.grav = prog.storage[prog.inputs["gravity"].buf][: FB * 3]
        self.algn = prog.storage[prog.inputs["aligned"].buf][: FB * 3]
        self.ds_own = self.cache["ds"]

    def _homo(self, j, batch):
        """The item's homogeneous coordinates on the device, safe to read after the caller got control back: a host tensor goes through
        the slot's staging ring; a device tensor of the caller is copied into the slot's own buffer on EVERY put (0.9 MB device to
        device on the lane's stream, overlapped).  Round 4 cached the copy keyed on (data_ptr, _version): writes through raw pointers
        -- this package's own kernels, dlpack / numpy views -- do not bump `_version` and a refilled grid was served stale, `_version`
        raises on inference-mode tensors, and the cache pinned the caller's tensor.  `pipe.static_grid = True` opts back into copying
        only when the tensor OBJECT changes (a camera whose grid is never rewritten in place)."""
        t = batch["homogeneous_coordinates"]
        if not t.is_cuda:
            return self.stagers[j]("homogeneous_coordinates", t, self.pipe.device)
        ent = self.cache["homo"][j]
        if ent is None or ent[1].shape != t.shape or ent[1].dtype != t.dtype:
            ent = [None, torch.empty_like(t)]
            self.cache["homo"][j] = ent
        if not (getattr(self.pipe, "static_grid", False) and ent[0] is not None and ent[0]() is t):
            ent[1].copy_(t, non_blocking=True)
            ent[0] = weakref.ref(t)
        return ent[1]

    # ---- launch sequence -------------------------------------------------------------------------------------------------------------
    def put(self, j, batch):
        """One item into batch slot j of the lane's program (everything the lane reads later is copied or staged here: the caller may
        refill its tensors as soon as this returns)."""
        pipe, dev = self.pipe, self.pipe.device
        self.stream.wait_stream(torch.cuda.current_stream())          # the item comes from the caller's stream
        with torch.cuda.stream(self.stream):
            rgb = self.stagers[j]("image", batch["image"], dev)
            if j == 0:
                self._prepare(rgb)
                if self.have_prev:                # the previous group's images are still in the surface-normal input; its normals and its
                    self.dc_image.copy_(self.sn_image, non_blocking=True)     # enriched depths were written in place
            B = self.shape0[0]
            if tuple(rgb.shape) != (B, 3) + self.shape0[1:]:
                raise ValueError("run_interleaved: all batches of a stream must have the same shape (got %s after %s); "
                                 "start a new stream for the remainder" % ((rgb.shape[0],) + tuple(rgb.shape[2:]), self.shape0))
            sl = slice(j * B, (j + 1) * B)
            mh = pipe._masks_begin(rgb, key=(self.index, j)) if pipe.args.enriched_samples != 0 else None
            self.sn_image[sl].copy_(rgb, non_blocking=True)
            self.grav[3 * j * B: 3 * (j + 1) * B].copy_(self.stagers[j]("gravity", batch["gravity"], dev).reshape(-1), non_blocking=True)
            self.algn[3 * j * B: 3 * (j + 1) * B].copy_(self.stagers[j]("aligned_direction", batch["aligned_direction"], dev).reshape(-1), non_blocking=True)
            ds = self.ds_own[sl]                  # own copy: the enrichment reads it after the caller may have refilled its buffer
            ds.copy_(self.stagers[j]("sparse_depth", batch["sparse_depth"], dev), non_blocking=True)
            homo = self._homo(j, batch) if pipe.args.enriched_samples != 0 else None
            # (host-resident images stay referenced for a `run_on_tensor` extractor, which is called with the item's own image later)
            self.cur[j] = (batch["image"], ds, homo, mh)
            taken = torch.cuda.Event()
            taken.record()
        torch.cuda.current_stream().wait_event(taken)     # the caller's later writes to its tensors are ordered behind the lane's reads

    def begin(self, n):
        """Segment 0 for the n items just put: all four pyramids + the surface-normal decoder (the surface-normal side alone when the
        lane has no previous group).  Draws nothing, waits for nothing."""
        for j in range(n, self.F):
            self.cur[j] = None
        with torch.cuda.stream(self.stream):
            prog = self.prog
            if not self.have_prev and prog.has_variant("head"):
                prog.launch_variant("head") if prog.captured else prog.run_variant("head")
            else:
                prog.launch_segment(0) if prog.captured else prog.run_segment(0)

    def decoder(self, copy_outputs):
        """Segment 1: the depth decoder of the lane's previous group (its pyramids ran in the segment 0 just before).  Returns
        (outputs of the F slots, event, number of items in that group)."""
        with torch.cuda.stream(self.stream):
            prog = self.prog
            if self.consumed is not None:         # the caller has read the previous output out of the program's own buffer
                self.stream.wait_event(self.consumed)
                self.consumed = None
            prog.launch_segment(1) if prog.captured else prog.run_segment(1)
            out = prog.tensor(prog.outputs["depth"])
            out = out.clone() if copy_outputs else out
            ev = torch.cuda.Event()
            ev.record()
        return out, ev, self.prev_n

    def drain(self, copy_outputs):
        """After the lane's last group: its three depth pyramids (the "tail" variant of segment 0) and its decoder."""
        with torch.cuda.stream(self.stream):
            prog = self.prog
            self.dc_image.copy_(self.sn_image, non_blocking=True)
            if prog.has_variant("tail"):
                prog.launch_variant("tail") if prog.captured else prog.run_variant("tail")
            else:
                prog.launch_segment(0) if prog.captured else prog.run_segment(0)
        out = self.decoder(copy_outputs)
        self.have_prev = False
        return out

    # ---- draw sequence ---------------------------------------------------------------------------------------------------------------
    def hypotheses(self, j, rng):
        """main.py:272-283 for slot j: hypothesis draws, the plane kernels, the asynchronous read of the candidate counts."""
        pipe = self.pipe
        image, ds, homo, mh = self.cur[j]
        B, H, W = self.shape0
        sl = slice(j * B, (j + 1) * B)
        with torch.cuda.stream(self.stream):
            if pipe.args.enriched_samples != 0:
                masks = pipe._masks_end(mh, image, H, W)
                di, info = self.planes[j].plane_depth(self.normals[sl], masks, ds, homo, rng=rng)
                self.pending[j] = (ds, di, info, self.planes[j].read_info_async(info))
            else:
                self.dc_depth[sl].copy_(ds, non_blocking=True)
                self.pending[j] = None
        self.cur[j] = None

    def enrich(self, j, rng):
        """main.py:285-294 for slot j: waits for the counts, draws the samples, writes the enriched depth where the lane's next
        segment 0 reads it."""
        if self.pending[j] is None:
            return
        ds, di, info, info_host = self.pending[j]
        self.pending[j] = None
        B = self.shape0[0]
        with torch.cuda.stream(self.stream):
            self.planes[j].enrich(ds, di, info, self.pipe.args.enriched_samples, rng=rng, info_host=info_host, out=self.dc_depth[j * B:(j + 1) * B])
