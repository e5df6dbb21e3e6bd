"""Training step of the depth-completion network on the GPU (BASELINE configs[4]).

`DepthCompletionTrainer(cnn, learning_rate).step(image, normal, depth_in, depth_gt)` is the body of the reference's
`ImageNetworkRunInterface._run_training_iteration` (network_run.py:231-254) for `self.cnn = ModifiedFPN`:

    cnn.train()                       BatchNorm on batch statistics, running statistics updated (momentum 0.1)
    outputs = cnn(image, normals, depth_enriched)          (the three inputs are what `_call_cnn`, main.py:261-298, feeds the network;
                                                            they do not depend on the trained parameters)
    loss = L1Loss(sum)(pred[gt > 0], gt[gt > 0]) / (H*W)   network_run.py:163-173
    loss.backward(); Adam(cnn.parameters(), lr).step()      network_run.py:228-229, 249-250

Everything numeric is a libvidc.so kernel: forward convs = the inference MFMA kernel (csrc/conv_mfma.hip) with the raw bias instead of a
folded BatchNorm, data gradients = the same kernel on flipped/transposed weights, weight gradients / BatchNorm / pooling / upsampling /
head / loss / Adam = csrc/train.hip.  PyTorch provides device memory, the parameter containers (the module keeps its reference
state_dict layout; parameters and their .grad become views of two flat buffers) and, across GPUs, `torch.distributed.all_reduce` of
the flat gradient buffer (RCCL): frames shard over ranks, every rank normalises BatchNorm over ITS frames (what the reference's
DataParallel replicas do as well) and gradients are SUMMED, which equals the reference's single loss over the whole batch.
There is no CPU path.  Arithmetic: exact fp32 MFMA products in all three conv passes (the reference's arithmetic: gradients sit at
the reference's own fp32 noise floor).  VIDC_TRAIN_PRECISION=bf16x3 runs the three conv passes in the split-bf16 3-pass mode of
the inference path (~2^-16 per product; 1.5x the step rate) and VIDC_TRAIN_PRECISION=bf16 in plain bf16 with fp32 accumulation -- the
arithmetic BASELINE configs[4] names (2x; fp32 master weights, BatchNorm, loss and Adam in all modes).  Both keep the loss to 4-6
digits; individual gradient tensors move by percents of their scale (flipped ReLU gates propagate through the train-mode BatchNorms):
throughput modes, not the parity mode (tests/test_training.py states their bars)."""
import ctypes as C
import json
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import engine
from .networks.fpn_decoder import _BRANCH_PLAN

BN_EPS, BN_MOMENTUM = 1e-5, 0.1
PYRAMIDS = ("resnet_rgb", "resnet_normal", "resnet_depth")      # the three ResNet-101 pyramids of ModifiedFPN (depth_completion.py:75-77)


def _keys(key):
    """A parameter-name prefix, or one per group of a grouped launch (the same layer of the three pyramids)."""
    return (key,) if isinstance(key, str) else tuple(key)


def _cat(p, suffix):
    return tuple(q + suffix for q in _keys(p)) if not isinstance(p, str) else p + suffix


class Act:
    """An activation: NHWC rows `t` (B,H,W,C view, possibly a channel slice of a wider buffer) + its gradient (same geometry)."""
    __slots__ = ("t", "grad", "bf", "grad_bf", "grad_t", "conv_out", "no_f32_grad", "stats")      # bf / grad_bf: bf16 operand copies (dense rows) written by the producing kernel
    # grad_t: (tensor, Mp) -- the gradient transposed as bf16 rows [C][Mp], written by the BatchNorm backward behind a conv (conv_out): the
    # left operand of that conv's weight-gradient GEMM

    def __init__(self, t):
        self.t, self.grad, self.bf, self.grad_bf, self.grad_t, self.conv_out = t, None, None, None, None, False
        self.stats = None             # conv output in the plain-bf16 mode: per-32-row-block channel sums written by the conv's epilogue (VIDC_STATS_OUT)
        self.no_f32_grad = False      # conv output whose conv reads dY only through grad_bf / grad_t: the BatchNorm backward skips the fp32 form

    @property
    def ld(self):
        return self.t.stride(2)

    @property
    def rows(self):
        return self.t.shape[0] * self.t.shape[1] * self.t.shape[2]


_TRAIN_TUNING = None


def training_table():
    """Per-shape [tile_fp32, splitk_fp32, tile_bf16x3, splitk_bf16x3] of the training step's conv launches (forward, dgrad and the
    wgrad GEMMs), measured on MI355X by tools/autotune_train.py.  VIDC_TRAIN_TUNING=0 ignores it (cost-model plan)."""
    global _TRAIN_TUNING
    if _TRAIN_TUNING is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "train_tuning.json")
        use = os.environ.get("VIDC_TRAIN_TUNING", "1") != "0" and os.path.exists(path)
        _TRAIN_TUNING = json.load(open(path)) if use else {}
    return _TRAIN_TUNING


def _ld(t):
    return t.stride(2)


class GradientBuckets:
    """The flat gradient buffer cut into buckets for the cross-rank SUM (one collective per bucket, ~64 MB each: large enough that a
    ring all-reduce over xGMI runs at link speed, small enough to start before the whole backward has finished if overlapped).

    compress="bf16" (DepthCompletionTrainer: VIDC_TRAIN_GRAD_BF16=1): every bucket travels rounded to bf16 -- 0.62 GB instead of
    1.24 GB per step for ModifiedFPN, i.e. ~7 instead of ~14 ms on a ring at 153 GB/s per xGMI link -- and is widened back into the
    fp32 buffer before Adam reads it (vidc_grad_narrow_bf16 / vidc_grad_widen_bf16; the SUM itself is RCCL's, in bf16).  An opt-in:
    the reference's DataParallel sums its replicas' gradients in fp32 (network_run.py:97-99)."""

    def __init__(self, n_elements, bucket_elements=16 * 1024 * 1024, compress=None):
        self.ranges = [(a, min(n_elements, a + bucket_elements)) for a in range(0, n_elements, bucket_elements)]
        self.compress = compress
        self._narrow = None            # persistent bf16 image of the flat buffer (allocated on first use)

    def all_reduce(self, flat):
        for w in self.all_reduce_async(flat):
            w()
        return flat

    def all_reduce_async(self, flat, lo=0, hi=None):
        """Starts the SUM over ranks of flat[lo:hi], bucket by bucket, and returns the completion callbacks (call every one before the
        data is used).  RCCL: the collectives run on the process group's own stream, after the work already enqueued on the current
        stream, and overlap whatever is enqueued next -- the decoder's gradients travel while the pyramids' backward runs
        (DepthCompletionTrainer.step).  gloo (trying the N > 1 path on a box without a second GPU): through the host, synchronous."""
        import torch.distributed as dist
        from .sharding import collectives_active
        if not collectives_active():
            return []
        hi = flat.numel() if hi is None else hi
        stage = dist.get_backend() == "gloo" and flat.is_cuda
        bf16 = self.compress == "bf16" and flat.is_cuda
        if bf16 and (self._narrow is None or self._narrow.numel() != flat.numel() or self._narrow.device != flat.device):
            self._narrow = torch.empty(flat.numel(), dtype=torch.bfloat16, device=flat.device)
        waits = []
        for a, b in self.ranges:
            a, b = max(a, lo), min(b, hi)
            if a >= b:
                continue
            if bf16 and a % 8 and b - a > 8:      # a range that starts off the kernels' 16-byte grid (the decoder's first parameter): its first
                a_up = a + 8 - a % 8              # few elements travel as they are, in fp32
                head = flat[a:a_up]
                if stage:
                    h = head.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    head.copy_(h)
                else:
                    waits.append(dist.all_reduce(head, op=dist.ReduceOp.SUM, async_op=True).wait)
                a = a_up
            if bf16 and a % 8 == 0:
                src, nb = flat[a:b], self._narrow[a:b]
                L.check(L.lib().vidc_grad_narrow_bf16(L.ptr(src), L.ptr(nb), b - a, L.current_stream()), "grad narrow")
                if stage:
                    h = nb.cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    nb.copy_(h)
                    L.check(L.lib().vidc_grad_widen_bf16(L.ptr(nb), L.ptr(src), b - a, L.current_stream()), "grad widen")
                else:
                    work = dist.all_reduce(nb, op=dist.ReduceOp.SUM, async_op=True)

                    def done(work=work, nb=nb, src=src, n=b - a):
                        work.wait()               # (the current stream waits for the collective; the widening pass follows on it)
                        L.check(L.lib().vidc_grad_widen_bf16(L.ptr(nb), L.ptr(src), n, L.current_stream()), "grad widen")
                    waits.append(done)
            elif stage:
                h = flat[a:b].cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                flat[a:b].copy_(h)
            else:
                waits.append(dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, async_op=True).wait)
        return waits


class DepthCompletionTrainer:
    """Build the trainer AFTER `torch.distributed.init_process_group` (and after the network is on its GPU): the flat layout of parameters, gradients, Adam
    moments and BatchNorm running statistics is chosen here (`self.layout`: "grouped" = the three pyramids interleaved per parameter name, the multi-rank
    default; "per_pyramid" otherwise) and every launch of the step addresses tensors as base + offset into those buffers.  A later `cnn.to()` / `.float()` /
    `.cuda()` that rebinds `p.data` is caught at the next `step()` (`_check_bindings`), not silently read through stale pointers."""

    def __init__(self, cnn, learning_rate=1e-4, betas=(0.9, 0.999), eps=1e-8):
        if not torch.cuda.is_available():
            raise RuntimeError("DepthCompletionTrainer needs a GPU: the HIP path has no CPU fallback")
        self.cnn = cnn
        self.lr, self.betas, self.eps = float(learning_rate), betas, float(eps)
        self.named = [(k, p) for k, p in cnn.named_parameters()]
        # Grouped pyramids (round 5, VIDC_TRAIN_GROUPED=0 switches back to one launch chain per pyramid): the same layer of the three
        # pyramids runs as ONE launch with three groups -- forward conv, data gradient, weight-gradient GEMM -- and ONE BatchNorm launch over
        # the 3 x C channels of the grouped tensor, as the inference frame program has always done.  A group selects base pointers only, so
        # the three layers' parameters (and gradients, Adam moments, BatchNorm running statistics) are laid out next to each other in the
        # flat buffers: [rgb | normal | depth] per parameter name.  Values, gradients and the all-reduced sums are what they were.
        # (the grouped weight-gradient GEMM writes the three .grad tensors in place; the A/B switch VIDC_TRAIN_WGRAD_INPLACE=0 -- staging
        #  buffer + permute, tap-major operand rows -- therefore runs the per-pyramid chains)
        # Default ("auto"): grouped + ONE stream when the step runs across ranks, three per-pyramid stream lanes on a single rank.  Measured
        # (MI355X, batch 8, bf16, profiles/r5_train_grouped_ab.txt): grouped = 1252 launches / 27.8 ms of kernel time per step instead of
        # 3258 / 42.0 ms, but as ONE dependency chain (27.3-27.5 ms per step) against 25.9 ms for the three lanes, whose launch latencies
        # hide under each other -- on this runtime kernels of different streams fill each other's gaps, they do not add throughput.
        # Across ranks the step is cut into two graphs around the decoder's all-reduce, and there the three-lane form falls into the
        # runtime's slow regime above 4 hardware queues (50.4 against 27.4 ms at GPU_MAX_HW_QUEUES=8) while the single-stream grouped chain
        # does not care (27.8 / 28.4 ms at 4 / 8 queues): the multi-rank default must not depend on a queue count nobody controls.
        mode = os.environ.get("VIDC_TRAIN_GROUPED", "auto")
        want = self._distributed() if mode == "auto" else mode == "1"
        self.grouped = (want and os.environ.get("VIDC_TRAIN_WGRAD_INPLACE", "1") == "1" and all(hasattr(cnn, pn) for pn in PYRAMIDS))
        if self.grouped:
            by_name = dict(self.named)
            first = PYRAMIDS[0] + "."
            inter = []
            for k, _p in self.named:
                if k.startswith(first):
                    for pn in PYRAMIDS:
                        kk = pn + "." + k[len(first):]
                        inter.append((kk, by_name[kk]))
            seen = {k for k, _ in inter}
            self.named = inter + [(k, q) for k, q in self.named if k not in seen]
        dev = self.named[0][1].device
        if dev.type != "cuda":
            raise RuntimeError("move the network to the GPU before building the trainer")
        self.device = dev
        n = sum(p.numel() for _, p in self.named)
        self.flat_p = torch.empty(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        self.param, self.grad = {}, {}
        o = 0
        for k, p in self.named:            # parameters and gradients become views of the flat buffers (one Adam launch, bucketed all-reduce)
            k_n = p.numel()
            self.flat_p[o:o + k_n].copy_(p.detach().reshape(-1))
            p.data = self.flat_p[o:o + k_n].view(p.shape)
            p.grad = self.flat_g[o:o + k_n].view(p.shape)
            self.param[k], self.grad[k] = p.data, p.grad
            o += k_n
        self.buf = {k: b for k, b in cnn.named_buffers()}
        if self.grouped:      # BatchNorm running statistics of the three pyramids: the same interleaved order, in a flat buffer of their own
            first = PYRAMIDS[0] + "."
            fl = [k for k in self.buf if k.startswith(first) and self.buf[k].dtype == torch.float32]
            tot = sum(self.buf[k].numel() for k in fl) * len(PYRAMIDS)
            self.flat_b = torch.empty(tot, dtype=torch.float32, device=dev)
            o = 0
            for k in fl:
                for pn in PYRAMIDS:
                    b = self.buf[pn + "." + k[len(first):]]
                    self.flat_b[o:o + b.numel()].copy_(b.detach().reshape(-1))
                    b.data = self.flat_b[o:o + b.numel()].view(b.shape)
                    o += b.numel()
        self._adjacent = {}
        self.layout = "grouped" if self.grouped else "per_pyramid"      # (exported with `flat_state()`: moments of one layout are not another's)
        self._bind_probe = [(self.named[i][1], self.param[self.named[i][0]].data_ptr(), self.grad[self.named[i][0]].data_ptr()) for i in (0, len(self.named) - 1)]
        if self.grouped:
            fb = [k for k in self.buf if self.buf[k].dtype == torch.float32 and any(k.startswith(pn + ".") for pn in PYRAMIDS)]
            self._bind_probe += [(self.buf[k], self.buf[k].data_ptr(), None) for k in (fb[0], fb[-1])] if fb else []
        self.buckets = GradientBuckets(n, compress=("bf16" if os.environ.get("VIDC_TRAIN_GRAD_BF16", "0") == "1" else None))
        # the decoder's parameters (feature*_upsamping, feature_concat) form the tail of the flat buffers (named_parameters order): their
        # gradients are complete when the decoder's backward is, long before the pyramids' -- they are all-reduced while those still run
        offs = {k: int((self.grad[k].data_ptr() - self.flat_g.data_ptr()) // 4) for k, _ in self.named}
        dec = [offs[k] for k, _ in self.named if k.startswith("feature")]
        self._dec_off = min(dec) if dec and all(k.startswith("feature") for k, _ in self.named if offs[k] >= min(dec)) else n
        self.step_count = 0
        self._ones, self._zeros, self._packed, self._scratch, self._nbt = {}, {}, {}, {}, []
        self._gemm_ws = {}          # split-K workspace per stream lane (lane 0 = the caller's stream)
        self._cur = 0               # stream lane the ops being recorded / replayed run on (0 = main, 1..3 = the three pyramids, 1..4 = the decoder branches)
        self._lanes = None
        self.n_lanes = int(os.environ.get("VIDC_TRAIN_STREAMS", "1" if self.grouped else "3"))      # (grouped: the decoder's four branches stay on the caller's stream too)
        self._pack_items, self._pack_table, self._packed_fresh = [], None, False
        self.use_graph = os.environ.get("VIDC_TRAIN_GRAPH", "1") != "0"
        self._graphs, self._graph_seen = {}, {}
        # Opt-in experiment (VIDC_TRAIN_WGRAD_STREAM=1), measured SLOWER and therefore off: dW of a conv (two operand transposes, a GEMM, a
        # permute, the bias column sum -- 4-5 launches) depends on dY only and nothing depends on it until Adam, so it can run on a side
        # stream of its lane while the lane goes on with the data gradient (in the captured graph: a parallel branch per conv), taking a
        # third of the launches out of a pyramid's dependency chain.  Round 3, batch 8, same box, identical losses: 41.6 ms per step
        # against 32.6 in bf16, 80.4 against 71.6 in fp32 -- ~340 extra fork / join edges per step cost more than the shorter chain
        # saves (the same finding as for side streams inside the inference graphs).
        self.wgrad_side = os.environ.get("VIDC_TRAIN_WGRAD_STREAM", "0") == "1"
        # Round-3 launch reductions, each bit-identical to the form it replaces; the switches exist for the A/B runs and the tests:
        # dY^T of a conv written by the BatchNorm backward that produces dY; the weight-gradient GEMM writing .grad in place (channel-major
        # operand rows); the residual add writing the block output's bf16 operand copy.
        self.dyt_fused = os.environ.get("VIDC_TRAIN_DYT_FUSED", "1") == "1"
        self.wgrad_inplace = os.environ.get("VIDC_TRAIN_WGRAD_INPLACE", "1") == "1"
        self.add_bf16 = os.environ.get("VIDC_TRAIN_ADD_BF16", "1") == "1"
        self.bn_add_fused = os.environ.get("VIDC_TRAIN_BN_ADD_FUSED", "1") == "1"   # Bottleneck tail relu(bn3(.) + identity) inside bn3's apply pass
        self.skip_f32_dy = os.environ.get("VIDC_TRAIN_SKIP_F32_DY", "1") == "1"    # the BatchNorm backward writes no fp32 dY where only the bf16 forms are read
        self.xt_from_bf16 = os.environ.get("VIDC_TRAIN_XT_BF16", "1") == "1"      # 1x1 convs: the wgrad GEMM's right operand transposed from the bf16 copy
        # Round 4: the train-mode BatchNorm behind a conv takes its per-channel sums from the conv's epilogue (VIDC_STATS_OUT, plain-bf16 mode)
        # instead of a partial-sum pass of its own over the conv output: one launch and one read of the tensor less per conv + BatchNorm.
        self.conv_stats = os.environ.get("VIDC_TRAIN_CONV_STATS", "1") == "1"
        self.conv_stats_max_m = int(os.environ.get("VIDC_TRAIN_CONV_STATS_MAX_M", str(1 << 30)))
        if "VIDC_TRAIN_BN_FOLD" in os.environ:   # (A/B runs) the BatchNorm chunk sums reduced in the consumer's prologue (default) or by a launch of their own
            L.lib().vidc_train_bn_fold(int(os.environ["VIDC_TRAIN_BN_FOLD"] != "0"))
        self._wgrad_streams, self._wgrad_used = {}, []
        self._retired = []          # outgrown scratch / workspace buffers that captured graphs still address (see _retire)
        self._keepalive = []        # backward closures already run in the current _run_tape, kept until the stream lanes have joined
        self.tune_hook = None      # tools/autotune_train.py: called with every conv descriptor before it is planned
        self.precision = {"fp32": L.PREC_FP32, "bf16x3": L.PREC_BF16X3, "bf16": L.PREC_BF16}[os.environ.get("VIDC_TRAIN_PRECISION", "fp32")]
        self.last_loss = None

    # ---- small helpers ------------------------------------------------------------------------------------------------------
    def _const(self, store, n, value):
        if n not in store:
            store[n] = torch.full((n,), value, dtype=torch.float32, device=self.device)
            torch.cuda.current_stream().synchronize()     # made on one stream lane, read by all of them from now on
        return store[n]

    def _scratch_bytes(self, nbytes):
        sc = self._scratch.get(self._cur)                 # one scratch per stream lane: the lanes run concurrently
        if sc is None or sc.numel() < nbytes:
            self._retire(sc)
            sc = self._scratch[self._cur] = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=self.device)
        return sc

    def _retire(self, t):
        """A scratch / split-K workspace that is being replaced by a larger one.  Captured step graphs of OTHER input shapes hold its raw
        address (BN / column-sum partials, split-K partial tiles and their ticket counters), so it must never go back to the allocator
        while such a graph can still be replayed: it is parked for the lifetime of the trainer (the graphs keep using it; new eager
        steps and new captures use the larger buffer)."""
        if t is not None and self._graphs:
            self._retired.append(t)

    def _record(self, fn):
        fn._lane = self._cur
        self.tape.append(fn)

    def _beside(self, fn):
        """Runs `fn` (launches that nothing on the current lane waits for before the end of the backward) on the lane's side stream, with
        its own scratch / split-K workspace key; `_join_wgrad` makes the main stream wait for all of them."""
        if not self.wgrad_side:
            fn()
            return
        lane = self._cur
        cur = torch.cuda.current_stream()
        side = self._wgrad_streams.get(lane)
        if side is None:
            side = self._wgrad_streams[lane] = torch.cuda.Stream(device=self.device)
        side.wait_stream(cur)                     # dY (and its ReLU mask) are complete on the lane
        self._keepalive.append(fn)                # fn owns tensors of the LANE's allocator pool (the masked dY): alive until the join
        self._cur = lane + 8
        try:
            with torch.cuda.stream(side):
                fn()
        finally:
            self._cur = lane
        if side not in self._wgrad_used:
            self._wgrad_used.append(side)

    def _join_wgrad(self, main):
        for side in self._wgrad_used:
            main.wait_stream(side)
        self._wgrad_used = []

    def _lane_streams(self):
        if self._lanes is None:
            self._lanes = [torch.cuda.Stream(device=self.device) for _ in range(4)]
        return self._lanes

    def _train_scratch(self, M, Cc):
        return self._scratch_bytes(L.lib().vidc_train_scratch_bytes(M, Cc))

    def _empty(self, *shape):
        return torch.empty(shape, dtype=torch.float32, device=self.device)

    # ---- conv: forward, dgrad, wgrad --------------------------------------------------------------------------------------------
    def _conv_call(self, x_t, w_packed, shift, y_t, kh, kw, stride, pad, relu, accumulate, x_bf=None, stats=None, groups=1):
        """One launch of the inference conv kernel.  In bf16x3 mode the activations are split here (one extra pass over x); `w_packed`
        must already be in the matching format (`_pack`).  groups = G > 1: x_t / y_t hold G groups as channel slices ([.., G*cin] /
        [.., G*cout]), w_packed the G packed weights back to back; `shift` ([cout]) is shared by the groups."""
        B, H, W, cin = x_t.shape
        _, Ho, Wo, cout = y_t.shape
        G = groups
        ldx = _ld(x_t)
        if self.precision == L.PREC_BF16X3:
            xs = self._empty(B, H, W, cin)
            L.check(L.lib().vidc_split_bf16x3(L.ptr(x_t), L.ptr(xs), B * H * W, cin, ldx, L.current_stream()), "split")
            x_t, ldx = xs, cin
        elif self.precision == L.PREC_BF16:      # plain bf16 rows; the descriptor counts two channels per element (include/vidc.h)
            if (cin // G) % 64:
                raise RuntimeError("bf16 training needs conv input channels in multiples of 64 (got %d)" % (cin // G))
            xs = x_bf                             # the producer wrote the bf16 copy already (bn / bn backward)
            if xs is not None and tuple(xs.shape) != (B, H, W, cin // 2):
                raise RuntimeError("bf16 operand copy of shape %s for an activation of shape %s" % (tuple(xs.shape), (B, H, W, cin)))
            if xs is None:
                xs = self._empty(B, H, W, cin // 2)
                L.check(L.lib().vidc_cast_bf16(L.ptr(x_t), L.ptr(xs), B * H * W, cin, ldx, L.current_stream()), "cast")
            x_t, cin = xs, cin // 2
            ldx = cin
        cin, cout = cin // G, cout // G           # per group (cin in the descriptor's units: two bf16 channels per element in the bf16 mode)
        d = L.ConvDesc()
        d.x, d.w, d.y = L.ptr(x_t), L.ptr(w_packed), L.ptr(y_t)
        d.scale1, d.shift1 = L.ptr(self._const(self._ones, cout, 1.0)), L.ptr(shift)
        d.B, d.H, d.W, d.Cin, d.ldx = B, H, W, cin, ldx
        d.Ho, d.Wo, d.Cout, d.ldy = Ho, Wo, cout, _ld(y_t)
        d.KH, d.KW, d.stride, d.pad = kh, kw, stride, pad
        d.flags = (L.RELU1 if relu else 0) | (L.ACCUM if accumulate else 0)
        if stats is not None:                 # (the descriptor's y_split field carries the partials buffer: include/vidc.h VIDC_STATS_OUT)
            d.flags |= L.STATS_OUT
            d.y_split = L.ptr(stats)
        d.groups, d.splitk, d.precision, d.tile = G, 1, self.precision, 0
        d.x_gs, d.w_gs, d.y_gs, d.p_gs = cin, cout * kh * kw * cin, cout, (cout if G == 1 else 0)
        self._plan(d, "conv")
        L.check(L.lib().vidc_conv2d_bn_act(C.byref(d), L.current_stream()), "conv")

    def _plan(self, d, role):
        """Tile / split-K of one launch: the table measured on MI355X at the training shapes (train_tuning.json, tools/autotune_train.py:
        [tile, splitk] per arithmetic mode), else the planner's cost model.  Split-K partial sums go through one persistent workspace
        (ticket counters at its head, zeroed once; the last workgroup of a tile resets its ticket)."""
        lib = L.lib()
        fixed = self.tune_hook(d, role) if self.tune_hook is not None else None      # (a hook that returns True has set tile / splitk itself)
        ent = training_table().get(engine.conv_signature(d))
        if fixed:
            pass
        elif ent is not None and len(ent) > 2 * d.precision + 1 and ent[2 * d.precision]:
            d.tile, d.splitk = ent[2 * d.precision], ent[2 * d.precision + 1]
        else:
            L.check(lib.vidc_conv2d_plan(C.byref(d)), "conv plan")
            if role == "conv":
                d.splitk = 1
        need = lib.vidc_conv2d_workspace_bytes(C.byref(d))
        if need:
            ws = self._gemm_ws.get(self._cur)
            if ws is None or ws.numel() * 4 < need:
                self._retire(ws)
                ws = self._gemm_ws[self._cur] = torch.zeros(int(need // 4 * 1.5) + 16, dtype=torch.float32, device=self.device)
            d.workspace = L.ptr(ws)

    def _pack(self, key, kind, w):
        """Packed weights of conv `key` for the forward ('f') or the dgrad ('d': kernel flipped, channels transposed) launch, in the
        trainer's arithmetic mode.  The parameters move every step: `repack()` rebuilds every packed copy with one launch at the start of
        a forward; a conv met for the first time is packed here and joins the table."""
        keys = _keys(key)                      # (a tuple: the G packed copies back to back, what a grouped launch reads as w + g * w_gs)
        ent = self._packed.get((keys, kind))
        if ent is not None and self._packed_fresh:
            return ent[0]
        co, ci, kh, kw = w.shape
        if ent is None:
            n1 = co * ci * kh * kw // (2 if self.precision == L.PREC_BF16 else 1)
            buf = self._empty(n1 * len(keys))
            mine = []
            for g, k in enumerate(keys):
                wk = self.param[k + ".weight"] if len(keys) > 1 else w
                assert tuple(wk.shape) == (co, ci, kh, kw), k
                mine.append((wk, buf[g * n1:(g + 1) * n1], co, ci, kh, kw, (1 if kind == "d" else 0) | {L.PREC_FP32: 0, L.PREC_BF16X3: 2, L.PREC_BF16: 4}[self.precision]))
            self._pack_items += mine
            ent = self._packed[(keys, kind)] = (buf, mine)
            self._pack_table = None
        self._launch_pack(ent[1])
        return ent[0]

    def _launch_pack(self, items):
        table = (L.PackItem * len(items))()
        blocks = 0
        for t, (w, out, co, ci, kh, kw, kind) in zip(table, items):
            t.w, t.packed, t.Cout, t.Cin, t.KH, t.KW, t.kind, t.block_begin = L.ptr(w), L.ptr(out), co, ci, kh, kw, kind, blocks
            nb = L.lib().vidc_pack_item_blocks(co, ci, kh, kw, kind)
            if nb <= 0:
                raise RuntimeError("conv weight %dx%dx%dx%d cannot be packed for kind %d (K-side channels must fill whole 128-byte units)" % (co, ci, kh, kw, kind))
            blocks += nb
        dev = torch.frombuffer(bytearray(bytes(table)), dtype=torch.uint8).to(self.device)
        L.check(L.lib().vidc_pack_conv_weights_batched(L.ptr(dev), len(items), blocks, L.current_stream()), "pack")
        return dev, blocks

    def repack(self):
        """One launch re-packs every conv weight seen so far (forward and dgrad copies) from the current parameters."""
        if self._pack_items:
            if self._pack_table is None:
                dev, blocks = self._launch_pack(self._pack_items)
                self._pack_table = (dev, blocks, len(self._pack_items))
            else:
                dev, blocks, n = self._pack_table
                L.check(L.lib().vidc_pack_conv_weights_batched(L.ptr(dev), n, blocks, L.current_stream()), "pack")
        self._packed_fresh = True

    def _check_bindings(self):
        """O(1) per step: the first and last parameter (and, grouped, running statistic) still live where the flat buffers put them.  `cnn.to()`,
        `.float()`, `.cuda()` or a `load_state_dict(assign=True)` after construction rebind `.data`; the captured graphs and the grouped launches
        (base + g * numel) would then read the old storage."""
        for t, p_ptr, g_ptr in self._bind_probe:
            if t.data_ptr() != p_ptr or (g_ptr is not None and (t.grad is None or t.grad.data_ptr() != g_ptr)):
                raise RuntimeError("DepthCompletionTrainer: a parameter / buffer of the network was rebound after the trainer was built (layout %r); "
                                   "build a new trainer after moving or casting the network" % self.layout)

    def flat_state(self):
        """Optimizer state for a checkpoint: the flat Adam moments with the layout they belong to and the parameter order that defines the offsets."""
        return {"layout": self.layout, "order": [k for k, _ in self.named], "m": self.m, "v": self.v, "step_count": self.step_count}

    def load_flat_state(self, st):
        if st["layout"] != self.layout or list(st["order"]) != [k for k, _ in self.named]:
            raise RuntimeError("optimizer state of layout %r / another parameter order cannot be loaded into a trainer of layout %r (build the trainer under the "
                               "same VIDC_TRAIN_GROUPED setting and process-group state)" % (st["layout"], self.layout))
        self.m.copy_(st["m"]); self.v.copy_(st["v"]); self.step_count = int(st["step_count"])

    def _adjacent_base(self, store, keys, suffix):
        """The G tensors `store[k + suffix]` of a grouped layer as ONE contiguous run (group g at base + g * numel): the first one's
        tensor after a check, once per layer, that the flat layout really put them next to each other."""
        t0 = store[keys[0] + suffix]
        if len(keys) > 1:
            ck = (id(store), keys, suffix)
            if ck not in self._adjacent:
                for g, k in enumerate(keys):
                    t = store[k + suffix]
                    if t.shape != t0.shape or t.data_ptr() != t0.data_ptr() + 4 * g * t0.numel():
                        raise RuntimeError("grouped training: %s%s of the pyramids are not adjacent in the flat buffers" % (k, suffix))
                self._adjacent[ck] = True
        return t0

    def _wgrad_fits(self, geom, G=1):
        """Whether `_wgrad_gemm` takes this shape (else the direct pixel-reduction kernel runs, which reads the fp32 dY)."""
        B, H, W, ci, Ho, Wo, co, kh, kw, stride, pad = geom
        if os.environ.get("VIDC_WGRAD", "gemm") != "gemm":
            return False
        taps, M = kh * kw, B * Ho * Wo
        bf16 = self.precision == L.PREC_BF16
        Mp = (M + 63) // 64 * 64 if bf16 else (M + 31) // 32 * 32
        e = 2 if bf16 else 1
        return not (taps * ci * Mp * 4 // e >= (1 << 31) or G * co * Mp // e >= (1 << 29) or ci % 32 or co % 4)

    def _wgrad_gemm(self, g, x, key, geom, g_t=None):
        """dW through the conv kernel: dW[co][ci][tap] = sum over pixels of dY^T[co][m] * Xt[ci*taps + tap][m] is the 1x1 case of
        vidc_conv2d_bn_act with the rows of dY^T as activations and the rows of Xt (the transposed im2col of x, channel-major) as weights
        -- LDS-tiled, split-K, at several times the rate of the direct pixel-reduction kernel (vidc_conv_wgrad, kept for shapes beyond the
        32-bit limits of the conv kernel) -- and its output is the parameter's .grad in place.  g_t: dY^T if the BatchNorm backward has
        written it already.  Returns False when the shape does not fit."""
        B, H, W, ci, Ho, Wo, co, kh, kw, stride, pad = geom          # ci, co: per group
        keys = _keys(key)
        G = len(keys)
        if not self._wgrad_fits(geom, G) or (G > 1 and not self.wgrad_inplace):
            return False
        lib, st = L.lib(), L.current_stream()
        taps, M = kh * kw, B * Ho * Wo
        bf16 = self.precision == L.PREC_BF16
        Mp = (M + 63) // 64 * 64 if bf16 else (M + 31) // 32 * 32
        e = 2 if bf16 else 1                     # pixels per 4-byte element of an operand row
        # grouped: the rows of both operands are channel-major over ALL groups' channels, i.e. group g's rows are a contiguous run
        xt = self._empty(G * taps * ci, Mp // e)
        split = {L.PREC_FP32: 0, L.PREC_BF16X3: 1, L.PREC_BF16: 2}[self.precision]      # operands written in the GEMM's format directly
        if g_t is not None and bf16 and g_t[1] == Mp and g_t[0].numel() == G * co * Mp // e:
            gt = g_t[0]                           # written by the BatchNorm backward that produced g (vidc_bn_train_backward_t)
        else:
            if g is None:
                raise RuntimeError("wgrad %s: neither dY nor a matching dY^T" % (key,))
            gt = self._empty(G * co, Mp // e)
            L.check(lib.vidc_im2col_transposed(L.ptr(g), L.ptr(gt), B, Ho, Wo, G * co, _ld(g), Ho, Wo, 1, 1, 1, 0, Mp, split, st), "transpose dY")
        # rows of Xt in channel-major order (split + 4): the GEMM's output [co][ci*taps + tap] IS the OIHW weight gradient, written in place
        # (no staging buffer, no permute / copy launch)
        inplace = self.wgrad_inplace            # False: tap-major rows, staging buffer, permute / copy (A/B, tests)
        if bf16 and taps == 1 and stride == 1 and pad == 0 and x.bf is not None and self.xt_from_bf16:
            # 1x1 / stride 1: Xt is the plain transpose of x, taken from the bf16 operand copy the forward conv read (half the bytes, same bits)
            L.check(lib.vidc_transpose_bf16(L.ptr(x.bf), L.ptr(xt), M, G * ci, Mp, st), "transpose x (bf16)")
        elif bf16 and inplace and x.bf is not None and self.xt_from_bf16 and B * H * W * G * ci < (1 << 31):
            # any other geometry: the transposed im2col gathered from the bf16 copy as well (half the input bytes of the fp32 source)
            L.check(lib.vidc_im2col_transposed_bf16(L.ptr(x.bf), L.ptr(xt), B, H, W, G * ci, Ho, Wo, kh, kw, stride, pad, Mp, st), "im2col^T (bf16)")
        else:
            L.check(lib.vidc_im2col_transposed(L.ptr(x.t), L.ptr(xt), B, H, W, G * ci, x.ld, Ho, Wo, kh, kw, stride, pad, Mp, split | (4 if inplace else 0), st), "im2col^T")
        Mp //= e
        n_out = taps * ci
        gw = self._adjacent_base(self.grad, keys, ".weight")      # grouped: the G gradients are one contiguous run, group g at + g * co * n_out
        tmp = gw if inplace else self._empty(co, n_out)
        d = L.ConvDesc()
        d.x, d.w, d.y = L.ptr(gt), L.ptr(xt), L.ptr(tmp)
        d.scale1, d.shift1 = L.ptr(self._const(self._ones, n_out, 1.0)), L.ptr(self._const(self._zeros, n_out, 0.0))
        d.B, d.H, d.W, d.Cin, d.ldx = 1, 1, co, Mp, Mp
        d.Ho, d.Wo, d.Cout, d.ldy = 1, co, n_out, n_out
        d.KH, d.KW, d.stride, d.pad, d.flags = 1, 1, 1, 0, (L.X_PLANAR_GROUPS if G > 1 else 0)
        d.groups, d.splitk, d.precision, d.tile = G, 1, self.precision, 0
        d.x_gs, d.w_gs, d.y_gs, d.p_gs = (co * Mp if G > 1 else Mp), n_out * Mp, (co * n_out if G > 1 else n_out), (0 if G > 1 else n_out)
        self._plan(d, "gemm")
        L.check(lib.vidc_conv2d_bn_act(C.byref(d), st), "wgrad gemm")
        if not inplace:
            if taps == 1:
                gw.view(co, ci).copy_(tmp)
            else:
                L.check(lib.vidc_wgrad_permute(L.ptr(tmp), L.ptr(gw), co, ci, taps, st), "wgrad permute")
        return True

    def conv(self, x, key, stride=1, pad=0, relu=False, out=None):
        """nn.Conv2d (+ReLU when no BatchNorm sits in between, depth_completion.py:141-142).  Records its backward."""
        keys = _keys(key)                        # G > 1: the same layer of the three pyramids as one grouped launch (x, y: G channel slices)
        G = len(keys)
        w = self.param[keys[0] + ".weight"]
        bias = self.param.get(keys[0] + ".bias")
        if G > 1 and bias is not None:
            raise RuntimeError("grouped conv %s: a bias per group is not supported (the pyramids' convs have none)" % (keys,))
        co, ci, kh, kw = w.shape
        B, H, W, cx = x.t.shape
        if cx != G * ci:
            raise RuntimeError("conv %s: input has %d channels, expected %d x %d" % (keys[0], cx, G, ci))
        Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
        wp = self._pack(key, "f", w)
        if self.precision == L.PREC_BF16 and x.bf is None and ci % 64 == 0:      # the bf16 operand copy, made once per activation: every conv that
            x.bf = self._empty(B, H, W, G * ci // 2)                               # reads x uses it, and so does the weight-gradient transpose
            L.check(L.lib().vidc_cast_bf16(L.ptr(x.t), L.ptr(x.bf), B * H * W, G * ci, x.ld, L.current_stream()), "cast")
        y = Act(out if out is not None else self._empty(B, Ho, Wo, G * co))
        y.conv_out = not relu and self.dyt_fused
        # stride 1, no bias, bf16 operands, GEMM weight gradient: this conv's backward reads dY only as bf16 rows (dgrad) and as dY^T (wgrad)
        y.no_f32_grad = (y.conv_out and self.skip_f32_dy and self.precision == L.PREC_BF16 and stride == 1 and bias is None and co % 64 == 0 and
                         B * Ho * Wo < (1 << 31) and self._wgrad_fits((B, H, W, ci, Ho, Wo, co, kh, kw, stride, pad), G) and (G == 1 or self.wgrad_inplace))
        # (VIDC_TRAIN_CONV_STATS_MAX_M bounds the output rows it is used for: one partial per 32 rows makes the final reduction of a large
        #  map long -- 4800 partials at M = 153 600 -- but limiting it to 16 384 or 3 000 rows measured the same step time within 0.2 ms)
        if self.conv_stats and self.precision == L.PREC_BF16 and not relu and out is None and co % 32 == 0 and B * Ho * Wo <= self.conv_stats_max_m:
            y.stats = torch.empty(((B * Ho * Wo + 31) // 32) * 2 * G * co, dtype=torch.float64, device=self.device)
        self._conv_call(x.t, wp, bias if bias is not None else self._const(self._zeros, co, 0.0), y.t, kh, kw, stride, pad, relu, False, x_bf=x.bf, stats=y.stats,
                        groups=G)

        def backward():
            g, g_bf, g_t = y.grad, y.grad_bf, y.grad_t
            if g is None:
                # the BatchNorm backward skipped the fp32 dY (no_f32_grad): this closure must then find BOTH bf16 forms, in the geometry
                # its GEMMs expect -- anything else (a second consumer of y, a precision switched between forward and backward) would
                # read a stand-in tensor as if it were the gradient
                Mp_ = (B * Ho * Wo + 63) // 64 * 64
                if not (y.no_f32_grad and not relu and g_bf is not None and g_t is not None and self.precision == L.PREC_BF16 and
                        g_t[1] == Mp_ and g_t[0].numel() == G * co * Mp_ // 2):
                    raise RuntimeError("conv %s: no fp32 gradient and no matching bf16 forms of it (no_f32_grad invariant broken)" % (key,))
            if relu:                                         # y = relu(conv): mask first
                gm = self._empty(B, Ho, Wo, G * co)
                L.check(L.lib().vidc_relu_backward(L.ptr(g), L.ptr(y.t), L.ptr(gm), y.rows, G * co, _ld(g), y.ld, G * co, 0, L.current_stream()), "relu_bwd")
                g, g_bf, g_t = gm, None, None
            lib = L.lib()

            def weight_and_bias_gradient(g=g, g_t=g_t):     # (column sums for the bias); on the lane's side stream: see __init__
                if not self._wgrad_gemm(g, x, key, (B, H, W, ci, Ho, Wo, co, kh, kw, stride, pad), g_t=g_t):
                    if g is None:
                        raise RuntimeError("wgrad %s: the direct kernel needs the fp32 dY" % (key,))
                    for gi, k in enumerate(keys):                # (shapes beyond the GEMM's 32-bit limits: the direct kernel, group by group)
                        sc = self._scratch_bytes(lib.vidc_conv_wgrad_scratch_bytes(B, Ho, Wo, co, ci, kh, kw))
                        L.check(lib.vidc_conv_wgrad(L.ptr(g[..., gi * co:(gi + 1) * co]), L.ptr(x.t[..., gi * ci:(gi + 1) * ci]), L.ptr(self.grad[k + ".weight"]), B, H, W,
                                                    ci, x.ld, Ho, Wo, co, _ld(g), kh, kw, stride, pad, L.ptr(sc), L.current_stream()), "wgrad")
                if bias is not None:
                    L.check(lib.vidc_colsum(L.ptr(g), y.rows, co, _ld(g), L.ptr(self.grad[keys[0] + ".bias"]), L.ptr(self._train_scratch(y.rows, co)),
                                            L.current_stream()), "colsum")

            self._beside(weight_and_bias_gradient)
            if x.grad is False:                              # network input: no data gradient wanted
                return
            # data gradient: the conv kernel on flipped / transposed weights; a strided conv spreads dY over the input grid first
            wd = self._pack(key, "d", w)
            gz = g
            if stride > 1:
                gz, g_bf = self._empty(B, H, W, G * co), None
                L.check(lib.vidc_zero_stuff(L.ptr(g), L.ptr(gz), B, Ho, Wo, G * co, _ld(g), stride, H, W, L.current_stream()), "zero_stuff")
            acc = x.grad is not None
            if not acc:
                x.grad = self._empty(B, H, W, G * ci)
            if acc:
                x.grad_bf = x.grad_t = None                  # x.grad changes below
            # (gz None: the BatchNorm backward wrote dY as bf16 only -- y.t stands in for its geometry, the kernel reads g_bf)
            self._conv_call(gz if gz is not None else y.t, wd, self._const(self._zeros, ci, 0.0), x.grad, kh, kw, 1, kh - 1 - pad, False, acc, x_bf=g_bf,
                            groups=G)

        self._record(backward)
        return y

    # ---- BatchNorm (train mode) + ReLU ---------------------------------------------------------------------------------------------
    def bn(self, x, key, relu, out=None, residual=None):
        """BatchNorm2d in train mode (+ReLU).  residual: the Bottleneck tail in the same pass, y = relu?(bn(x) + residual) -- the value and
        the backward are those of bn(x, relu=False) followed by add(., residual, relu); the add's launch and its pass over the map go."""
        Cc = x.t.shape[-1]
        keys = _keys(key)            # G > 1: ONE BatchNorm launch over the G x C channels of a grouped tensor; the G parameter / gradient /
        y = Act(out if out is not None else torch.empty_like(x.t))      # running-statistics vectors are contiguous runs of the flat buffers
        mean, rstd = self._empty(Cc), self._empty(Cc)
        gamma, beta = self._adjacent_base(self.param, keys, ".weight"), self._adjacent_base(self.param, keys, ".bias")
        if gamma.numel() * len(keys) != Cc:
            raise RuntimeError("bn %s: %d channels for %d x %d parameters" % (keys[0], Cc, len(keys), gamma.numel()))
        run_mean, run_var = self._adjacent_base(self.buf, keys, ".running_mean"), self._adjacent_base(self.buf, keys, ".running_var")
        g_gamma, g_beta = self._adjacent_base(self.grad, keys, ".weight"), self._adjacent_base(self.grad, keys, ".bias")
        bf16 = self.precision == L.PREC_BF16 and Cc % 64 == 0
        if bf16:
            y.bf = self._empty(*x.t.shape[:-1], Cc // 2)
        if x.stats is not None and x.stats.numel() == ((x.rows + 31) // 32) * 2 * Cc and x.ld == Cc:
            # x is a conv's output and the conv's epilogue has written its channel sums: final reduction + apply pass only
            L.check(L.lib().vidc_bn_train_forward_stats(L.ptr(x.t), L.ptr(y.t), x.rows, Cc, x.ld, y.ld, L.ptr(gamma), L.ptr(beta), L.ptr(run_mean),
                                                        L.ptr(run_var), BN_EPS, BN_MOMENTUM, int(relu), L.ptr(mean), L.ptr(rstd),
                                                        L.ptr(y.bf) if y.bf is not None else None, L.ptr(residual.t) if residual is not None else None,
                                                        residual.ld if residual is not None else 0, L.ptr(x.stats), L.ptr(self._train_scratch(x.rows, Cc)),
                                                        L.current_stream()), "bn_forward (conv stats)")
            x.stats = None                    # (consumed; the buffer goes back to the allocator with the activation's other temporaries)
        else:
            L.check(L.lib().vidc_bn_train_forward_add(L.ptr(x.t), L.ptr(y.t), x.rows, Cc, x.ld, y.ld, L.ptr(gamma), L.ptr(beta), L.ptr(run_mean),
                                                      L.ptr(run_var), BN_EPS, BN_MOMENTUM, int(relu), L.ptr(mean), L.ptr(rstd),
                                                      L.ptr(y.bf) if y.bf is not None else None, L.ptr(residual.t) if residual is not None else None,
                                                      residual.ld if residual is not None else 0, L.ptr(self._train_scratch(x.rows, Cc)), L.current_stream()), "bn_forward")
        self._nbt += [self.buf[k + ".num_batches_tracked"] for k in keys]
        y_in = y               # what the BatchNorm part of the backward takes dy from
        if residual is not None:
            y_in = Act(y.t)    # (geometry only: its .grad is the masked gradient handed on by the add part)

        def add_backward():    # the add's backward (training.add): mask by the ReLU, hand the gradient to both summands
            g = y.grad
            if relu:
                gm = self._empty(*y.t.shape)
                L.check(L.lib().vidc_relu_backward(L.ptr(g), L.ptr(y.t), L.ptr(gm), y.rows, Cc, _ld(g), y.ld, Cc, 0, L.current_stream()), "relu_bwd")
                g = gm
            y_in.grad = g
            self._accumulate(residual, g)

        mask_relu = relu and residual is None      # with a residual the ReLU sits behind the sum: its mask is applied by add_backward

        def backward():
            if residual is not None:
                add_backward()
            acc = x.grad is not None
            skip_f32 = (not acc) and bf16 and x.conv_out and x.no_f32_grad and x.rows < (1 << 31)      # dY is read through its two bf16 forms only
            dx = self._empty(*x.t.shape) if acc else None
            target = dx if acc else (None if skip_f32 else self._empty(*x.t.shape))
            tbf = self._empty(*x.t.shape[:-1], Cc // 2) if (bf16 and not acc) else None
            # x = a conv's output: dx is that conv's dY, and its weight-gradient GEMM wants dY^T as bf16 rows [C][Mp] -- written here, by the
            # kernel that produces dY, instead of by a transpose launch of its own (one launch and one pass over dY less per conv)
            Mp = (x.rows + 63) // 64 * 64
            tbt = self._empty(Cc, Mp // 2) if (tbf is not None and x.conv_out and x.rows < (1 << 31)) else None
            L.check(L.lib().vidc_bn_train_backward_t(L.ptr(y_in.grad), L.ptr(x.t), L.ptr(y.t) if mask_relu else None, L.ptr(target) if target is not None else None,
                                                     x.rows, Cc, _ld(y_in.grad), x.ld, y.ld,
                                                     Cc, L.ptr(gamma), L.ptr(mean), L.ptr(rstd), L.ptr(g_gamma), L.ptr(g_beta),
                                                     L.ptr(tbf) if tbf is not None else None, L.ptr(tbt) if tbt is not None else None, Mp,
                                                     L.ptr(self._train_scratch(x.rows, Cc)), L.current_stream()), "bn_backward")
            if acc:
                self._accumulate(x, target)
            else:
                x.grad, x.grad_bf, x.grad_t = target, tbf, ((tbt, Mp) if tbt is not None else None)

        self._record(backward)
        return y

    def flush_counters(self):
        """Every BatchNorm's num_batches_tracked += 1 for the forward just recorded, in one launch (bookkeeping, not arithmetic)."""
        if self._nbt:
            torch._foreach_add_(self._nbt, 1)
        self._nbt = []

    def _accumulate(self, x, g):
        """x.grad += g (or = g when none yet; g is then shared, not copied: nothing writes it afterwards)."""
        if x.grad is None:
            x.grad = g
            return
        x.grad_bf = x.grad_t = None                          # the in-place sum below makes the bf16 copies of x.grad stale
        Cc = x.t.shape[-1]
        L.check(L.lib().vidc_relu_backward(L.ptr(g), None, L.ptr(x.grad), x.rows, Cc, _ld(g), 0, _ld(x.grad), 1, L.current_stream()), "accumulate")

    def add(self, a, b, relu, out=None):
        Cc = a.t.shape[-1]
        y = Act(out if out is not None else self._empty(*a.t.shape))
        if self.precision == L.PREC_BF16 and Cc % 64 == 0 and relu and self.add_bf16:      # a block output: the next block's convs read it as a bf16 operand
            y.bf = self._empty(*a.t.shape[:-1], Cc // 2)
        L.check(L.lib().vidc_add_rows_bf16(L.ptr(a.t), L.ptr(b.t), L.ptr(y.t), a.rows, Cc, a.ld, b.ld, y.ld, int(relu),
                                           L.ptr(y.bf) if y.bf is not None else None, L.current_stream()), "add")

        def backward():
            g = y.grad
            if relu:
                gm = self._empty(*a.t.shape)
                L.check(L.lib().vidc_relu_backward(L.ptr(g), L.ptr(y.t), L.ptr(gm), a.rows, Cc, _ld(g), y.ld, Cc, 0, L.current_stream()), "relu_bwd")
                g = gm
            self._accumulate(a, g)
            self._accumulate(b, g)

        self._record(backward)
        return y

    def maxpool(self, x):
        B, H, W, Cc = x.t.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = Act(self._empty(B, Ho, Wo, Cc))
        L.check(L.lib().vidc_maxpool3x3s2(L.ptr(x.t), L.ptr(y.t), B, H, W, Cc, x.ld, Cc, None, L.current_stream()), "maxpool")

        def backward():
            dx = self._empty(B, H, W, Cc)
            L.check(L.lib().vidc_maxpool3x3s2_backward(L.ptr(x.t), L.ptr(y.grad), L.ptr(dx), B, H, W, Cc, x.ld, _ld(y.grad), Cc, L.current_stream()), "maxpool_bwd")
            self._accumulate(x, dx)

        self._record(backward)
        return y

    def upsample(self, x, size):
        B, h, w, Cc = x.t.shape
        y = Act(self._empty(B, size[0], size[1], Cc))
        L.check(L.lib().vidc_upsample_bilinear_ac(L.ptr(x.t), L.ptr(y.t), B, h, w, Cc, x.ld, size[0], size[1], Cc, 0, None, L.current_stream()), "upsample")

        def backward():
            dx = self._empty(B, h, w, Cc)
            L.check(L.lib().vidc_upsample_bilinear_ac_backward(L.ptr(y.grad), L.ptr(dx), B, h, w, Cc, _ld(y.grad), Cc, size[0], size[1], L.current_stream()),
                    "upsample_bwd")
            self._accumulate(x, dx)

        self._record(backward)
        return y

    # ---- network walk (depth_completion.py:16-65, 154-165) ----------------------------------------------------------------------
    def _stem(self, x_nchw, p, out_channels=64):
        w = self.param[p + "conv1.conv1_1.weight"]
        B, cin, H, W = x_nchw.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = Act(self._empty(B, Ho, Wo, out_channels))
        L.check(L.lib().vidc_stem_conv3x3s2(L.ptr(x_nchw), L.ptr(w), L.ptr(y.t), B, cin, H, W, out_channels, out_channels, 1, None, 0, L.current_stream()), "stem")

        def backward():
            g = self._empty(B, Ho, Wo, out_channels)
            L.check(L.lib().vidc_relu_backward(L.ptr(y.grad), L.ptr(y.t), L.ptr(g), y.rows, out_channels, _ld(y.grad), out_channels, out_channels, 0,
                                               L.current_stream()), "relu_bwd")
            sc = self._scratch_bytes(L.lib().vidc_stem_wgrad_scratch_bytes(B, cin, H, W, out_channels))
            L.check(L.lib().vidc_stem_wgrad(L.ptr(g), L.ptr(x_nchw), L.ptr(self.grad[p + "conv1.conv1_1.weight"]), B, cin, H, W, out_channels, out_channels,
                                            L.ptr(sc), L.current_stream()), "stem_wgrad")

        self._record(backward)
        return y

    def _bottleneck(self, x, p, stride, project, out=None):
        """p: the block's parameter prefix, or one prefix per pyramid (grouped: x and the result hold the pyramids as channel slices)."""
        t = self.bn(self.conv(x, _cat(p, "conv1")), _cat(p, "bn1"), True)
        t = self.bn(self.conv(t, _cat(p, "conv2"), stride, 1), _cat(p, "bn2"), True)
        if not self.bn_add_fused:
            t = self.bn(self.conv(t, _cat(p, "conv3")), _cat(p, "bn3"), False)
            idn = self.bn(self.conv(x, _cat(p, "downsample.0"), stride, 0), _cat(p, "downsample.1"), False) if project else x
            return self.add(t, idn, True, out=out)
        t = self.conv(t, _cat(p, "conv3"))
        idn = self.bn(self.conv(x, _cat(p, "downsample.0"), stride, 0), _cat(p, "downsample.1"), False) if project else x
        return self.bn(t, _cat(p, "bn3"), True, out=out, residual=idn)      # relu(bn3(.) + identity) in the BatchNorm's apply pass

    def _stem_grouped(self, xs, ps, out_channels=64):
        """The three pyramids' stem convs (Cin 3, 3, 1: three small launches) into the channel slices of ONE [B, Ho, Wo, 3 x 64] tensor."""
        B, _c, H, W = xs[0].shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        G = len(ps)
        Ct = G * out_channels
        y = Act(self._empty(B, Ho, Wo, Ct))
        for g, (x_nchw, pp) in enumerate(zip(xs, ps)):
            L.check(L.lib().vidc_stem_conv3x3s2(L.ptr(x_nchw), L.ptr(self.param[pp + "conv1.conv1_1.weight"]), L.ptr(y.t[..., g * out_channels:]), B, x_nchw.shape[1],
                                                H, W, out_channels, Ct, 1, None, 0, L.current_stream()), "stem")

        def backward():
            g_ = self._empty(B, Ho, Wo, Ct)
            L.check(L.lib().vidc_relu_backward(L.ptr(y.grad), L.ptr(y.t), L.ptr(g_), y.rows, Ct, _ld(y.grad), Ct, Ct, 0, L.current_stream()), "relu_bwd")
            for g, (x_nchw, pp) in enumerate(zip(xs, ps)):
                cin = x_nchw.shape[1]
                sc = self._scratch_bytes(L.lib().vidc_stem_wgrad_scratch_bytes(B, cin, H, W, out_channels))
                L.check(L.lib().vidc_stem_wgrad(L.ptr(g_[..., g * out_channels:]), L.ptr(x_nchw), L.ptr(self.grad[pp + "conv1.conv1_1.weight"]), B, cin, H, W, out_channels,
                                                Ct, L.ptr(sc), L.current_stream()), "stem_wgrad")

        self._record(backward)
        return y

    def _pyramid(self, x_nchw, p, module, level_out):
        """ResNetPyramids.forward (depth_completion.py:55-65) in train mode; level l's output goes to `level_out[l]` (a channel slice of
        the concat buffer, depth_completion.py:151-152)."""
        grouped = not isinstance(p, str)      # p: one prefix per pyramid, x_nchw: their inputs; level_out[l]: the whole concat buffer of level l
        t = self._stem_grouped(x_nchw, p) if grouped else self._stem(x_nchw, p)
        t = self.bn(self.conv(t, _cat(p, "conv1.conv1_2"), 1, 1), _cat(p, "conv1.bn_2"), True)
        t = self.bn(self.conv(t, _cat(p, "conv1.conv1_3"), 1, 1), _cat(p, "conv1.bn1_3"), True)
        t = self.bn(t, _cat(p, "bn1"), True)
        t = self.maxpool(t)
        outs = []
        for li in range(1, 5):
            stage = getattr(module, "layer%d" % li)
            for bi, blk in enumerate(stage):
                last = bi == len(stage) - 1
                t = self._bottleneck(t, _cat(p, "layer%d.%d." % (li, bi)), blk.stride, blk.downsample is not None, out=level_out[li - 1] if last else None)
            outs.append(t)
        return outs

    def forward(self, image, normal, depth_in):
        """Train-mode forward; returns the predicted depth (B,1,H,W).  The tape of backward closures is left in self.tape."""
        self.tape, self._nbt = [], []
        self.repack()
        B, _, H, W = image.shape
        sizes = [((H - 1) // 2 + 1, (W - 1) // 2 + 1)]
        sizes[0] = ((sizes[0][0] - 1) // 2 + 1, (sizes[0][1] - 1) // 2 + 1)
        for _ in range(3):
            sizes.append(((sizes[-1][0] - 1) // 2 + 1, (sizes[-1][1] - 1) // 2 + 1))
        chans = [256, 512, 1024, 2048]
        cat = [self._empty(B, sizes[l][0], sizes[l][1], 3 * chans[l]) for l in range(4)]
        levels = [Act(c) for c in cat]
        subs = []
        # The three pyramids are independent until the decoder: each runs on its own HIP stream (forward here, backward in
        # loss_and_backward), so the fixed cost of one pyramid's ~100 small launches hides under the other two's -- in the captured graph
        # they are three parallel branches.  Scratch and split-K workspaces are per lane.  VIDC_TRAIN_STREAMS=1: one stream.
        main = torch.cuda.current_stream()
        multi = self.n_lanes > 1
        if self.grouped:
            # ONE chain of grouped launches for the three pyramids on the caller's stream: a third of the launches, each three times the
            # work; level l's last block writes the concat buffer directly and the decoder reads the very same activation object, so the
            # decoder's d(concat) IS the pyramids' level gradient (no slicing step)
            xs = [t.contiguous().float() for t in (image, normal, depth_in)]
            levels = self._pyramid(xs, tuple(n + "." for n in PYRAMIDS), getattr(self.cnn, PYRAMIDS[0]), cat)
        for pi, (name, x) in enumerate(() if self.grouped else (("resnet_rgb", image), ("resnet_normal", normal), ("resnet_depth", depth_in))):
            outs = [cat[l][..., pi * chans[l]:(pi + 1) * chans[l]] for l in range(4)]
            x = x.contiguous().float()
            if not multi:
                subs.append(self._pyramid(x, name + ".", getattr(self.cnn, name), outs))
                continue
            side = self._lane_streams()[pi]
            side.wait_stream(main)
            self._cur = pi + 1
            with torch.cuda.stream(side):
                subs.append(self._pyramid(x, name + ".", getattr(self.cnn, name), outs))
            self._cur = 0
        if multi and not self.grouped:
            for side in self._lane_streams():
                main.wait_stream(side)

        def split_level_grads():                     # runs (in the backward) once the decoder has produced d(concat): slices become the
            for l in range(4 if not self.grouped else 0):      # gradients of the three pyramids' level outputs (grouped: the same objects)
                for pi in range(3):
                    g = levels[l].grad[..., pi * chans[l]:(pi + 1) * chans[l]]
                    subs[pi][l].grad = g if subs[pi][l].grad is None else subs[pi][l].grad
        # NB: appended BEFORE the decoder ops, so it runs after all of them in the reversed tape; a level's slice is also fed by the next
        # stage of its pyramid, whose backward (later in the reversed order) accumulates into the same slice.
        split_level_grads._decoder_done = True       # everything recorded after this point (= run before it) is the decoder's backward
        self._record(split_level_grads)

        def branch(b):
            t = levels[b - 1]
            idx, target = 0, b
            for step in _BRANCH_PLAN[b]:
                q = "feature%d_upsamping.%d" % (b, idx)
                if step == "u":
                    target -= 1
                    t = self.upsample(t, sizes[target - 1])
                    idx += 1
                else:
                    k = step[0]
                    t = self.bn(self.conv(t, q, 1, k // 2), "feature%d_upsamping.%d" % (b, idx + 1), True)
                    idx += 3
            return t

        zs = []
        for b in (1, 2, 3, 4):                       # the four decoder branches are independent until z1 + z2 + z3 + z4: one lane each
            if not multi:
                zs.append(branch(b))
                continue
            side = self._lane_streams()[b - 1]
            side.wait_stream(main)
            self._cur = b
            with torch.cuda.stream(side):
                zs.append(branch(b))
            self._cur = 0
        if multi:
            for side in self._lane_streams():
                main.wait_stream(side)
        z = self.add(self.add(self.add(zs[0], zs[1], False), zs[2], False), zs[3], False)
        h = self.conv(z, "feature_concat.0", 1, 1, relu=True)
        # padded 1x1 head -> bilinear to (H, W) -> ReLU (depth_completion.py:143-147)
        w2, b2 = self.param["feature_concat.2.weight"], self.param["feature_concat.2.bias"]
        hh, hw_ = h.t.shape[1], h.t.shape[2]
        low = self._empty(B, 1, hh + 2, hw_ + 2)
        pred = self._empty(B, 1, H, W)
        L.check(L.lib().vidc_head_conv1x1_upsample(L.ptr(h.t), L.ptr(w2), L.ptr(b2), L.ptr(low), L.ptr(pred), B, hh, hw_, 192, h.ld, 1, 1, H, W, 1,
                                                   L.current_stream()), "head")
        self._pred_grad = None

        def head_backward():
            lib = L.lib()
            n = B * H * W
            g = self._empty(n)
            L.check(lib.vidc_relu_backward(L.ptr(self._pred_grad), L.ptr(pred), L.ptr(g), n // 4, 4, 4, 4, 4, 0, L.current_stream()), "relu_bwd")
            g_low = self._empty(B, hh + 2, hw_ + 2)
            L.check(lib.vidc_upsample_bilinear_ac_backward(L.ptr(g), L.ptr(g_low), B, hh + 2, hw_ + 2, 1, 1, 1, H, W, L.current_stream()), "upsample_bwd")
            h.grad = self._empty(B, hh, hw_, 192)
            sc = self._scratch_bytes(lib.vidc_head_backward_scratch_bytes(B, hh, hw_, 192))
            L.check(lib.vidc_head_backward(L.ptr(g_low), L.ptr(h.t), L.ptr(w2), L.ptr(h.grad), L.ptr(self.grad["feature_concat.2.weight"]),
                                           L.ptr(self.grad["feature_concat.2.bias"]), B, hh, hw_, 192, h.ld, 192, L.ptr(sc), L.current_stream()), "head_bwd")

        self._record(head_backward)
        self._pred = pred
        self.flush_counters()
        return pred

    def loss_and_backward(self, pred, depth_gt, stop_after_decoder=False):
        """network_run.py:163-173 + `total_loss.backward()`: fills the flat gradient buffer; returns the loss (0-dim fp64 GPU tensor)."""
        B, _, H, W = pred.shape
        n = pred.numel()
        gt = depth_gt.contiguous().float()
        loss = torch.zeros((), dtype=torch.float64, device=self.device)
        self._pred_grad = self._empty(n)
        terms = self._empty(n)
        sc = self._scratch_bytes((n // 512 + 64) * 8)
        L.check(L.lib().vidc_masked_l1_loss(L.ptr(pred), L.ptr(gt), n, H * W, L.ptr(loss), L.ptr(self._pred_grad), L.ptr(terms), L.ptr(sc), L.current_stream()),
                "loss")
        self._run_tape(stop_after_decoder)
        return loss

    def _run_tape(self, stop_after_decoder=False):
        """Runs the recorded backward closures, last first, each on the stream lane it was recorded on.  stop_after_decoder: return once
        the decoder's part is done (its last closure hands the level gradients to the pyramids), leaving the pyramids' closures in
        self.tape for a second call -- the caller starts the all-reduce of the decoder's gradients in between."""
        # Lifetime of cross-lane tensors: a closure owns the last references to activations / gradients that OTHER lanes still read or
        # write (levels[l].grad is allocated by a decoder branch on lane 1 and its channel slices are accumulated into by the pyramids on
        # lanes 2 and 3).  Dropping a closure right after it ran would hand such a block back to the allocating lane's pool while the
        # other lanes' kernels are still queued -- and inside a captured graph, where the three pyramids are parallel branches and a
        # freed block is reusable at once, the allocating lane's next temporary could overwrite it.  So every closure that has run is
        # kept until all lanes have joined the main stream at the end of the tape.  Cost: the backward's peak memory is the SUM of the
        # activation gradients and their bf16 / transposed forms instead of the live set (not measured; bounded by the gradient volume of
        # one backward, i.e. of the order of the forward's activations -- a few GB at batch 8, 320x240, of 288 GB);
        # a step that raises drops them at once (the except below), so a failed step does not pin them until the next good one.
        try:
            self._run_tape_inner(stop_after_decoder)
        except BaseException:
            self.tape, self._keepalive = [], []           # a failed step must not pin every activation gradient until the next good one
            raise

    def _run_tape_inner(self, stop_after_decoder):
        main = torch.cuda.current_stream()
        forked = []
        while self.tape:
            fn = self.tape.pop()
            self._keepalive.append(fn)
            lane = getattr(fn, "_lane", 0)
            if lane == 0 or self.n_lanes <= 1:
                for side in forked:
                    main.wait_stream(side)
                forked = []
                fn()
                if stop_after_decoder and getattr(fn, "_decoder_done", False):
                    self._join_wgrad(main)                # the decoder's weight gradients are complete before their all-reduce starts
                    return                                # (closures stay parked: the second call releases them after its join)
                continue
            side = self._lane_streams()[lane - 1]
            if side not in forked:
                side.wait_stream(main)                    # the work recorded before this lane's (decoder backward / loss) is on main
                forked.append(side)
            self._cur = lane
            with torch.cuda.stream(side):
                fn()
            self._cur = 0
        for side in forked:
            main.wait_stream(side)
        self._join_wgrad(main)
        self._keepalive = []                              # every lane has joined: nothing queued anywhere still touches these tensors

    @torch.no_grad()
    def forward_backward(self, image, normal, depth_in, depth_gt):
        for t in (image, normal, depth_in, depth_gt):
            if not t.is_cuda:
                raise RuntimeError("DepthCompletionTrainer takes GPU tensors only (no CPU fallback)")
        if not self.cnn.training:
            raise RuntimeError("call cnn.train() first (network_run.py:232): the trainer implements BatchNorm's train() mode")
        pred = self.forward(image.float(), normal.float(), depth_in.float())
        loss = self.loss_and_backward(pred, depth_gt)
        return loss, pred

    @torch.no_grad()
    def optimizer_step(self, reduced=False):
        """torch.optim.Adam.step over the flat buffers; gradients are summed over ranks first (frame-sharded batch) unless `step` has
        done that already, overlapped with the backward."""
        if not reduced:
            self.buckets.all_reduce(self.flat_g)
        self.step_count += 1
        L.check(L.lib().vidc_adam_step(L.ptr(self.flat_p), L.ptr(self.flat_g), L.ptr(self.m), L.ptr(self.v), self.flat_p.numel(), self.lr, self.betas[0],
                                       self.betas[1], self.eps, self.step_count, L.current_stream()), "adam")
        self._packed_fresh = False      # so are the trainer's own packed copies (re-made by the next forward's repack())
        self.cnn._invalidate()          # the inference programs' packed / BN-folded copies are stale now

    def _distributed(self):
        from .sharding import collectives_active
        return collectives_active()

    def step(self, image, normal, depth_in, depth_gt):
        """One `_run_training_iteration`: returns the loss (0-dim fp64 GPU tensor, this rank's frames).

        The forward + backward of a step is ~4000 launches from Python (~17 us of host time each: host-bound once the convs run in the
        bf16 modes), all with shape-static arguments, so from the third step of a given input shape on it is replayed as captured
        hipGraphs (VIDC_TRAIN_GRAPH=0: always eager).  The all-reduce and the Adam launch (its bias correction takes the step number
        by value) stay outside the graphs.  Across ranks the backward is cut where the decoder's part ends: the all-reduce of the
        decoder's gradients (59 % of the 1.24 GB) is started there and runs under the pyramids' backward, the rest follows."""
        multi = self._distributed()
        self._check_bindings()
        if self.use_graph:
            loss, waits = self._graphed_forward_backward(image, normal, depth_in, depth_gt, multi)
        else:
            loss, waits = self._eager_forward_backward(image, normal, depth_in, depth_gt, multi)
        for w in waits:
            w()
        self.optimizer_step(reduced=multi)
        self.last_loss = loss
        return loss

    @torch.no_grad()
    def _eager_forward_backward(self, image, normal, depth_in, depth_gt, multi):
        if not multi:
            return self.forward_backward(image, normal, depth_in, depth_gt)[0], []
        pred = self.forward(image.float(), normal.float(), depth_in.float())
        loss = self.loss_and_backward(pred, depth_gt, stop_after_decoder=True)
        waits = self.buckets.all_reduce_async(self.flat_g, self._dec_off, None)
        self._run_tape()
        waits += self.buckets.all_reduce_async(self.flat_g, 0, self._dec_off)
        return loss, waits

    def _graphed_forward_backward(self, image, normal, depth_in, depth_gt, multi=False):
        ins = (image, normal, depth_in, depth_gt)
        key = tuple((tuple(t.shape), t.dtype) for t in ins) + (multi,)
        ent = self._graphs.get(key)
        if ent is None:
            seen = self._graph_seen[key] = self._graph_seen.get(key, 0) + 1
            if seen <= 2:                     # eager: creates the packed-weight table, constants, scratch and split-K workspace
                return self._eager_forward_backward(*ins, multi)
            static = [t.clone() for t in ins]
            torch.cuda.synchronize()
            graph, rest = torch.cuda.CUDAGraph(), None
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):      # (RCCL's watchdog thread polls events while this thread captures)
                if multi:
                    pred = self.forward(static[0].float(), static[1].float(), static[2].float())
                    loss = self.loss_and_backward(pred, static[3], stop_after_decoder=True)
                else:
                    loss, pred = self.forward_backward(*static)
            if multi:                         # second graph: the pyramids' backward (same memory pool: it reads the first one's tensors)
                rest = torch.cuda.CUDAGraph()
                with torch.cuda.graph(rest, pool=graph.pool(), capture_error_mode="thread_local"):
                    self._run_tape()
            ent = self._graphs[key] = (graph, rest, static, loss, pred)
        graph, rest, static, loss, pred = ent
        for dst, src in zip(static, ins):
            dst.copy_(src)
        self._packed_fresh = True             # the graph starts with repack()
        graph.replay()
        waits = []
        if rest is not None:
            waits = self.buckets.all_reduce_async(self.flat_g, self._dec_off, None)
            rest.replay()
            waits += self.buckets.all_reduce_async(self.flat_g, 0, self._dec_off)
        return loss.clone(), waits
