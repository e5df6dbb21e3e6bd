#!/usr/bin/env python3
"""Headline benchmark: frames/s of the per-frame depth-completion hot path on N MI355X (BASELINE.json).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python bench.py --gpus N ...           no RANK in the environment: spawns the N ranks itself (a torch.distributed.run child process,
                                           started BEFORE this process touches the GPU) and exits with the child's code
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload = BASELINE.json configs[1]: synthetic 320x256 RGB + 200-point sparse depth, batch 1, plane mask fixed;
one "step" = one frame through warp -> surface-normal net -> plane block + enrichment -> depth-completion net
(RunDepthCompletion._call_cnn, main.py:261-298), inputs resident in HBM, seeded random-init weights.
Frames shard over ranks with no data-path collective (weak scaling: every rank runs K frames); RCCL only gathers
4 doubles per rank at the end.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
_T0 = time.perf_counter()


def _one_rank_job():
    gpus = [a.split("=", 1)[1] if "=" in a else (sys.argv[i + 1] if i + 1 < len(sys.argv) else "1") for i, a in enumerate(sys.argv) if a == "--gpus" or a.startswith("--gpus=")]
    return int(os.environ.get("WORLD_SIZE", "1")) == 1 and os.environ.get("VIDC_DIST_WORLD1", "0") != "1" and (not gpus or gpus[-1] == "1")


if "--train" in sys.argv and _one_rank_job():
    # The training step runs on five HIP streams (main + up to four lanes); the runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware
    # queues (default 4), so two lanes share a queue and their kernels serialise: 26.3-26.5 against 26.7-27.4 ms per step with 8 queues.
    # ONE rank without a process group only: across ranks the step is two captured graphs with the decoder's all-reduce in between, and
    # that form runs ~2x slower with more than 4 queues (50-57 against 27.3-28.3 ms; profiles/r4_train_side_stream_experiments.txt).
    # (The inference stream mode, three lanes + the caller's stream, is FASTER with the default 4: 362 against 311 frames/s -- not set there.)
    # Read by the runtime when it initialises, i.e. at the first device call of this process.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from vi_depth_completion_amd import sharding, synthetic as S   # noqa: E402

FLOPS_PER_FRAME = {(240, 320): 293.88e9, (256, 320): 311.63e9}   # SURVEY.md §8d, reference formulation, 2 FLOP/MAC
PEAK_F32_MFMA_TFLOPS = 157.3                                       # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
PEAK_BF16_MFMA_TFLOPS = 2500.0                                     # MI355X_MICROARCH.md: dense bf16 MFMA


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--width", type=int, default=320)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--pool", type=int, default=4, help="distinct synthetic frames cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=4)
    ap.add_argument("--per-op", type=str, default="", help="write the per-op timing table to this file")
    ap.add_argument("--source", default="", help="WxH of a raw camera stream (e.g. 640x480, 1280x720: BASELINE configs[2], [3]); every step "
                                                  "then starts from uint8 frames resident in HBM and includes the device-side pre-processing "
                                                  "(PIL-exact resize + ToTensor, sparse-point rasterisation) of SURVEY 8f-2")
    ap.add_argument("--plane-head", action="store_true",
                    help="BASELINE configs[2] 'full pipeline incl. plane_mask_detection head': the plane-instance maps come from the "
                         "R-101-FPN Mask R-CNN detector (plane_mask.PlaneMaskDetector, seeded weights) every frame instead of a fixed map")
    ap.add_argument("--mode", choices=("interleaved", "streams", "sequential"), default="interleaved",
                    help="interleaved: software pipeline over frames (pipeline.run_interleaved: tick t = surface-normal net of frame t "
                         "+ depth-completion net of frame t-1 as one 4-group program); streams: --in-flight frames on separate HIP "
                         "streams; sequential: back-to-back _call_cnn")
    ap.add_argument("--in-flight", type=int, default=2, help="frames executing concurrently in --mode streams")
    ap.add_argument("--lanes", type=int, default=0, help="--mode interleaved: software-pipelined frame streams on this many HIP streams, group p on lane "
                                                          "p mod L (pipeline.run_interleaved(lanes=L)); results are bit-identical for every L.  0 (default): 3 with "
                                                          "--frames-per-launch > 1 (both legs: DESIGN 4.5), else 2")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="--mode interleaved: this many consecutive items of the stream share every launch of a tick (pipeline.run_interleaved("
                         "frames_per_launch=F): the frame program is recorded for batch F x B; items stay --batch frames each, with their own "
                         "gravity, plane block and draws in _call_cnn order).  1 = one item per launch (rounds 1-3); 0 (default) = 4 for batch-1 items, 1 for "
                         "items that are batches themselves (--batch > 1: the measured tile table covers those program batches).  The throughput / latency "
                         "knob of the stream mode: 4 (default; ResNet-101 layer 3 at M = 1280 = exactly 5 workgroups per CU) measures 373-377 "
                         "frames/s in fp32 at 20 steps with the first item complete after 26 ms, 2 measures 361-364 with 15 ms (DESIGN 5.1)")
    ap.add_argument("--no-fp32-leg", action="store_true", help="only the mixed-mode leg (bf16x3 MFMA on the layers the measured table selects): it becomes "
                                                               "the headline, with its dtype in the line")
    ap.add_argument("--no-mixed-leg", action="store_true", help="only the headline leg (every conv in exact fp32 MFMA arithmetic)")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the short child-process runs of BASELINE configs[4] (training step, bf16) and "
                                                                 "configs[2] (640x480 stream, batch 8, plane head) that the default N=1 run appends as `extra_legs`")
    ap.add_argument("--extra-legs-budget", type=float, default=130.0, help="seconds of wall clock since the start of this process after which no further extra leg is started")
    ap.add_argument("--no-sequential-leg", action="store_true", help="skip the short back-to-back _call_cnn measurement (per-frame latency and the "
                                                                       "rate of the operator the reference's harness calls, network_run.py:294-296)")
    ap.add_argument("--sequential-frames", type=int, default=20)
    ap.add_argument("--cpu-threads", default="8,16,32,64,128", help="thread counts tried for the CPU baseline (one frame each); the best one runs --cpu-frames")
    ap.add_argument("--train", action="store_true",
                    help="BASELINE configs[4] instead of the inference path: one step = one `_run_training_iteration` (network_run.py:231-254) of "
                         "ModifiedFPN on --batch frames per GPU (train-mode BatchNorm, masked L1 / (H*W), Adam; fwd + dgrad + wgrad on the MFMA conv "
                         "kernel; VIDC_TRAIN_PRECISION=bf16 is the arithmetic the configuration names), gradients summed over ranks with a "
                         "bucketed RCCL all-reduce overlapped with the backward")
    ap.add_argument("--regions", type=int, default=5, help="the K-step timed region is run this many times back to back (each bracketed by barrier + "
                    "synchronize, max over ranks per region); `value` / `ms_per_step` are the MEDIAN region, all of them are listed in `regions` (SURVEY 8d: median of 5)")
    ap.add_argument("--steady-frames", type=int, default=0, help="frames of the untimed long stream whose device time stamps give `steady_state_frames_per_s` "
                    "(0 = 16 x lanes x frames per launch; negative = skip)")
    ap.add_argument("--launcher-selftest", action="store_true",
                    help="no GPU work: every rank only joins the process group and gathers a fake record (tests/test_bench_launcher.py runs "
                         "`bench.py --gpus 2 --launcher-selftest` under gloo on the CPU: spawn, rendezvous, gather, the n_gpus check)")
    args = ap.parse_args()
    if args.frames_per_launch <= 0:
        args.frames_per_launch = 4 if args.batch == 1 else 1
    return args


def launch_ranks(args):
    """`bench.py --gpus N` called without a launcher: start the N ranks as a CHILD process (torch.distributed.run --standalone, which
    picks its own rendezvous port on 127.0.0.1) and return its exit code.  This process makes no HIP call at all (the ranks check
    world size against the visible GPUs themselves and exit non-zero), and it never replaces itself with another program."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or args.gpus) // args.gpus)))
    # every rank's stderr is also kept in a file of its own (torchrun --log-dir / --tee 2; stdout stays untouched: rank 0's ONE JSON line):
    # a rank that dies -- a HIP initialisation error, a signal (faulthandler is on in the ranks) -- leaves its last words there even when
    # the launcher's own stream is cut; they are replayed below on failure
    log_dir = os.environ.get("VIDC_RANK_LOG_DIR") or os.path.join(os.environ.get("TMPDIR", "/tmp"), "vidc_rank_logs_%d" % os.getpid())
    os.makedirs(log_dir, exist_ok=True)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--log-dir", log_dir, "--tee", "2", os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        sys.stderr.write("bench.py: the %d-rank job exited with code %d; per-rank logs under %s\n" % (args.gpus, rc, log_dir))
        for root, _dirs, files in sorted(os.walk(log_dir)):
            for fn in sorted(files):
                if fn.endswith((".log", ".json")) or fn in ("stderr", "stdout", "error.json"):
                    try:
                        txt = open(os.path.join(root, fn), errors="replace").read()
                    except OSError:
                        continue
                    if txt.strip():
                        sys.stderr.write("---- %s (last 1500 chars) ----\n%s\n" % (os.path.relpath(os.path.join(root, fn), log_dir), txt[-1500:]))
    return rc


def launcher_selftest(args, rank, world):
    """The N > 1 plumbing without the GPU: group, gather, the rank-count check, ONE line from rank 0."""
    import torch.distributed as dist
    if os.environ.get("VIDC_SELFTEST_KILL_RANK") == str(rank):        # (tests: what the launcher does when a rank dies)
        sys.stderr.write("selftest: rank %d told to die\n" % rank)
        sys.stderr.flush()
        os._exit(17)
    rec = sharding.metric_record(args.steps, 1.0 + 0.5 * rank, 0.0, 0.0)
    got = sharding.gather_records(rec)
    assert got.shape[0] == world, "gathered %d records from a world of %d" % (got.shape[0], world)
    job = sharding.combine(got)
    if rank == 0:
        print(json.dumps({"metric": "launcher-selftest", "n_gpus": int(got.shape[0]), "frames": job["frames"], "seconds": job["seconds"],
                          "backend": (dist.get_backend() if world > 1 else "none")}), flush=True)


def build_pipeline(H, W, dev, plane_head=False):
    from vi_depth_completion_amd.pipeline import DepthCompletionPipeline, FixedPlaneMask
    cc = (0.5 * 319.87654 * W / 320.0, 0.5 * 239.87603 * H / 240.0)
    pipe = DepthCompletionPipeline(enriched_samples=200, cc_img=cc, device=dev, rng=np.random.RandomState(1234))
    sn_sd = S.seeded_state_dict(pipe.surface_normal_cnn.state_dict(), 1234, device=dev)
    dc_sd = S.seeded_state_dict(pipe.cnn.state_dict(), 1234, device=dev)
    pipe.load_state_dicts(sn_sd, dc_sd)
    pipe.plane_masks_extraction = FixedPlaneMask(S.plane_id_map(H, W))
    det_sd = None
    if plane_head:
        from vi_depth_completion_amd.plane_mask import PlaneMaskDetector
        det = PlaneMaskDetector(device=dev)
        det_sd = S.seeded_detector_state_dict(det.state_dict(), 1234, device=dev)
        det.load_state_dict(det_sd)
        pipe.plane_masks_extraction = det
    return pipe, sn_sd, dc_sd, cc, det_sd


def _sig_flops(name):
    m = re.search(r"M(\d+)_N(\d+)_K(\d+)_k\ds\d_G(\d+)", name)
    M, N, K, G = (int(v) for v in m.groups())
    return 2.0 * M * N * K * G


# vidc_conv_tile name -> template arguments <BM, BN, WM, WN, WK, NS> of conv_igemm_f32 (csrc/conv_mfma.hip kTiles): the kernel
# name rocprofv3 reports for that tiling is conv_igemm_f32<BM, BN, WM, WN, WK, NS, PREC>
TILE_TEMPLATE = {"128x128": (128, 128, 2, 2, 1, 2), "128x64": (128, 64, 2, 2, 1, 3), "64x128": (64, 128, 2, 2, 1, 3), "64x64": (64, 64, 2, 2, 1, 4),
                 "64x64k2": (64, 64, 2, 2, 2, 3), "32x64k2": (32, 64, 1, 2, 2, 3), "32x32k4": (32, 32, 1, 1, 4, 3), "32x128": (32, 128, 1, 4, 1, 4),
                 "32x32k8": (32, 32, 1, 1, 8, 2), "32x64k2d5": (32, 64, 1, 2, 2, 5), "32x32k4d4": (32, 32, 1, 1, 4, 4),
                 "32x128d6": (32, 128, 1, 4, 1, 6), "64x64k2d4": (64, 64, 2, 2, 2, 4),
                 # loader-wave variants: conv_igemm_f32<..., PREC, 1>
                 "32x64k2L": (32, 64, 1, 2, 2, 3), "32x64k2d5L": (32, 64, 1, 2, 2, 5), "32x32k4d4L": (32, 32, 1, 1, 4, 4), "64x64L": (64, 64, 2, 2, 1, 4),
                 "64x64k2d4L": (64, 64, 2, 2, 2, 4), "64x128L": (64, 128, 2, 2, 1, 3), "128x64L": (128, 64, 2, 2, 1, 3),
                 "64x32k2": (64, 32, 2, 1, 2, 3), "64x32k2d5": (64, 32, 2, 1, 2, 5), "64x32k2d5L": (64, 32, 2, 1, 2, 5),
                 "128x128d3": (128, 128, 2, 2, 1, 3), "128x128d3L": (128, 128, 2, 2, 1, 3), "256x128": (256, 128, 4, 2, 1, 3),
                 "128x256": (128, 256, 2, 4, 1, 3),
                 "32x64k2d2": (32, 64, 1, 2, 2, 2), "64x64d2": (64, 64, 2, 2, 1, 2), "32x32k4d2": (32, 32, 1, 1, 4, 2), "64x128d2": (64, 128, 2, 2, 1, 2),
                 "64x32k2d2": (64, 32, 2, 1, 2, 2),
                 # loader waves + pipelined fragment reads: conv_igemm_f32<..., PREC, 2>
                 "128x128d4P": (128, 128, 2, 2, 1, 4), "128x128d3P": (128, 128, 2, 2, 1, 3), "64x64d4P": (64, 64, 2, 2, 1, 4), "128x64d4P": (128, 64, 2, 2, 1, 4),
                 "64x64k2d4P": (64, 64, 2, 2, 2, 4), "64x32k2d5P": (64, 32, 2, 1, 2, 5), "32x64k2d5P": (32, 64, 1, 2, 2, 5)}


def kernel_name(tile, prec):
    if tile == "wino4f":          # Winograd F(4 x 4) in one launch (csrc/wfused.hip); <2, 0> = 32 output channels per workgroup (the full tick's form)
        return "wino4_fused_kernel<2, 0>"
    if tile in ("g96x32s", "g96x64s3"):          # the streamed few-row tiles of csrc/wgemm.hip (only behind VIDC_TUNING_OVERRIDE: not in the measured table)
        return "wgemm_stream_kernel<%s>" % ("1, 2" if tile == "g96x32s" else "2, 3")
    return "conv_igemm_f32<%s, %d, %d>" % (", ".join(str(v) for v in TILE_TEMPLATE[tile]), prec, 2 if tile.endswith("P") else int(tile.endswith("L")))


def conv_stack_times(prog, iters=5):
    """Per-op durations of one program execution: HIP events recorded between consecutive ops on the launch stream
    (eager issue, averaged over `iters`), rescaled by (hipGraph replay time of the program / eager time of the program):
    an event between every two launches costs ~1-2 us of GPU idle per op, which the production path (graph replay) does
    not pay.  Returns {kernel: (ms, launches, flops)} for the fused-conv launches grouped by kernel (= tiling x arithmetic
    mode = one template instantiation; a split-K launch includes its finalize kernel), the program's graph-replay time, and
    the (name, ms) table."""
    total, per = prog.time(iters=iters, use_graph=False, per_op=True)
    if prog.captured:
        graph_ms = prog.time(iters=20, use_graph=True)
        per = [t * graph_ms / total for t in per]
        total = graph_ms
    out = {}
    for n, t in zip(prog.op_names, per):
        if n.startswith("conv:"):
            _c, _key, tile, _sk, rest = n.split(":", 4)
            kern = (tile, "bf16x3" if rest.startswith("bf16x3 ") else "fp32")
            ms, cnt, fl = out.get(kern, (0.0, 0, 0.0))
            out[kern] = (ms + t, cnt + 1, fl + _sig_flops(n))
    return out, total, list(zip(prog.op_names, per))


def winograd_summary(prog, ops, fpt, frames_per_s, peak_tflops, ref_flops_per_tick):
    """How much of the conv stack runs in the Winograd domain, and the frame rate priced in DIRECT-form FLOPs: the 3x3 layers that run as
    F(m x m, 3x3) execute 2.25x - 4x fewer multiply-adds than their direct form, so frames/s x the reference formulation's FLOPs can exceed
    the MFMA peak; `conv_stack.*executed*` above always counts what the MFMA kernel really executes (transform-domain GEMMs)."""
    wl = [n for n, _t in ops if n.startswith("conv:") and "@wino" in n]
    t_in = sum(t for n, t in ops if n.startswith("wino_in"))
    t_out = sum(t for n, t in ops if n.startswith("wino_out"))
    t_gemm = sum(t for n, t in ops if n.startswith("conv:") and "@wino" in n)
    out = {"layers_per_tick": len(wl), "F4_layers": sum("@wino4" in n for n in wl), "F2_layers": sum("@wino2" in n for n in wl),
           "transform_ms_per_frame": round((t_in + t_out) / fpt, 4), "gemm_ms_per_frame": round(t_gemm / fpt, 4),
           "reference_formulation_tflops_at_measured_frame_rate": round(ref_flops_per_tick / fpt * frames_per_s / 1e12, 2),
           "reference_formulation_frac_of_peak": round(ref_flops_per_tick / fpt * frames_per_s / 1e12 / peak_tflops, 4),
           "note": "direct-form (reference formulation, 2 FLOP/MAC) conv FLOPs per frame x measured frames/s over the dense MFMA peak of the leg's "
                   "arithmetic: an ALGORITHMIC rate (can exceed 1: Winograd layers execute fewer multiplies); MFMA utilisation = conv_stack.at_measured_frame_rate"}
    if prog is not None:
        out["executed_over_direct_flops"] = round(prog.flops / max(1, prog.direct_flops), 4)
    return out


def measure(args, dev, rank, world, precision):
    """One leg: builds the pipeline in `precision` mode ("mixed" | "fp32": engine.precision_mode reads VIDC_PRECISION when a program is
    recorded), runs W untimed + K timed steps bracketed by barrier + synchronize, then (rank 0) the live roofline of what was timed."""
    import torch.distributed as dist
    os.environ["VIDC_PRECISION"] = precision
    H, W, B = args.height, args.width, args.batch
    lanes = args.lanes if args.lanes > 0 else (3 if (args.frames_per_launch > 1 and args.mode == "interleaved") else 2)
    pipe, sn_sd, dc_sd, cc, det_sd = build_pipeline(H, W, dev, args.plane_head)

    # frame f of the job is a function of (seed, f) only: rank r takes frames r, r+world, ... (round-robin shards)
    pool = []
    pre = None
    if args.source:
        from vi_depth_completion_amd.preprocess import FramePreprocessor
        sw_, sh_ = (int(v) for v in args.source.lower().split("x"))
        pre = FramePreprocessor(dev, in_hw=(sh_, sw_), out_hw=(H, W), cc=(S.DEMO_CC[0], S.DEMO_CC[1] * H / 240.0))
    for j in range(args.pool):
        if pre is not None:
            cam = S.synthetic_camera_batch(B, sh_, sw_, 1234, frame0=(rank + j * world) * B, out_hw=(H, W))
            cam["image_u8"] = cam["image_u8"].to(dev)            # the frames are resident in HBM; tracks / gravity are host data like in the reference
            pool.append(cam)
        else:
            b = S.synthetic_batch(B, H, W, 1234, frame0=(rank + j * world) * B)
            pool.append({k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()})

    def frames(n):
        """the n input batches of a run; with --source each one goes through the device-side pre-processing inside the timed region"""
        for i in range(n):
            item = pool[i % len(pool)]
            yield pre(item["image_u8"], item["gravity_raw"], item["klt_tracks"]) if pre is not None else item

    def run(n):
        """n steps (= n frames of batch B through the whole hot path); all n outputs are complete on return."""
        out = None
        if args.mode == "sequential" or (args.mode == "streams" and args.in_flight <= 1):
            for b_ in frames(n):
                out = pipe._call_cnn(b_)
        elif args.mode == "streams":
            for out in pipe.run_stream(frames(n), in_flight=args.in_flight):
                pass
        else:       # n frames = n + 1 pipeline ticks, all inside the timed region; outputs stay in the program's buffer (valid until the
            for out in pipe.run_interleaved(frames(n), copy_outputs=False, lanes=lanes, frames_per_launch=args.frames_per_launch):      # next item
                pass                                                                                                                      # is requested: documented lifetime)
        return out

    if args.mode == "interleaved":      # set-up like the weight load above: every lane's program recorded, captured and uploaded before the first step
        pipe.prepare_interleaved(next(iter(frames(1))), lanes=lanes, frames_per_launch=args.frames_per_launch)
    run(args.warmup)
    region_s = []
    for _r in range(max(1, args.regions)):      # every region: exactly K steps between barrier + synchronize on both sides
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        region_s.append(time.perf_counter() - t0)
    elapsed = sorted(region_s)[len(region_s) // 2]      # this rank's median region (roofline bookkeeping below); the line uses max over ranks per region

    # ---- steady state of the stream mode (outside the timed regions): one long stream, a device time stamp behind every item's result;
    #      rate between two group boundaries well inside it, i.e. without the fill and the drain a K-step region contains ------------------
    steady = None
    if rank == 0 and args.mode == "interleaved" and args.steady_frames >= 0:
        grp = lanes * args.frames_per_launch
        n_long = args.steady_frames if args.steady_frames > 0 else 16 * grp
        n_long = max(6 * grp, n_long // grp * grp)
        evs = []
        from vi_depth_completion_amd import ops as _ops
        stamps = _ops.clock_stamps(n_long, dev)
        for _out in pipe.run_interleaved(frames(n_long), copy_outputs=False, lanes=lanes, frames_per_launch=args.frames_per_launch):
            e = torch.cuda.Event(enable_timing=True)
            e.record()                                  # the caller's stream has just been made to wait for this item's result
            _ops.clock_stamp(stamps, len(evs))          # ... and a (shader cycles, 100 MHz ticks) stamp behind it: the clock under this load
            evs.append(e)
        torch.cuda.synchronize()
        i0, i1 = 2 * grp - 1, n_long - 2 * grp - 1      # last items of two groups, two groups away from either end
        ms = evs[i0].elapsed_time(evs[i1])
        ghz = _ops.shader_clock_ghz(stamps, i0, i1)
        steady = {"frames_per_s": round((i1 - i0) * B / (ms * 1e-3), 2), "frames": n_long, "shader_clock_ghz": (round(ghz, 3) if ghz else None),
                  "what": "device time stamps (HIP events on the caller's stream behind every item's result) of one untimed stream of %d items: items "
                          "%d..%d / elapsed device time -- the K-step regions above additionally contain the fill and the drain of the pipeline" % (n_long, i0, i1)}

    # ---- latency of the first item of a stream in the mode that was timed (outside the timed region): host time from the first request to
    #      the first depth map being complete on the device; with --frames-per-launch F the first F items finish together ------------------
    first_item_ms = None
    if rank == 0 and args.mode == "interleaved":
        lat = []
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            gen = pipe.run_interleaved(frames(2 * lanes * args.frames_per_launch), copy_outputs=False, lanes=lanes, frames_per_launch=args.frames_per_launch)
            next(gen)
            torch.cuda.current_stream().synchronize()
            lat.append(time.perf_counter() - t1)
            for _o in gen:
                pass
        torch.cuda.synchronize()
        first_item_ms = round(1e3 * sorted(lat)[1], 3)

    # ---- the operator the reference's harness calls, frame after frame (network_run.py:294-296: one _call_cnn per batch, its output
    #      consumed before the next call): per-frame latency and the rate with ONE frame in flight.  Outside the timed region. ----------
    sequential = None
    if rank == 0 and not args.no_sequential_leg and args.sequential_frames > 0:
        it = frames(args.sequential_frames + 3)
        for _ in range(3):
            pipe._call_cnn(next(it))
        torch.cuda.synchronize()
        lat = []
        for b_ in it:
            t1 = time.perf_counter()
            pipe._call_cnn(b_)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        lat.sort()
        sequential = {"what": "%d back-to-back _call_cnn calls, host synchronised after each (one frame in flight)" % len(lat),
                      "frames_per_s": round(len(lat) * B / sum(lat), 2), "latency_ms_median": round(1e3 * lat[len(lat) // 2], 3),
                      "latency_ms_max": round(1e3 * lat[-1], 3)}

    # ---- roofline of the dominant kernel (fused conv), measured live with HIP events -- before the
    #      CPU baseline, whose OpenMP workers keep spinning and would slow the launching thread ---------------------
    roofline = None
    extra = {}
    if rank == 0:
        fpt = B * (args.frames_per_launch if args.mode == "interleaved" else 1)      # frames per execution of the program measured below
        if args.mode == "interleaved":     # the program that was timed: one tick = both networks, 4-group pyramid launches, F items per launch
            fp = pipe.frame_program(fpt, H, W)
            sn_t, sn_total, sn_ops = conv_stack_times(fp)
            dc_t, dc_total, dc_ops = {}, 0.0, []
        else:
            sn_prog = pipe.surface_normal_cnn.program(B, dev)
            dc_prog = pipe.cnn.program(B, H, W, dev)
            sn_t, sn_total, sn_ops = conv_stack_times(sn_prog)
            dc_t, dc_total, dc_ops = conv_stack_times(dc_prog)
        flops = FLOPS_PER_FRAME.get((H, W), 293.88e9 * H * W / (240.0 * 320.0)) * fpt
        kernels = {}
        for d in (sn_t, dc_t):
            for k, (ms, cnt, fl) in d.items():
                a0, a1, a2 = kernels.get(k, (0.0, 0, 0.0))
                kernels[k] = (a0 + ms, a1 + cnt, a2 + fl)
        conv_ms = sum(v[0] for v in kernels.values())
        conv_flops = sum(v[2] for v in kernels.values())
        n_launch = sum(v[1] for v in kernels.values())

        def roof(kern):
            """`achieved` = the kernel's own 2*M*N*K FLOPs per launch / its average launch duration (HIP events, this process).
            bf16x3 executes 3 bf16 MFMA products per fp32-equivalent product: `frac` prices the fp32-equivalent FLOPs against
            the dense bf16 peak, `frac_executed` the executed ones."""
            tile, mode = kern
            ms, cnt, fl = kernels[kern]
            ach = fl / (ms * 1e-3) / 1e12
            prec = 1 if mode == "bf16x3" else 0
            peak = PEAK_BF16_MFMA_TFLOPS if prec else PEAK_F32_MFMA_TFLOPS
            r = {"bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                 "kernel": kernel_name(tile, prec),
                 "arithmetic": ("bf16x3: 3 x v_mfma_f32_32x32x16_bf16 per fp32-equivalent product" if prec else "v_mfma_f32_32x32x2_f32"),
                 "launches_per_frame": round(cnt / fpt, 2), "launches_per_tick": cnt, "frames_per_tick": fpt,
                 "avg_launch_us": round(1e3 * ms / cnt, 2), "gflop_per_launch": round(fl / cnt / 1e9, 3),
                 "ms_per_frame": round(ms / fpt, 3),
                 "timing_note": "per-launch durations are taken on ONE stream (the frame program alone, HIP events between ops, rescaled to its graph replay time); "
                                "with several lanes launches of the streams overlap and a kernel trace of the run shows longer per-kernel durations: "
                                "profiles/r6_kernel_stats_fp32.csv / r6_kernel_stats_mixed.csv (--lanes 1) are the traces these numbers agree with",
                 "traffic_note": "no PMC pass on file for this instantiation in profiles/pmc_traffic.json (tools/evidence_r5.sh collects them on "
                                 "tools/frame_replay.py: rocprofv3 --pmc on the whole bench process segfaults in rocprofv3 on this pool)"}
            if prec:
                r.update(executed_tflops=round(3 * ach, 2), frac_executed=round(3 * ach / peak, 4))
            return r

        dominant = max(kernels, key=lambda k: kernels[k][0])
        roofline = roof(dominant)
        # HBM-side bytes per launch of that instantiation, if a PMC pass on it is on file (profiles/pmc_traffic.json)
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json"))).get(roofline["kernel"])
        except (OSError, ValueError):
            pmc = None
        if pmc:
            roofline["traffic"] = pmc["traffic_bytes"]
            roofline["mfma_busy_pmc"] = pmc.get("mfma_busy_fraction")       # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs), same passes
            roofline["traffic_note"] = ("bytes per launch = FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE from rocprofv3 --pmc passes: %s; %s; "
                                        "algorithmic bytes of that launch %d (profiles/pmc_traffic.json; --pmc on the whole bench process "
                                        "segfaults in rocprofv3 on this pool)" % (pmc["command"], pmc["shape"], pmc["algorithmic_bytes"]))
        ranked = sorted(kernels, key=lambda k: -kernels[k][0])
        extra = {"program_ms": ({"frame_program_tick": round(sn_total, 3), "frames_per_tick": fpt} if args.mode == "interleaved" else
                                {"surface_normal": round(sn_total, 3), "depth_completion": round(dc_total, 3)}),
                 "conv_ms_per_frame": round(conv_ms / fpt, 3), "conv_launches_per_frame": round(n_launch / fpt, 2), "conv_launches_per_tick": n_launch,
                 "first_item_latency_ms": first_item_ms, "lanes": (lanes if args.mode == "interleaved" else 1),
                 "conv_stack": {"executed_gflop_per_frame": round(conv_flops / fpt / 1e9, 2),
                                "tflops_fp32_equivalent": round(conv_flops / (conv_ms * 1e-3) / 1e12, 2),
                                "tflops_bf16_executed": round(sum(v[2] * (3 if k[1] == "bf16x3" else 1) for k, v in kernels.items()) / (conv_ms * 1e-3) / 1e12, 2),
                                "note": "over the conv time of ONE frame program alone (single stream)",
                                # the same FLOPs at the frame rate of the timed region of this rank (with --lanes 2 the launches of two
                                # frame programs overlap, so the chip executes more per second than one program's own conv time implies)
                                "at_measured_frame_rate": {
                                    "tflops_fp32_equivalent": round(conv_flops / fpt * (args.steps * B / elapsed) / 1e12, 2),
                                    "tflops_executed": round(sum(v[2] * (3 if k[1] == "bf16x3" else 1) for k, v in kernels.items()) / fpt * (args.steps * B / elapsed) / 1e12, 2),
                                    "frac_of_peak_executed": round(sum(v[2] * (3 if k[1] == "bf16x3" else 1) for k, v in kernels.items()) / fpt * (args.steps * B / elapsed) / 1e12
                                                                   / (PEAK_BF16_MFMA_TFLOPS if precision == "mixed" else PEAK_F32_MFMA_TFLOPS), 4)}},
                 "winograd": winograd_summary(fp if args.mode == "interleaved" else None, (sn_ops + dc_ops), fpt, args.steps * B / elapsed,
                                              PEAK_BF16_MFMA_TFLOPS if precision == "mixed" else PEAK_F32_MFMA_TFLOPS, flops),
                 "reference_formulation_gflop_per_frame": round(flops / fpt / 1e9, 2),
                 "conv_stack_tflops_reference_formulation": round(flops / (conv_ms * 1e-3) / 1e12, 2),
                 "precision_mode": precision, "sequential_call_cnn": sequential,
                 "other_conv_kernels": [roof(k) for k in ranked[1:4]]}
        for r in extra["other_conv_kernels"]:
            r.pop("traffic_note", None)
        if args.per_op:
            with open(args.per_op if precision == "mixed" else args.per_op + ".fp32", "w") as f:
                for name, ops in ((("frame_program" if args.mode == "interleaved" else "surface_normal"), sn_ops), ("depth_completion", dc_ops)):
                    for n, t in ops:
                        f.write("%s\t%.2f\t%s\n" % (name, t * 1e3, n))
    return {"elapsed": elapsed, "region_s": region_s, "steady": steady, "pipe": pipe, "lanes": lanes, "sn_sd": sn_sd, "dc_sd": dc_sd, "cc": cc, "det_sd": det_sd, "roofline": roofline, "extra": extra,
            "frames": frames, "pre": pre}


def train_leg(args, dev, rank, world):
    """`--train`: the training step of BASELINE configs[4] under the same contract (W untimed + K timed steps between barriers, max over
    ranks, ONE line from rank 0).  Weak scaling: every rank trains on --batch of its own frames; the data-path collective is the
    gradient all-reduce (vi_depth_completion_amd/training.py)."""
    import torch.distributed as dist
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    B, H, W = args.batch, args.height, args.width
    cnn = ModifiedFPN().to(dev)
    cnn.load_state_dict(S.seeded_state_dict(cnn.state_dict(), 1234, device=dev))
    cnn.train()
    tr = DepthCompletionTrainer(cnn, 1e-4)
    b = S.synthetic_batch(B, H, W, 1234, frame0=rank * B)
    image = b["image"].to(dev)
    normal = torch.nn.functional.normalize(image - 0.5, dim=1)
    depth_in = b["sparse_depth"].to(dev)
    gt = S.synthetic_ground_truth_depth(b["image"], 1234).to(dev)
    losses = [tr.step(image, normal, depth_in, gt) for _ in range(max(args.warmup, 3))]      # (steps 1-2 eager, 3rd captures the hipGraphs)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(tr.step(image, normal, depth_in, gt))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    rec = sharding.metric_record(args.steps * B, time.perf_counter() - t0, device=dev)
    got = sharding.gather_records(rec)
    assert got.shape[0] == world
    job = sharding.combine(got)
    if rank == 0:
        mode = os.environ.get("VIDC_TRAIN_PRECISION", "fp32")
        # forward + data gradient + weight gradient of every conv: 3 x the forward MACs of ModifiedFPN (247.58 GFLOP at 320x240)
        gflop = 3.0 * 247.58 * (H * W) / (240.0 * 320.0) * B
        tf = gflop / 1e3 / (job["seconds"] / args.steps)
        peak = PEAK_F32_MFMA_TFLOPS if mode == "fp32" else PEAK_BF16_MFMA_TFLOPS
        print(json.dumps({
            "metric": "training frames/sec", "value": round(job["frames_per_s"], 3), "unit": "frames/s", "n_gpus": int(got.shape[0]), "steps": args.steps,
            "warmup": max(args.warmup, 3), "ms_per_step": round(1e3 * job["seconds"] / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": {"fp32": "f32", "bf16x3": "f32+bf16x3", "bf16": "bf16"}[mode], "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: ModifiedFPN training step (train-mode BatchNorm, masked L1 / (H*W), Adam), %dx%d, batch %d per "
                                   "GPU, seeded weights and ground truth" % (W, H, B), "height": H, "width": W, "batch_per_gpu": B,
                       "sharding": "frames over %d rank(s); gradients summed by a bucketed all-reduce overlapped with the backward" % world,
                       "gradient_buckets": ("bf16" if os.environ.get("VIDC_TRAIN_GRAD_BF16", "0") == "1" else "f32"),
                       "collectives": ("through the backend" if sharding.collectives_active() else "none (one rank)")},
            "roofline": {"bound": "mfma", "achieved": round(tf * (3 if mode == "bf16x3" else 1), 2), "peak": peak, "unit": "TFLOP/s",
                         "frac": round(tf * (3 if mode == "bf16x3" else 1) / peak, 4), "traffic": None,
                         "note": "whole step: 3 x forward conv FLOPs per frame / step time (fwd + dgrad + wgrad launches of the conv kernel; "
                                 "BatchNorm, pooling, loss and Adam are HBM-bound and inside the same time)"},
            "cpu_baseline": None, "losses": [round(float(x), 6) for x in losses[-min(len(losses), 6):]]}), flush=True)


def main():
    args = parse()
    if "RANK" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))              # nothing below runs in the launching process
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import faulthandler
        faulthandler.enable()                     # a rank killed by SIGSEGV / SIGBUS / SIGABRT writes its Python stack to stderr (the per-rank log)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    backend = os.environ.get("VIDC_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" to try the N > 1 path on a box with fewer GPUs
    if args.launcher_selftest:
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        launcher_selftest(args, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("bench.py: no GPU visible (the HIP path has no CPU fallback)")
    if world > n_dev and backend == "nccl":
        raise SystemExit("bench.py: %d ranks on %d GPU(s): one rank per GPU (set VIDC_DIST_BACKEND=gloo to share GPUs on purpose)" % (world, n_dev))
    local = int(os.environ.get("LOCAL_RANK", "0")) % n_dev     # (the modulo only matters for the gloo try-out on a smaller box)
    if world > 1:
        # the ranks of a node open the GPU driver one after the other (0.25 s apart, outside every timed region): round 4 saw one rank of a
        # two-rank job die within seconds of starting as the first GPU processes of a fresh box, before any launch of this package
        time.sleep(0.25 * int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.set_grad_enabled(False)
    if not args.train:
        # the stream mode's lanes take their hardware queues before RCCL makes its streams (else a lane shares the caller's queue: 324 instead
        # of 363 frames/s in fp32 with a process group present -- the N > 1 runs would read as a scaling loss that is none)
        from vi_depth_completion_amd.pipeline import reserve_lane_streams
        reserve_lane_streams(dev, args.lanes if args.lanes > 0 else 3)
    if world > 1:
        # N ranks share the host: keep every rank's torch CPU pool (synthetic inputs are generated on the CPU before the timed region)
        # to its share of the cores, so that N pools of spinning OpenMP workers do not slow the N launching threads down
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
        import torch.distributed as dist
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    elif os.environ.get("VIDC_DIST_WORLD1", "0") == "1":
        # a world of ONE that still sends every collective through the backend (sharding.collectives_active): the RCCL code paths of the
        # N-GPU job -- metric gather, the training step's bucketed gradient all-reduce -- on a one-GPU box
        import socket
        import torch.distributed as dist
        if "MASTER_PORT" not in os.environ:
            sock = socket.socket()
            sock.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
            sock.close()
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": dev} if backend == "nccl" else {}))

    if args.train:
        if args.height == 256 and "--height" not in " ".join(sys.argv):
            args.height = 240                      # the training fixtures and BASELINE configs[4] use the reference's own 320x240
        train_leg(args, dev, rank, world)
        if world > 1 or os.environ.get("VIDC_DIST_WORLD1", "0") == "1":
            dist.barrier()
            dist.destroy_process_group()
        return
    H, W, B = args.height, args.width, args.batch
    # Two legs over the same steps, both inside this process (no re-exec).  The HEADLINE leg runs every conv in the reference's own
    # arithmetic (fp32 MFMA: exact fp32 products and sums) -- value, ms_per_step, dtype "f32", rmse_vs_oracle and roofline describe it.
    # The mixed mode (bf16x3 MFMA on the layers the measured table selects; narrower products, RMSE ~1e-5 against the 1e-3 bar) is
    # reported beside it as value_mixed / dtype_mixed / roofline_mixed / mixed_leg.
    if os.environ.get("VIDC_PRECISION") in ("fp32", "mixed"):
        legs = [os.environ["VIDC_PRECISION"]]                  # an explicit mode: that leg alone
    else:
        legs = (["fp32"] if not args.no_fp32_leg else []) + (["mixed"] if not args.no_mixed_leg else [])
        if not legs:
            raise SystemExit("bench.py: --no-fp32-leg and --no-mixed-leg leave nothing to run")
    main_mode = legs[0]
    res = {}
    for mode in legs:
        res[mode] = measure(args, dev, rank, world, mode)      # (both pipelines stay resident: ~6 GB of 288)
    lead = res[main_mode]
    cc = lead["cc"]

    # ---- parity of what was just timed: frame 0 of this rank against the CPU oracle (outside the timed region) ----
    recs = {m: torch.zeros(4, dtype=torch.float64, device=dev) for m in legs}   # frames, seconds, sum sq err, n px
    for m in legs:
        recs[m][0], recs[m][1] = args.steps * B, res[m]["elapsed"]
    n_reg = len(lead["region_s"])
    cpu_baseline = None
    if rank == 0 and not args.no_cpu_baseline:
        from oracle import vidc_oracle as O
        intr = O.Intrinsics(202.0, 202.0, cc[0], cc[1])
        cpu_sn = {k: v.cpu() for k, v in lead["sn_sd"].items()}
        cpu_dc = {k: v.cpu() for k, v in lead["dc_sd"].items()}
        hb = S.synthetic_batch(B, H, W, 1234, frame0=rank * B)
        if args.plane_head:      # the oracle's own detector (CPU) on the same frames
            from oracle import plane_mask_oracle as PM
            cpu_det = {k: v.cpu() for k, v in lead["det_sd"].items()}
            plane_masks = lambda batch: [PM.run_on_tensor(cpu_det, batch["image"][i]) for i in range(B)]
        else:
            plane_masks = lambda batch: [S.plane_id_map(H, W)] * B
        masks = plane_masks(hb)
        ref = O.call_cnn(cpu_sn, cpu_dc, hb, masks, intr, 200, rng=np.random.RandomState(77))   # also the warm-up
        for m in legs:
            # through the mode that was timed: the frame as the first item of a stream (with --frames-per-launch F it rides in the
            # batch-F program next to a partner frame, which does not enter its result), draws from a generator in the oracle's state
            pipe = res[m]["pipe"]
            os.environ["VIDC_PRECISION"] = m
            pipe.rng = np.random.RandomState(77)
            dev_hb = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in hb.items()}
            if args.mode == "interleaved":
                partner = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(B, H, W, 1234, frame0=(rank + world) * B).items()}
                got = list(pipe.run_interleaved(iter([dev_hb, partner]), lanes=res[m]["lanes"], frames_per_launch=args.frames_per_launch))[0].cpu()
            else:
                got = pipe._call_cnn(dev_hb).cpu()
            recs[m][2], recs[m][3] = float((got - ref).double().pow(2).sum()), float(ref.numel())
        if world == 1:      # the CPU baseline is timed at N=1 only (the other ranks would idle in the gather)
            # batch-1 convs do not scale to every host thread (0.26 frames/s on 128 threads in round 2 against 0.93 on 8 in the survey
            # container): one frame per candidate thread count, then --cpu-frames frames at the best one
            n_cpu = os.cpu_count() or 1
            cands = sorted({min(int(v), n_cpu) for v in args.cpu_threads.split(",") if v.strip()})
            sweep = {}
            for nt in cands:
                torch.set_num_threads(nt)
                hbj = S.synthetic_batch(B, H, W, 1234, frame0=B)
                tc = time.perf_counter()
                O.call_cnn(cpu_sn, cpu_dc, hbj, plane_masks(hbj), intr, 200, rng=np.random.RandomState(0))
                sweep[nt] = time.perf_counter() - tc
            best_nt = min(sweep, key=sweep.get)
            torch.set_num_threads(best_nt)
            tc = time.perf_counter()
            for j in range(args.cpu_frames):
                hbj = S.synthetic_batch(B, H, W, 1234, frame0=(j + 1) * B)
                O.call_cnn(cpu_sn, cpu_dc, hbj, plane_masks(hbj), intr, 200, rng=np.random.RandomState(j))
            cpu_s = time.perf_counter() - tc
            cpu_baseline = {"value": round(args.cpu_frames * B / cpu_s, 4), "unit": "frames/s", "cores": best_nt, "kind": "port",
                            "sample": "%d frames of the same %dx%d batch-%d workload through oracle/vidc_oracle.call_cnn (torch CPU fp32) on %d "
                                      "threads of %d host CPUs -- the best of a one-frame sweep over thread counts" % (
                                          args.cpu_frames, W, H, B, best_nt, n_cpu),
                            "thread_sweep_frames_per_s": {str(k): round(B / v, 4) for k, v in sorted(sweep.items())}}

    gathered = {m: sharding.gather_records(recs[m]) for m in legs}    # the only collective: 4 doubles per rank and leg over RCCL/xGMI
    for m in legs:
        assert gathered[m].shape[0] == world, "gathered %d records from a world of %d ranks" % (gathered[m].shape[0], world)
    jobs = {m: sharding.combine(gathered[m]) for m in legs}
    # per region: frames of all ranks / max over ranks of that region's time (one more 4-double gather per region and leg, outside every timed region)
    region_fps = {}
    for m in legs:
        fps = []
        for r_ in range(n_reg):
            rr = torch.zeros(4, dtype=torch.float64, device=dev)
            rr[0], rr[1] = args.steps * B, res[m]["region_s"][r_]
            jr = sharding.combine(sharding.gather_records(rr))
            fps.append((jr["frames"] / jr["seconds"], jr["seconds"]))
        region_fps[m] = fps
    extra_legs = None
    if rank == 0 and world == 1 and not args.no_extra_legs and not args.source and not args.plane_head and B == 1 and args.mode == "interleaved":
        torch.cuda.synchronize()
        extra_legs = run_extra_legs(args)
    if rank == 0:
        job = jobs[main_mode]
        med = sorted(region_fps[main_mode])[n_reg // 2]          # the median region: (frames/s over all ranks, max-over-ranks seconds)
        F = args.frames_per_launch if args.mode == "interleaved" else 1
        spread = (max(v[0] for v in region_fps[main_mode]) - min(v[0] for v in region_fps[main_mode])) / med[0]
        line = {
            "metric": "frames/sec", "value": round(med[0], 3), "unit": "frames/s", "n_gpus": int(gathered[main_mode].shape[0]), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * med[1] / args.steps, 4), "higher_is_better": True,
            "regions": [round(v[0], 3) for v in region_fps[main_mode]], "regions_spread": round(spread, 4),
            "regions_note": "%d back-to-back timed regions of exactly %d steps each (barrier + synchronize on both sides, max over ranks per region); "
                            "value / ms_per_step = the median region" % (n_reg, args.steps),
            "steady_state_frames_per_s": (lead["steady"] or {}).get("frames_per_s"), "steady_state": lead["steady"],
            "scaling": "weak", "vs_baseline": None,
            "dtype": ("f32" if main_mode == "fp32" else "f32+bf16x3"), "data": "synthetic",
            "config": {"workload": (("BASELINE configs[1]: synthetic %dx%d RGB + 200-pt sparse depth, batch %d per GPU, plane mask "
                                    "%s; warp + surface-normal net + plane block/enrichment + depth-completion net" % (
                                        W, H, B, "from the Mask R-CNN plane head every frame" if args.plane_head else "fixed")) if not args.source else
                                   ("%s uint8 camera stream + 200 VI-SLAM tracks per frame, batch %d per GPU, plane mask %s; device-side "
                                    "pre-processing (PIL-exact resize to %dx%d, rasterisation) + warp + surface-normal net + plane "
                                    "block/enrichment + depth-completion net" % (args.source, B, "from the Mask R-CNN plane head every frame" if args.plane_head else "fixed", W, H))),
                       "height": H, "width": W, "batch_per_gpu": B, "weights": "seeded random-init (seed 1234)",
                       "mode": args.mode, "lanes": (lead["lanes"] if args.mode == "interleaved" else 1), "frames_per_launch": F,
                       "frames_per_launch_note": ("one step = one batch-%d item (its own gravity, plane block and draws, main.py:261-298); %d consecutive items of "
                                                  "the stream share every launch of a tick (program recorded for batch %d), an item's result does not depend on "
                                                  "its partner" % (B, F, F * B)) if F > 1 else None,
                       "frames_in_flight": (2 * lead["lanes"] * F if args.mode == "interleaved" else args.in_flight if args.mode == "streams" else 1),
                       "latency_note": ("`value` is a stream rate: %d batch-%d items share every launch on %d lanes (%d frames in flight; first depth map of a stream "
                                        "after first_item_latency_ms).  The reference's operator called one frame at a time is `sequential_call_cnn`; the "
                                        "one- and two-items-per-launch stream rates are in `extra_legs`." % (F, B, lead["lanes"], 2 * lead["lanes"] * F)) if F > 1 else None,
                       "sharding": "frames round-robin over %d rank(s), no data-path collective" % world},
            "rmse_vs_oracle": (round(job["rmse"], 8) if job["rmse"] is not None else None),
            "roofline": lead["roofline"], "cpu_baseline": cpu_baseline,
        }
        line.update(lead["extra"])
        if main_mode == "fp32" and (lead["steady"] or {}).get("shader_clock_ghz"):
            # The nominal fp32 MFMA peak assumes 2.4 GHz; under the stream's mix of MFMA, LDS and HBM traffic the chip clocks lower (stamps behind
            # every item of the steady-state stream, include/vidc.h vidc_clock_stamp).  `roofline.frac` stays against the nominal peak; this is
            # the executed conv FLOP/s against the peak any kernel could reach at the clock the chip actually held.
            ghz = lead["steady"]["shader_clock_ghz"]
            peak_s = PEAK_F32_MFMA_TFLOPS * ghz / 2.4
            ex = lead["extra"]["conv_stack"]["at_measured_frame_rate"]["tflops_executed"]
            line["sustained_clock"] = {
                "shader_clock_ghz": ghz, "nominal_ghz": 2.4, "fp32_mfma_peak_at_that_clock_tflops": round(peak_s, 1),
                "conv_stack_frac_of_that_peak": round(ex / peak_s, 4),
                "conv_stack_frac_of_that_peak_steady_state": round(lead["extra"]["conv_stack"]["executed_gflop_per_frame"] * lead["steady"]["frames_per_s"] / 1e3 / peak_s, 4),
                "note": "average shader clock between the two items that bound the steady-state window (an idle XCD's counter hardly advances, so gaps pull it down): difference of the shader-cycle counter over the "
                        "difference of the 100 MHz wall clock, written per XCD by a stamp kernel behind each item's result (median over the XCDs)"}
        for m in legs[1:]:
            jm, rm = jobs[m], res[m]
            line.update({
                "value_" + m: round(sorted(region_fps[m])[n_reg // 2][0], 3), "ms_per_step_" + m: round(1e3 * sorted(region_fps[m])[n_reg // 2][1] / args.steps, 4),
                "regions_" + m: [round(v[0], 3) for v in region_fps[m]], "steady_state_frames_per_s_" + m: (rm["steady"] or {}).get("frames_per_s"),
                "dtype_" + m: ("f32" if m == "fp32" else "f32+bf16x3"), "rmse_vs_oracle_" + m: (round(jm["rmse"], 8) if jm["rmse"] is not None else None),
                "roofline_" + m: rm["roofline"],
                m + "_leg": {"what": ("the same %d steps in the mixed mode: the compute-bound convs on 3 x v_mfma_f32_32x32x16_bf16 per fp32-equivalent product "
                                      "(operands split hi + lo in bf16, fp32 accumulate: narrower products than the reference's fp32), fresh pipeline in this "
                                      "process" % args.steps) if m == "mixed" else
                                     ("the same %d steps with every conv on v_mfma_f32_32x32x2_f32 (exact fp32 products and sums), fresh pipeline in this process" % args.steps),
                             "lanes": rm["lanes"], "program_ms": rm["extra"].get("program_ms"), "conv_ms_per_frame": rm["extra"].get("conv_ms_per_frame"),
                             "first_item_latency_ms": rm["extra"].get("first_item_latency_ms"),
                             "sequential_call_cnn": rm["extra"].get("sequential_call_cnn"),
                             "conv_stack": rm["extra"].get("conv_stack")}})
        if extra_legs is not None:
            line["extra_legs"] = extra_legs
        print(json.dumps(line), flush=True)
    if world > 1 or os.environ.get("VIDC_DIST_WORLD1", "0") == "1":
        dist.barrier()
        dist.destroy_process_group()


def run_extra_legs(args):
    """Driver-visible numbers for BASELINE configs[4] and configs[2] and for the stream mode's two-items-per-launch setting: each one is THIS script started as a child process
    (its own GPU context, its own environment; a crash or a hang there cannot take the headline down) after both main legs have been
    measured and before the line is printed.  Skipped once the wall clock of this run passes --extra-legs-budget."""
    import subprocess
    t_start = _T0
    out = {}
    specs = [("configs[4] training step, bf16, batch 8 per GPU, 320x240", {"VIDC_TRAIN_PRECISION": "bf16"},
              ["--train", "--batch", "8", "--steps", "5", "--warmup", "3"]),
             ("configs[2] 640x480 stream, batch 8, Mask R-CNN plane head every frame", {},
              ["--batch", "8", "--source", "640x480", "--height", "240", "--plane-head", "--steps", "20", "--warmup", "4", "--frames-per-launch", "1",
               "--no-cpu-baseline", "--no-sequential-leg", "--no-extra-legs"]),
             ("configs[3] 1280x720 stream, one GPU's share of batch 32 = 4 frames per item (plane mask fixed)", {},
              ["--batch", "4", "--source", "1280x720", "--height", "240", "--steps", "20", "--warmup", "4", "--frames-per-launch", "1",
               "--no-cpu-baseline", "--no-sequential-leg", "--no-extra-legs"]),
             # the headline runs four items per launch (first depth map of a stream after ~26 ms); this is the same workload at the
             # lower-latency setting of the knob (two items per launch, ~15 ms), in the headline's arithmetic
             ("configs[1] with two items per launch (lower first-item latency), fp32", {},
              ["--steps", "20", "--warmup", "5", "--frames-per-launch", "2", "--no-mixed-leg", "--no-cpu-baseline", "--no-sequential-leg", "--no-extra-legs"]),
             ("configs[1] with eight items per launch (the throughput end of the knob: 48 frames in flight), fp32", {},
              ["--steps", "40", "--warmup", "8", "--frames-per-launch", "8", "--no-mixed-leg", "--no-cpu-baseline", "--no-sequential-leg", "--no-extra-legs"]),
             ("configs[1] with one item per launch (every launch is one batch-1 frame), fp32", {},
              ["--steps", "20", "--warmup", "5", "--frames-per-launch", "1", "--no-mixed-leg", "--no-cpu-baseline", "--no-sequential-leg", "--no-extra-legs"])]
    for name, env_add, flags in specs:
        if time.perf_counter() - t_start > args.extra_legs_budget:
            out[name] = {"skipped": "wall-clock budget of the run (%.0f s) used up" % args.extra_legs_budget}
            continue
        # the child is a fresh one-rank job: nothing of this process's launcher / process-group environment may reach it (a parent run with
        # VIDC_DIST_WORLD1=1 still holds its TCPStore on MASTER_PORT: a child that inherited it would fail with address-in-use)
        drop = ("VIDC_PRECISION", "RANK", "WORLD_SIZE", "LOCAL_RANK", "GROUP_RANK", "LOCAL_WORLD_SIZE", "ROLE_RANK", "ROLE_WORLD_SIZE", "VIDC_DIST_WORLD1",
                "MASTER_PORT", "MASTER_ADDR")
        env = {k: v for k, v in os.environ.items() if k not in drop and not k.startswith("TORCHELASTIC_")}
        env.update(env_add)
        t1 = time.perf_counter()
        remaining = args.extra_legs_budget - (t1 - t_start)
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--gpus", "1"] + flags, env=env, capture_output=True, text=True,
                               timeout=max(20.0, min(80.0, remaining + 30.0)))      # a leg started inside the budget gets at most 30 s past it
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not lines:
                out[name] = {"error": "exit code %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
                continue
            d = json.loads(lines[-1])
            out[name] = {k: d.get(k) for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "dtype", "rmse_vs_oracle", "value_mixed", "ms_per_step_mixed",
                                                       "dtype_mixed", "losses", "first_item_latency_ms", "regions", "steady_state_frames_per_s", "config")
                         if d.get(k) is not None}
            out[name]["roofline"] = {k: (d.get("roofline") or {}).get(k) for k in ("bound", "achieved", "peak", "unit", "frac")}
            out[name]["command"] = "bench.py --gpus 1 " + " ".join(flags) + ("  [" + " ".join("%s=%s" % kv for kv in env_add.items()) + "]" if env_add else "")
            out[name]["child_wall_s"] = round(time.perf_counter() - t1, 1)
        except Exception as e:      # noqa: BLE001  (timeout, unparsable output: recorded, never raised)
            out[name] = {"error": "%s: %s" % (type(e).__name__, str(e)[-300:])}
    return out


if __name__ == "__main__":
    main()
