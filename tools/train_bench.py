#!/usr/bin/env python3
"""Training throughput of the depth-completion network (BASELINE configs[4]: "Training loop: depth_completion.py L1 + normal loss,
batch=64 on 8xMI355X"): one `_run_training_iteration` per step (network_run.py:231-254) on this rank's share of the batch.

    python tools/train_bench.py --batch 8 --steps 5                                  # one GPU's share of batch 64
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 ... tools/train_bench.py --batch 8

Frames shard over ranks (weak scaling), BatchNorm statistics per rank (like the reference's DataParallel replicas), gradients
summed with a bucketed RCCL all-reduce of the flat 1.24 GB gradient buffer.  Prints one JSON line (rank 0)."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vi_depth_completion_amd import synthetic as S  # noqa: E402
from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN  # noqa: E402
from vi_depth_completion_amd.training import DepthCompletionTrainer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--same-data", action="store_true", help="every rank trains on rank 0's frames: the summed gradient is N x the single-rank one and\n"
                                                             "Adam's update does not depend on the gradient's scale, so the losses must follow the single-rank run")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.set_grad_enabled(False)
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("VIDC_DIST_BACKEND", "nccl")      # "nccl" is RCCL on ROCm; "gloo" to try the N > 1 path on a 1-GPU box
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))
    cnn = ModifiedFPN().to(dev)
    cnn.load_state_dict(S.seeded_state_dict(cnn.state_dict(), 1234, device=dev))
    cnn.train()
    tr = DepthCompletionTrainer(cnn, 1e-4)
    B = args.batch
    b = S.synthetic_batch(B, 240, 320, 1234, frame0=0 if args.same_data else rank * B)
    image = b["image"].to(dev)
    normal = torch.nn.functional.normalize(image - 0.5, dim=1)
    depth_in = b["sparse_depth"].to(dev)
    gt = S.synthetic_ground_truth_depth(b["image"], 1234).to(dev)
    losses = []
    for _ in range(args.warmup):
        losses.append(tr.step(image, normal, depth_in, gt))
    warm = [round(float(x), 6) for x in losses]
    losses = []
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        losses.append(tr.step(image, normal, depth_in, gt))
    t_enq = time.perf_counter() - t0          # host time to enqueue the steps (the GPU runs behind it when the step is GPU-bound)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": "training frames/sec", "value": round(world * B * args.steps / dt, 2), "unit": "frames/s", "n_gpus": world,
                          "batch_per_gpu": B, "ms_per_step": round(1e3 * dt / args.steps, 1), "host_enqueue_ms_per_step": round(1e3 * t_enq / args.steps, 1), "dtype": {"fp32": "f32 (fp32 MFMA fwd / dgrad / wgrad)", "bf16x3": "f32+bf16x3 (split-bf16 3-pass MFMA fwd / dgrad / wgrad)",
                                    "bf16": "bf16 operands, fp32 accumulate (MFMA fwd / dgrad / wgrad); fp32 master weights, BatchNorm, loss, Adam"}[os.environ.get("VIDC_TRAIN_PRECISION", "fp32")],
                          "losses": warm + [round(float(x), 6) for x in losses],
                          "config": "BASELINE configs[4]: ModifiedFPN training step (train-mode BN, masked L1 / (H*W), Adam), 320x240, synthetic"}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
