"""CPU, world_size 2 over gloo: the N>1 logic of bench.py (frame shards, metric gather, whole-job combine)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from vi_depth_completion_amd import sharding, synthetic as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round_robin_covers_every_frame_once():
    for world in (1, 2, 4, 8):
        seen = sorted(f for r in range(world) for f in sharding.frames_of_rank(r, world, 37))
        assert seen == list(range(37))
        sizes = [len(sharding.frames_of_rank(r, world, 37)) for r in range(world)]
        assert max(sizes) - min(sizes) <= 1


def test_frames_do_not_depend_on_world_size():
    """Frame f is a function of (seed, f) only, so a frame's inputs are identical whichever rank owns it."""
    a = S.synthetic_batch(1, 24, 32, 1234, frame0=5)
    b = S.synthetic_batch(3, 24, 32, 1234, frame0=4)
    assert torch.equal(a["image"][0], b["image"][1]) and torch.equal(a["sparse_depth"][0], b["sparse_depth"][1])
    assert torch.equal(a["gravity"][0], b["gravity"][1])


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = sharding.frames_of_rank(rank, world, 10)
    # stand-in for the timed loop: rank r "spends" (r+1) seconds on its frames and has some squared error
    rec = sharding.metric_record(len(mine), float(rank + 1), sum_sq_err=0.25 * (rank + 1), n_px=100.0)
    allrec = sharding.gather_records(rec)
    dist.barrier()
    # frame-sharded evaluation: every rank holds the 8 running sums of its own frames; one all_reduce makes the job's totals
    from vi_depth_completion_amd import evaluation
    tot = torch.tensor([100.0 * (rank + 1), 10.0 * (rank + 1), 4.0 * (rank + 1), 50.0, 60.0, 70.0, 80.0, 90.0 + rank], dtype=torch.float64)
    evaluation.all_reduce_totals(tot)
    if rank == 0:
        res = sharding.combine(allrec)
        res["eval"] = evaluation.stats_to_figures(tot.numpy())
        q.put(res)
    dist.destroy_process_group()


def test_two_rank_gloo_gather():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res["frames"] == 10.0 and res["seconds"] == 2.0            # max over ranks
    assert abs(res["frames_per_s"] - 5.0) < 1e-12
    assert abs(res["rmse"] - (0.75 / 200.0) ** 0.5) < 1e-12
    ev = res["eval"]
    assert ev["n"] == 300 and abs(ev["MAD"] - 30.0 / 300.0) < 1e-12 and abs(ev["RMSE"] - (12.0 / 300.0) ** 0.5) < 1e-12
    assert abs(ev["1.05"] - 100.0 * 100.0 / 300.0) < 1e-9 and abs(ev["1.25^3"] - 100.0 * 181.0 / 300.0) < 1e-9


def test_single_process_paths():
    rec = sharding.metric_record(7, 2.0, 1.0, 4.0)
    out = sharding.combine(sharding.gather_records(rec))
    assert out["frames_per_s"] == 3.5 and out["rmse"] == 0.5
