"""`bench.py --gpus N` without a launcher spawns its N ranks itself (replaces network_run.py:97-99's DataParallel
as the way to use N GPUs).  CPU: the launcher, the gloo rendezvous, the gather and the rank-count checks, with `--launcher-selftest`
(no GPU work).  GPU (1-GPU box): the real bench under two gloo ranks sharing the GPU, and every collective of the package on a
world-size-1 `nccl` (= RCCL) group."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(kw)
    return env


def _last_json(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert lines, stdout[-2000:]
    return json.loads(lines[-1])


def test_bench_spawns_its_own_ranks_and_counts_them():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest", "--steps", "7"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["backend"] == "gloo"
    assert line["frames"] == 14.0 and line["seconds"] == 1.5          # sum over ranks / max over ranks
    assert sum(1 for ln in r.stdout.splitlines() if ln.startswith("{")) == 1      # ONE line, from rank 0


def test_bench_eight_ranks_spawn_gather_and_count():
    """The width the driver's scaling run uses (--gpus 8), once, on the CPU: eight ranks spawned by the launcher, gloo rendezvous on
    127.0.0.1, one 4-double gather, `n_gpus == 8`, ONE line; and the per-rank log files the launcher keeps."""
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--launcher-selftest", "--steps", "3"], env=_env(VIDC_RANK_LOG_DIR=tmp, OMP_NUM_THREADS="1"),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        line = _last_json(r.stdout)
        assert line["n_gpus"] == 8 and line["backend"] == "gloo"
        assert line["frames"] == 24.0 and line["seconds"] == 1.0 + 0.5 * 7      # sum over ranks / max over ranks
        assert sum(1 for ln in r.stdout.splitlines() if ln.startswith("{")) == 1
        assert any(files for _root, _dirs, files in os.walk(tmp)), "the launcher keeps a log file per rank"


def test_launcher_replays_the_rank_logs_when_a_rank_dies():
    """A rank that exits non-zero takes the job down with a non-zero code, and its last words come back on the launcher's stderr."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launcher-selftest", "--steps", "1"], env=_env(VIDC_SELFTEST_KILL_RANK="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "selftest: rank 1 told to die" in r.stderr and "per-rank logs under" in r.stderr


def test_bench_single_rank_needs_no_group():
    r = subprocess.run([sys.executable, BENCH, "--launcher-selftest", "--steps", "5"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _last_json(r.stdout)["n_gpus"] == 1


def test_bench_refuses_more_ranks_than_gpus():
    """--gpus 2 on a node with fewer than 2 GPUs: every rank refuses (the launching process itself makes no HIP call, so it does not
    count devices) and the job exits non-zero, unless the backend is gloo."""
    if torch.cuda.device_count() >= 2:
        pytest.skip("needs a node with fewer than 2 GPUs")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1"], env=_env(VIDC_DIST_BACKEND="nccl"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and ("one rank per GPU" in r.stderr or "no GPU visible" in r.stderr), r.stderr[-2000:]


def test_bench_refuses_a_world_that_is_not_gpus():
    """A launcher that starts another number of ranks than --gpus says is an error, not a line with the wrong n_gpus."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--launcher-selftest"], env=_env(RANK="0", WORLD_SIZE="1"), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "--gpus 4 but the launcher started 1" in (r.stderr + r.stdout)


@pytest.mark.gpu
def test_bench_two_gloo_ranks_on_one_gpu():
    """The whole bench (timed region between barriers, parity check, gather) with two ranks sharing the one GPU over gloo."""
    cmd = [sys.executable, BENCH, "--gpus", "2", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--no-fp32-leg", "--no-sequential-leg"]
    # No second try (round 4 retried once around "a rank gone within seconds of starting" seen one time on a fresh box): the launcher now
    # keeps every rank's output in a file of its own and replays it on failure, so a death shows its cause here -- and fails the test.
    log_dir = os.path.join(ROOT, "gpurun_out", "rank_logs_two_gloo_ranks")
    r = subprocess.run(cmd, env=_env(VIDC_DIST_BACKEND="gloo", VIDC_RANK_LOG_DIR=log_dir), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, "two-rank job failed on its FIRST attempt (per-rank logs: %s):\n%s" % (log_dir, (r.stdout + r.stderr)[-8000:])
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "weak" and line["value"] > 0
    assert line["steps"] == 4 and abs(line["value"] * line["ms_per_step"] * 1e-3 - 2.0) < 1e-2      # 2 ranks x 4 frames / max time


@pytest.mark.gpu
def test_bench_eight_gloo_ranks_with_real_pipelines_on_one_gpu():
    """Width-8 rehearsal: the command the driver's scaling run issues, `bench.py --gpus 8`, with eight REAL pipelines -- weights,
    frame programs, three lanes each -- sharing the one GPU of the box over gloo (8 x ~6 GB).  No scaling claim follows from it; it takes "never ran
    at width 8 with a pipeline" off the list of things the 8-GPU run can trip over: spawn before any HIP call, rank-staggered GPU start, barriers
    around the timed regions of eight ranks, the gather, `n_gpus == 8`, one line, eight per-rank logs."""
    log_dir = os.path.join(ROOT, "gpurun_out", "rank_logs_eight_gloo_ranks")
    cmd = [sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--regions", "2", "--no-cpu-baseline", "--no-extra-legs", "--no-mixed-leg",
           "--no-sequential-leg"]
    r = subprocess.run(cmd, env=_env(VIDC_DIST_BACKEND="gloo", VIDC_RANK_LOG_DIR=log_dir, OMP_NUM_THREADS="2"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, "eight-rank job failed (per-rank logs: %s):\n%s" % (log_dir, (r.stdout + r.stderr)[-8000:])
    line = _last_json(r.stdout)
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["value"] > 0 and line["steps"] == 2
    assert sum(1 for ln in r.stdout.splitlines() if ln.startswith("{")) == 1
    logs = [f for _root, _dirs, files in os.walk(log_dir) for f in files]
    assert len([f for f in logs if "stderr" in f or f.endswith(".log") or f.endswith(".err")]) >= 8 or len(logs) >= 8, logs


def _nccl_world1(port):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0",
                      VIDC_DIST_WORLD1="1")       # a world of one still goes through RCCL (sharding.collectives_active)
    import torch.distributed as dist
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    return dist, dev


@pytest.mark.gpu
def test_rccl_world_size_one_runs_every_collective_of_the_package():
    """RCCL itself executes (a world of one GPU): the metric gather on a device record, the evaluation all-reduce, and the training
    step's bucketed gradient all-reduce on device buffers incl. the overlapped (two-bucket) path -- same code the 8-GPU job runs."""
    code = r'''
import os, sys, socket
sys.path.insert(0, %r)
import numpy as np, torch
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
sys.path.insert(0, os.path.join(%r, "tests"))
from test_bench_launcher import _nccl_world1
dist, dev = _nccl_world1(port)
from vi_depth_completion_amd import sharding, evaluation, synthetic as S
rec = sharding.metric_record(5, 2.0, 1.0, 4.0, device=dev)
out = [torch.zeros_like(rec)]
dist.all_gather(out, rec)                      # what gather_records does for world > 1, on the device record
assert torch.equal(out[0].cpu(), rec.cpu())
assert sharding.combine(sharding.gather_records(rec))["frames_per_s"] == 2.5
tot = torch.arange(8, dtype=torch.float64, device=dev)
evaluation.all_reduce_totals(tot)
assert torch.equal(tot.cpu(), torch.arange(8, dtype=torch.float64))
from vi_depth_completion_amd.training import DepthCompletionTrainer
from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
torch.manual_seed(0)
m = ModifiedFPN().to(dev)
m.load_state_dict(S.seeded_state_dict(m.state_dict(), 7, device=dev))
ref = ModifiedFPN().to(dev)
ref.load_state_dict(S.seeded_state_dict(ref.state_dict(), 7, device=dev))
B, H, W = 2, 64, 96
img = torch.rand(B, 3, H, W, device=dev); nrm = torch.nn.functional.normalize(torch.randn(B, 3, H, W, device=dev), dim=1)
dep = torch.rand(B, 1, H, W, device=dev) * (torch.rand(B, 1, H, W, device=dev) < 0.02)
gt = torch.rand(B, 1, H, W, device=dev) * 4 + 0.5
m.train(); ref.train()
t_dist = DepthCompletionTrainer(m, 1e-4)
assert t_dist._distributed()                   # world of one, collectives forced on
t_solo = DepthCompletionTrainer(ref, 1e-4)
t_solo._distributed = lambda: False            # the same steps with no collective at all
t_solo.buckets.all_reduce_async = lambda *a, **k: []
for it in range(4):                            # steps 1-2 eager (cut backward + overlapped all-reduce), 3-4 the two captured graphs
    l1 = t_dist.step(img, nrm, dep, gt)
    l0 = t_solo.step(img, nrm, dep, gt)
    assert float(l1) == float(l0), (it, float(l1), float(l0))
for (k, a), (_, b) in zip(m.state_dict().items(), ref.state_dict().items()):
    assert torch.equal(a, b), k               # SUM over a world of one = identity: the RCCL path must not change a bit
torch.cuda.synchronize()
dist.barrier()
dist.destroy_process_group()
print("RCCL_WORLD1_OK")
''' % (ROOT, ROOT)
    r = subprocess.run([sys.executable, "-c", code], env=_env(HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout + r.stderr)[-4000:]


def test_extra_legs_can_never_take_the_headline_down(monkeypatch):
    """`bench.run_extra_legs` (the driver-visible configs[4] / configs[2] numbers appended to the default N = 1 line) runs each leg as a
    child process and records whatever happens -- a crash, a hang past the timeout, unparsable output, an exhausted wall-clock budget --
    as data; nothing is raised into the process that is about to print the headline.  No GPU: the child is faked."""
    import argparse
    import importlib
    import time
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    calls = []

    def fake_run(cmd, **kw):
        calls.append((cmd, kw))
        k = len(calls)
        if k == 1:
            return subprocess.CompletedProcess(cmd, 0, stdout='noise\n{"metric": "training frames/sec", "value": 290.4, "unit": "frames/s", "steps": 5, "warmup": 3, '
                                               '"ms_per_step": 27.5, "dtype": "bf16", "losses": [1.0, 0.9], "roofline": {"bound": "mfma", "achieved": 216.0, "peak": 2500.0, '
                                               '"unit": "TFLOP/s", "frac": 0.0864}, "config": {"workload": "w"}}\n', stderr="")
        raise subprocess.TimeoutExpired(cmd, kw.get("timeout"))

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(bench, "_T0", time.perf_counter())
    for k, v in (("VIDC_DIST_WORLD1", "1"), ("MASTER_PORT", "29500"), ("MASTER_ADDR", "127.0.0.1"), ("GROUP_RANK", "0"), ("TORCHELASTIC_RUN_ID", "x")):
        monkeypatch.setenv(k, v)                  # a parent that runs inside a launcher / a world-of-one group: none of it may reach the child
    out = bench.run_extra_legs(argparse.Namespace(extra_legs_budget=75.0))
    names = list(out)
    assert len(names) == 6 and "configs[4]" in names[0] and "configs[2]" in names[1] and "configs[3]" in names[2] and "two items per launch" in names[3]
    assert "eight items per launch" in names[4] and "one item per launch" in names[5]
    assert out[names[0]]["value"] == 290.4 and out[names[0]]["dtype"] == "bf16" and out[names[0]]["roofline"]["frac"] == 0.0864
    assert "--train" in out[names[0]]["command"] and "VIDC_TRAIN_PRECISION=bf16" in out[names[0]]["command"]
    assert "TimeoutExpired" in out[names[1]]["error"]
    assert all(kw.get("timeout") and kw.get("capture_output") for _c, kw in calls)
    assert all("--no-extra-legs" in c or "--train" in c for c, _kw in calls), "a child must not start grandchildren"
    for env in (kw["env"] for _c, kw in calls):
        assert "RANK" not in env and "VIDC_PRECISION" not in env
        assert not any(k in env for k in ("VIDC_DIST_WORLD1", "MASTER_PORT", "MASTER_ADDR", "GROUP_RANK")) and not any(k.startswith("TORCHELASTIC_") for k in env)
    assert all(kw["timeout"] <= 80.0 for _c, kw in calls)
    # a failing child and an exhausted budget
    calls.clear()
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: subprocess.CompletedProcess(cmd, 139, stdout="", stderr="Segmentation fault"))
    out = bench.run_extra_legs(argparse.Namespace(extra_legs_budget=75.0))
    assert all("exit code 139" in v["error"] for v in out.values())
    monkeypatch.setattr(bench, "_T0", time.perf_counter() - 1000.0)
    out = bench.run_extra_legs(argparse.Namespace(extra_legs_budget=75.0))
    assert all("skipped" in v for v in out.values())


def test_items_per_launch_default_follows_the_item_size(monkeypatch):
    """`bench.py` without `--frames-per-launch`: four stream items per launch for batch-1 items (the configuration the metric is quoted
    on), one for items that are batches themselves (the measured tile table covers those program batches); an explicit value wins."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    for argv, want in ((["bench.py"], 4), (["bench.py", "--batch", "8"], 1), (["bench.py", "--frames-per-launch", "2"], 2),
                       (["bench.py", "--batch", "4", "--frames-per-launch", "2"], 2)):
        monkeypatch.setattr(sys, "argv", argv)
        assert bench.parse().frames_per_launch == want, argv
