"""Worker of tests/test_training.py::test_two_ranks_through_training_steps: started twice by `python -m torch.distributed.run`
(gloo; both ranks share the one GPU of the box), NOT collected by pytest.

Replaces network_run.py:97-99 (`DataParallel`: replicas normalise BatchNorm over THEIR frames, gradients are summed).  Every rank trains
`ModifiedFPN` on its own shard through `DepthCompletionTrainer.step` -- steps 1-2 eager (the backward cut after the decoder, the
decoder's gradients all-reduced while the pyramids' backward is queued, then the rest), steps 3-4 as the two captured graphs -- and,
beside it, keeps two single-process trainers, one per shard, whose gradients it adds up itself:

    g_sum = g(shard 0) + g(shard 1)                  fp32, two terms: the order cannot matter
    both replicas step Adam with g_sum

The distributed trainer's flat gradient must equal g_sum bit for bit after every step, its loss the shard's loss, and its module's
parameters AND running statistics those of the replica that saw the same shard.  With VIDC_TRAIN_GRAD_BF16=1 the buckets travel in
bf16: the comparison is then against round_bf16(g(shard 0)) + round_bf16(g(shard 1)), rounded to bf16 again (what a bf16 SUM yields).
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert world == 2
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("gloo")
    from vi_depth_completion_amd import synthetic as S
    from vi_depth_completion_amd.networks.depth_completion import ModifiedFPN
    from vi_depth_completion_amd.training import DepthCompletionTrainer
    bf16_buckets = os.environ.get("VIDC_TRAIN_GRAD_BF16", "0") == "1"
    B, H, W = 2, 64, 96

    def shard(r):
        g = torch.Generator().manual_seed(100 + r)
        img = torch.rand(B, 3, H, W, generator=g)
        nrm = torch.nn.functional.normalize(torch.randn(B, 3, H, W, generator=g), dim=1)
        dep = torch.rand(B, 1, H, W, generator=g) * (torch.rand(B, 1, H, W, generator=g) < 0.02)
        gt = torch.rand(B, 1, H, W, generator=g) * 4 + 0.5
        return [t.to(dev) for t in (img, nrm, dep, gt)]

    def model():
        m = ModifiedFPN().to(dev)
        m.load_state_dict(S.seeded_state_dict(m.state_dict(), 7, device=dev))
        m.train()
        return m

    shards = [shard(0), shard(1)]
    m_dist = model()
    t_dist = DepthCompletionTrainer(m_dist, 1e-4)
    assert t_dist._distributed() and (t_dist.buckets.compress == "bf16") == bf16_buckets
    replicas = []
    for r in range(2):
        t = DepthCompletionTrainer(model(), 1e-4)
        t._distributed = lambda: False
        t.buckets.all_reduce_async = lambda *a, **k: []
        replicas.append(t)

    def rbf(x):
        return x.to(torch.bfloat16).to(torch.float32)

    for it in range(4):
        loss_d = t_dist.step(*shards[rank])
        losses = [t.forward_backward(*shards[r])[0] for r, t in enumerate(replicas)]
        g0, g1 = replicas[0].flat_g, replicas[1].flat_g
        if bf16_buckets:
            want = rbf(rbf(g0) + rbf(g1))
            off = t_dist._dec_off
            if off % 8:                       # the few elements in front of the 16-byte grid of the decoder's range travel in fp32
                up = off + 8 - off % 8
                want[off:up] = (g0 + g1)[off:up]
        else:
            want = g0 + g1
        assert float(loss_d) == float(losses[rank]), (it, float(loss_d), float(losses[rank]))
        same = torch.equal(t_dist.flat_g, want)
        if not same:
            d = (t_dist.flat_g - want).abs()
            raise AssertionError("step %d: all-reduced gradient differs from the sum of the shards' gradients: %d elements, max %.3e (scale %.3e)"
                                 % (it, int((d > 0).sum()), float(d.max()), float(want.abs().max())))
        for t in replicas:
            t.flat_g.copy_(want)
            t.optimizer_step(reduced=True)
        mine = replicas[rank].cnn.state_dict()
        for k, v in m_dist.state_dict().items():
            assert torch.equal(v, mine[k]), "step %d: %s differs from the single-process replica of shard %d" % (it, k, rank)
        # both ranks hold the same parameters (the replicas' running statistics differ: per-rank BatchNorm, as in DataParallel)
        chk = t_dist.flat_p.double().sum().cpu()
        both = [torch.zeros_like(chk) for _ in range(2)]
        dist.all_gather(both, chk)
        assert float(both[0]) == float(both[1]), (it, both)
    assert len(t_dist._graphs) == 1, "steps 3-4 must have run as captured graphs"
    a = replicas[0].cnn.state_dict()["resnet_rgb.bn1.running_mean"]
    b = replicas[1].cnn.state_dict()["resnet_rgb.bn1.running_mean"]
    assert not torch.equal(a, b), "the two shards must differ"
    torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("TWO_RANK_TRAINING_OK bf16_buckets=%d losses=%s" % (int(bf16_buckets), [round(float(x), 6) for x in losses]))


if __name__ == "__main__":
    main()
