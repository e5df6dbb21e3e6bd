import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from vi_depth_completion_amd import synthetic as S
torch.set_grad_enabled(False)
dev = torch.device("cuda")
H, W = 256, 320
r = bench.build_pipeline(H, W, dev)
pipe = r[0] if isinstance(r, tuple) else r
pool = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in S.synthetic_batch(1, H, W, 1234, frame0=i).items()} for i in range(4)]
def frames(n):
    for i in range(n):
        yield pool[i % 4]
for es in (200, 0):
    pipe.args.enriched_samples = es
    for lanes in (1, 2):
        for out in pipe.run_interleaved(frames(30), copy_outputs=False, lanes=lanes): pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for out in pipe.run_interleaved(frames(400), copy_outputs=False, lanes=lanes): pass
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("enriched_samples=%d lanes=%d: %.1f fps (%.3f ms)" % (es, lanes, 400 / dt, 1e3 * dt / 400), flush=True)
