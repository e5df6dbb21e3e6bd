import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from vi_depth_completion_amd import ops
stamps = ops.clock_stamps(8)
x = torch.randn(4096, 4096, device="cuda")
for i in range(8):
    ops.clock_stamp(stamps, i)
    for _ in range(10):
        x = (x @ x) * 1e-3
torch.cuda.synchronize()
print(stamps[:2].cpu())
for i in range(7):
    print(i, ops.shader_clock_ghz(stamps, i, i + 1))
print("0..7", ops.shader_clock_ghz(stamps, 0, 7))
