#!/usr/bin/env python3
"""Developer tool: self.mlp (64->256->64) fused bf16 MFMA kernel vs the torch fp32 layers, forward, HIP-event timing."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpnet_amd.fused_mlp import fused_mlp
torch.manual_seed(0)
mlp = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64)).cuda()
def timeit(fn, reps=30):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
with torch.no_grad():
    for n in (2000, 20000, 160000, 800000):
        x = torch.rand(n, 64, device="cuda") * 10
        tt = timeit(lambda: mlp(x)); tf = timeit(lambda: fused_mlp(mlp, x))
        fl = n * 2 * (64 * 256 + 256 * 64)
        print(f"n={n}: torch fp32 {tt:.1f} us, fused bf16 MFMA {tf:.1f} us ({fl / tf / 1e6:.1f} TFLOP/s incl. weight prep) -> {tt / tf:.2f}x")
# backward (weight gradients): tpnet_mlp64_bwd_bf16 + the sum of the workgroups' partials vs the fp32 torch expressions
import tpnet_amd.fused_mlp as fm
prep = fm._prepared(mlp)
for n in (2000, 20000, 200000):
    x = torch.rand(n, 64, device="cuda") * 10; gy = torch.randn(n, 64, device="cuda")
    fm.BACKWARD = "torch"; tt = timeit(lambda: fm.weight_grads(x, gy, mlp[0].weight, mlp[0].bias, mlp[2].weight, prep))
    fm.BACKWARD = "mfma"; tf = timeit(lambda: fm.weight_grads(x, gy, mlp[0].weight, mlp[0].bias, mlp[2].weight, prep))
    print(f"backward n={n}: torch fp32 {tt:.1f} us, bf16 MFMA kernel + partial sum {tf:.1f} us -> {tt / tf:.2f}x")
