#!/usr/bin/env python3
"""Developer tool: self.mlp in the fp32 class -- tpnet_mlp64_f32 (split-bf16 operands on the bf16 matrix cores by default; the
fp32-MFMA variant with TPNET_DEV_MLP_F32_MODE=1 on the dev library) against the torch layers: time and error."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpnet_amd import fused_feature as ff
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
    for n in (2000, 8192, 20000, 80000, 80007, 800000):
        x = torch.rand(n, 64, device="cuda") * 12           # log(1 + G) features: 0 .. ~12
        want = mlp(x.double().cpu().to(torch.float64)) if False else None
        ref = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64)).double().cuda()
        ref.load_state_dict({k: v.double() for k, v in mlp.state_dict().items()})
        exact = ref(x.double())
        y = ff.mlp_f32(mlp, x)
        yt = mlp(x)
        scale = float(exact.abs().max())
        tt = timeit(lambda: mlp(x)); tf = timeit(lambda: ff.mlp_f32(mlp, x))
        print(f"n={n}: torch fp32 {tt:.1f} us (err {float((yt.double() - exact).abs().max()) / scale:.2e} of scale), "
              f"tpnet_mlp64_f32 {tf:.1f} us (err {float((y.double() - exact).abs().max()) / scale:.2e}) -> {tt / tf:.2f}x", flush=True)
