#!/usr/bin/env python3
"""Developer tool: LinkPredictor_v1 forward (172 + 172 + 64 -> 172 -> 1, the reference's Wikipedia sizes) fused on the bf16
matrix cores (tpnet_decoder_bf16) vs the torch fp32 layers, HIP-event timing."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpnet_amd.callers import LinkPredictor_v1
dev = torch.device("cuda:0")


class Feat(torch.nn.Module):
    pair_wise_feature_dim = 64
    def __init__(self, n):
        super().__init__(); self.f = torch.rand(n, 64, device=dev)
    def get_pair_wise_feature(self, src_node_ids, dst_node_ids):
        return self.f[: len(src_node_ids)]


for n in (1000, 2000, 20000, 200000):
    dec = LinkPredictor_v1(172, 172, 172, 1, Feat(n), False).to(dev)
    src, dst = torch.randn(n, 172, device=dev), torch.randn(n, 172, device=dev)
    ids = np.arange(n)
    res = {}
    with torch.no_grad():
        for fused in (False, True):
            dec.fused = fused
            for _ in range(5):
                out = dec(ids, ids, src, dst)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                out = dec(ids, ids, src, dst)
            e1.record(); torch.cuda.synchronize()
            res[fused] = e0.elapsed_time(e1) / 50 * 1e3
    fl = 2 * n * (408 * 172 + 172)
    print(f"n={n}: torch fp32 {res[False]:.1f} us, fused bf16 {res[True]:.1f} us -> {res[False] / res[True]:.1f}x "
          f"({fl / res[True] / 1e6:.1f} TFLOP/s)")
