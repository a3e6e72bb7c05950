#!/usr/bin/env python3
"""Developer tool: PINT walk-matrix state (f-4) at Wikipedia size -- tpnet_amd.MatrixMemory (TPNet engine) against the
reference's own op sequence run with stock torch ops on the same GPU (models/MemoryModel.py:387-405)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpnet_amd.matrix_memory import MatrixMemory
from tpnet_amd.stream import CONFIGS, synthetic_stream
c = CONFIGS["C1"]; B = 200; nb = 60; H = 3
src, dst, t, N = synthetic_stream(c["U"], c["I"], nb * B, c["span"] * nb * B / c["E"], 0)
dev = torch.device("cuda:0")
rng = np.random.RandomState(0)
init = torch.zeros((N, N, H + 1), device=dev)
init[:, :, 0] = torch.eye(N, device=dev)
init[:, :, 1:] = (torch.rand((N, N, H), device=dev) < 0.01).float()          # non-trivial hops (from reset nothing moves)
print(f"N={N} H={H}: matrix {N * N * (H + 1) * 4 / 1e9:.2f} GB, batch {B}", flush=True)

mm = MatrixMemory(num_node=N, num_hop=H, device="cuda:0").to(dev)
mm.reload_memory(init)
for rep in range(2):
    mm.reload_memory(init); mm.update(src[:B], dst[:B]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(1, nb):
        mm.update(src[b * B:(b + 1) * B], dst[b * B:(b + 1) * B])
        mem = mm.get_memory(src[b * B:(b + 1) * B], dst[b * B:(b + 1) * B])
    torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f"tpnet_amd.MatrixMemory: {el / (nb - 1) * 1e6:.0f} us per batch (update + get_memory), {(nb - 1) * B / el / 1e6:.2f} M edges/s")
got = mm.matrix.detach().clone()

matrix = init.clone(); P = mm.P.data.clone()
def ref_update(s, d):
    ids = torch.from_numpy(np.concatenate([s, d])).to(dev)
    msg = torch.matmul(matrix[np.concatenate([d, s])], P[None, :, :])
    matrix.scatter_add_(dim=0, index=ids[:, None, None].expand(msg.shape), src=msg)
def ref_get(s, d):
    m = matrix[s, d]
    return m / (torch.sum(m, dim=1, keepdim=True) + 1e-4)
for rep in range(2):
    matrix.copy_(init); ref_update(src[:B], dst[:B]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for b in range(1, nb):
        ref_update(src[b * B:(b + 1) * B], dst[b * B:(b + 1) * B])
        rmem = ref_get(src[b * B:(b + 1) * B], dst[b * B:(b + 1) * B])
    torch.cuda.synchronize(); el2 = time.perf_counter() - t0
print(f"stock torch ops on the GPU: {el2 / (nb - 1) * 1e6:.0f} us per batch -> {el2 / el:.1f}x")
err = float((got - matrix).abs().max() / matrix.abs().max())
print(f"max |diff| / max |ref| of the final matrices: {err:.2e}; last get_memory max diff {float((mem - rmem).abs().max()):.2e}")
