"""TEST/BENCH INFRASTRUCTURE ONLY -- torch-CPU port of the reference hot path, used as `cpu_baseline` by bench.py
(kind "port") and cross-checked against the numpy oracle in tests.

It issues the same ATen op sequence on the CPU device as the reference does (models/TPNet.py:67-99, 112-128:
index, mul, exp, scatter_add_, stack, cat, matmul, masked assignment, log), including the eager dense decay,
because that is what the reference's CPU path costs.  The reference itself cannot travel to the GPU box.
Never imported by tpnet_amd/.
"""
import numpy as np
import torch


class TorchPort:
    def __init__(self, P0: np.ndarray, num_layer: int, lam: float, t0: float):
        self.L = num_layer
        self.lam = lam
        self.P = [torch.from_numpy(np.array(P0, dtype=np.float32))]
        for _ in range(num_layer):
            self.P.append(torch.zeros_like(self.P[0]))
        self.now = np.float64(t0)
        self.d = self.P[0].shape[1]

    def update(self, src: np.ndarray, dst: np.ndarray, t: np.ndarray):          # TPNet.py:67-99
        s = torch.from_numpy(src)
        d = torch.from_numpy(dst)
        nxt = t[-1]
        tf = torch.from_numpy(t).to(torch.float32)
        w = torch.exp(-self.lam * (nxt - tf))[:, None]
        g = np.exp(-self.lam * (nxt - self.now))
        for i in range(1, self.L + 1):
            self.P[i] = self.P[i] * np.power(g, i)
        for i in range(self.L, 0, -1):
            ms = self.P[i - 1][d] * w
            md = self.P[i - 1][s] * w
            self.P[i].scatter_add_(0, s[:, None].expand(-1, self.d), ms)
            self.P[i].scatter_add_(0, d[:, None].expand(-1, self.d), md)
        self.now = np.float64(nxt)

    def pair_gram(self, u: np.ndarray, v: np.ndarray, not_scale: bool = False):  # TPNet.py:112-128
        a = torch.stack([self.P[i][u] for i in range(self.L + 1)], dim=1)
        b = torch.stack([self.P[i][v] for i in range(self.L + 1)], dim=1)
        r = torch.cat([a, b], dim=1)
        f = torch.matmul(r, r.transpose(1, 2)).reshape(len(u), -1)
        if not not_scale:
            f[f < 0] = 0
            f = torch.log(f + 1.0)
        return f
