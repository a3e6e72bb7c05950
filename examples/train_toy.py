#!/usr/bin/env python3
"""Toy temporal link prediction with the drop-in pieces, in the reference's batch order
(train_link_prediction.py:246-386): reset at epoch start; per batch: negatives, encoder readouts (neighbour x src/dst
relative encodings), decoder readouts, update, then the optimiser step.  The encoder here is a small stand-in for the
reference's MLP-Mixer (out of scope): mean over the K neighbours of an MLP of their relative encodings.

    python examples/train_toy.py            # needs a GPU; ~10 s
"""
import os, sys, time
import numpy as np, torch
import torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpnet_amd import RandomProjectionModule
from tpnet_amd.callers import LinkPredictor_v1, RandomNegativeSampler, run_epoch
from tpnet_amd.sampler import GpuRecentNeighborSampler
from tpnet_amd.stream import synthetic_stream

dev = "cuda:0"
U, I, E, B, K, D = 600, 200, 24000, 200, 10, 32
src, dst, t, N = synthetic_stream(U, I, E, 2.0e6, seed=0)
torch.manual_seed(0); np.random.seed(0)
rp = RandomProjectionModule(node_num=N, edge_num=E, dim_factor=10, num_layer=3, time_decay_weight=1e-6, device=dev,
                            use_matrix=False, beginning_time=np.float64(t[0]), not_scale=False, enforce_dim=64).to(dev)
rp.fused_mlp = True                                   # self.mlp on the bf16 matrix cores
enc_mlp = nn.Sequential(nn.Linear(2 * 64, D), nn.ReLU(), nn.Linear(D, D)).to(dev)
decoder = LinkPredictor_v1(input_dim1=D, input_dim2=D, hidden_dim=D, output_dim=1, random_projections=rp,
                           not_encode=False).to(dev)
params = list(enc_mlp.parameters()) + list(decoder.parameters())       # decoder.parameters() includes rp.mlp
opt = torch.optim.Adam([p for p in params if p.requires_grad], lr=2e-3)
sampler = GpuRecentNeighborSampler(src, dst, t, device=dev)
negs = RandomNegativeSampler(src, dst)
encoder = lambda feats, node_ids, times: enc_mlp(feats).mean(dim=1)    # [2B, K, 128] -> [2B, D]
loss_fn = nn.BCEWithLogitsLoss()

for epoch in range(3):
    stats = []
    def on_batch(b, neg, res):
        _, (pos_logit, neg_logit) = res
        logits = torch.cat([pos_logit, neg_logit]).squeeze(-1)
        labels = torch.cat([torch.ones_like(pos_logit), torch.zeros_like(neg_logit)]).squeeze(-1)
        loss = loss_fn(logits, labels)
        opt.zero_grad(); loss.backward(); opt.step()
        stats.append((loss.item(), ((logits > 0).float() == labels).float().mean().item()))
    t0 = time.perf_counter()
    run_epoch(rp, sampler, negs, src, dst, t, B, K, encoder=encoder, decoder=decoder, on_batch=on_batch)
    torch.cuda.synchronize()
    half = len(stats) // 2
    print(f"epoch {epoch}: loss {np.mean([s[0] for s in stats[half:]]):.4f}  accuracy {np.mean([s[1] for s in stats[half:]]):.3f}"
          f"  ({E / (time.perf_counter() - t0) / 1e3:.0f} k edges/s end to end, python loop)")
