#!/usr/bin/env python3
"""An epoch loop on a device-resident stream (what train_link_prediction.py:234-253 does every epoch -- reset the projections, run
the same chronological batches, draw new negatives) with the stream call PREPARED once: `rp.prepare_stream(...)` checks the arguments and
allocates the feature rows once, every epoch then writes its negatives into the tensor the call holds and calls it.  The first epoch plans
the stream, every later one replays the plan.  usage: python examples/epoch_stream.py [epochs]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd                                            # noqa: E402
from tpnet_amd.stream import CONFIGS, synthetic_negatives, synthetic_stream   # noqa: E402


def main(epochs: int = 4):
    cfg = CONFIGS["C2"]
    B, d = cfg["B"], cfg["d"]
    src, dst, t, N = synthetic_stream(cfg["U"], cfg["I"], cfg["E"], cfg["span"], 0)
    dev = torch.device("cuda:0")
    rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=cfg["E"], dim_factor=10, num_layer=3, time_decay_weight=cfg["lam"],
                                          device=str(dev), use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                          enforce_dim=d).to(dev)
    to_dev = lambda x: torch.from_numpy(x).to(dev)
    d_src, d_dst, d_t = to_dev(src), to_dev(dst), to_dev(t)
    d_neg = torch.empty_like(d_src)
    call = rp.prepare_stream(d_src, d_dst, d_neg, d_t, B, t_end=float(t[-1]))      # checked once; holds its tensors and outputs
    for ep in range(epochs):
        call.neg.copy_(to_dev(synthetic_negatives(cfg["U"], N, len(src), B, 1 + ep)))   # this epoch's negatives, in place
        rp.reset_random_projections()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        feat_pos, feat_neg = call()                           # [E, 64] each: the pre-mlp pairwise features of every batch
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print(f"epoch {ep}: {len(src) / el / 1e6:.0f} M edges/s ({el * 1e6:.0f} us), plan replayed: {rp.last_stream_replayed}, "
              f"features {tuple(feat_pos.shape)} mean {float(feat_pos.mean()):.4f} / {float(feat_neg.mean()):.4f}", flush=True)
    rp.check_device_errors()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4)
