"""Generator of the golden vectors of SURVEY.md §8 row f-4 (tests/golden/g9_*.npz, g10_*.npz).

Runs ONLY in the build container, where the read-only reference checkout is mounted at /root/reference:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_f4.py

G9:  the reference's `models.MemoryModel.MatrixMemory` (PINT's walk-matrix state, models/MemoryModel.py:364-420) on
     seeded inputs: the matrix after every update, `get_memory` outputs, backup / reload -- from the reset state (where
     the reference's shift matrix makes every message zero) and from a reloaded non-trivial matrix.
G10: the state-dict layout of the model the training script saves (train_link_prediction.py:169-217,
     utils/EarlyStopping.py:64-87): key names and shapes of Sequential(holder-of-rp, LinkPredictor_v1(rp)) built from the
     reference's classes, and the size of the file `torch.save` writes for it.
Data only; the reference never travels."""
import io
import os
import sys

import numpy as np
import torch
import torch.nn as nn

REF = os.environ.get("TPNET_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
from models.MemoryModel import MatrixMemory  # noqa: E402
from models.TPNet import RandomProjectionModule  # noqa: E402
from models.modules import LinkPredictor_v1  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def matrix_memory(name, N, H, nb, B, seed):
    rng = np.random.RandomState(seed)
    src = rng.randint(1, N, nb * B).astype(np.int64)
    dst = rng.randint(1, N, nb * B).astype(np.int64)
    src[rng.rand(nb * B) < 0.3] = 2                       # a hub: many messages per batch, duplicates in the index
    dst[::7] = src[::7]                                   # self pairs
    qs = rng.randint(0, N, 40).astype(np.int64)
    qd = rng.randint(0, N, 40).astype(np.int64)
    qd[::5] = qs[::5]
    out = dict(N=N, H=H, B=B, src=src, dst=dst, qs=qs, qd=qd)
    with torch.no_grad():
        mm = MatrixMemory(num_node=N, num_hop=H, device='cpu')
        out["P_shift"] = mm.P.detach().numpy().copy()
        out["reset_matrix"] = mm.matrix.detach().numpy().copy()
        # (a) from the reset state
        half = nb // 2
        mats, mems = [], []
        for b in range(half):
            s = slice(b * B, (b + 1) * B)
            mm.update(src[s], dst[s])
            mats.append(mm.matrix.detach().numpy().copy())
            mems.append(mm.get_memory(qs, qd).detach().numpy().copy())
        out["a_matrix"] = np.stack(mats)
        out["a_memory"] = np.stack(mems)
        # (b) from a reloaded, non-trivial matrix (counts of walks are non-negative)
        data = torch.from_numpy((rng.rand(N, N, H + 1) * (rng.rand(N, N, H + 1) < 0.3)).astype(np.float32))
        out["b_reload"] = data.numpy().copy()
        mm.reload_memory(data)
        backup = None
        mats, mems = [], []
        for b in range(half, nb):
            s = slice(b * B, (b + 1) * B)
            if b == half + 1:
                backup = mm.backup_memory()
            mm.update(src[s], dst[s])
            mats.append(mm.matrix.detach().numpy().copy())
            mems.append(mm.get_memory(qs, qd).detach().numpy().copy())
        out["b_matrix"] = np.stack(mats)
        out["b_memory"] = np.stack(mems)
        out["b_backup_after_first"] = backup.numpy().copy()
        mm.reload_memory(backup)
        mm.update(src[(half + 1) * B:(half + 2) * B], dst[(half + 1) * B:(half + 2) * B])
        out["b_replayed"] = mm.matrix.detach().numpy().copy()       # must equal b_matrix[1]
        mm.reset_memory()
        out["b_after_reset"] = mm.matrix.detach().numpy().copy()    # reset only rewrites hop 0 (MemoryModel.py:381-385)
    np.savez_compressed(os.path.join(OUT, name), **out)


def checkpoint_layout(name):
    torch.manual_seed(0)
    N, d, L = 40, 16, 3
    rp = RandomProjectionModule(node_num=N, edge_num=500, dim_factor=10, num_layer=L, time_decay_weight=1e-6,
                                device='cpu', use_matrix=False, beginning_time=np.float64(0.0), not_scale=False,
                                enforce_dim=d)

    class Holder(nn.Module):                   # stands in for the backbone: it registers rp the way TPNet does
        def __init__(self, rp_):                # (models/TPNet.py:176: self.random_projections = random_projections)
            super().__init__()
            self.random_projections = rp_

    lp = LinkPredictor_v1(input_dim1=8, input_dim2=8, hidden_dim=8, output_dim=1, random_projections=rp,
                          not_encode=False)
    model = nn.Sequential(Holder(rp), lp)
    sd = model.state_dict()
    buf = io.BytesIO()
    torch.save({'model': sd, 'args': {}}, buf)                      # utils/EarlyStopping.py:72-75
    one_copy = sum(p.numel() * p.element_size() for p in rp.state_dict().values())
    np.savez_compressed(os.path.join(OUT, name), keys=np.array(list(sd.keys())),
                        shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
                        file_bytes=buf.tell(), rp_bytes_one_copy=one_copy,
                        rp_keys=np.array(list(rp.state_dict().keys())))


if __name__ == "__main__":
    torch.set_num_threads(1)
    matrix_memory("g9_matrix_memory_N24_H3.npz", N=24, H=3, nb=8, B=12, seed=21)
    matrix_memory("g9_matrix_memory_N17_H2.npz", N=17, H=2, nb=6, B=9, seed=22)
    checkpoint_layout("g10_checkpoint_layout.npz")
    print("f-4 golden vectors written to", OUT)
