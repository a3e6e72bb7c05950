#!/usr/bin/env python3
"""Developer tool: get_pair_wise_feature (readout + self.mlp) on device-resident ids, four ways: readout kernel + torch fp32
layers; readout kernel + bf16 MFMA mlp kernel; ONE kernel with the mlp on the bf16 matrix cores (tpnet_pair_feature_bf16);
ONE kernel in fp32 on the vector ALUs (tpnet_pair_feature, short lists).  HIP-event timing."""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tpnet_amd
from tpnet_amd import _lib, fused_mlp as fm, fused_feature as ff
from tpnet_amd.stream import CONFIGS, synthetic_stream
lib = _lib.load()
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for cfg, ns in (("C2", (1000, 80000)), ("C3", (10000, 800000)), ("C5", (10000, 200000))):
    c = CONFIGS[cfg]; B = c["B"]; E = 6 * B
    src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0)
    dev = torch.device("cuda:0")
    rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
            device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
    D = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    rp.run_stream(D(src), D(dst), None, D(t), B, want_neg=False, want_pos=False)
    rng = np.random.RandomState(0)
    for n in ns:
        u, v = D(rng.randint(1, N, n).astype(np.int64)), D(rng.randint(1, N, n).astype(np.int64))
        with torch.no_grad():
            rp.fused_mlp = False
            t_torch = timeit(lambda: rp.mlp(rp.pair_gram(u, v)))
            t_two = timeit(lambda: fm.fused_mlp(rp.mlp, rp.pair_gram(u, v)))
            rp.fused_mlp = True
            t_one = timeit(lambda: rp.get_pair_wise_feature(u, v))
            rp.fused_mlp = False
            t_gram = timeit(lambda: rp.pair_gram(u, v))
            line = (f"{cfg} d={c['d']} n={n}: readout alone {t_gram:.1f} us; + torch fp32 mlp {t_torch:.1f}; + bf16 mlp kernel {t_two:.1f}; "
                    f"ONE kernel, mlp on the matrix cores {t_one:.1f}")
            prep = ff.prepared(rp.mlp, 64)
            out = torch.empty((n, 64), device=dev)
            t_f32 = timeit(lambda: lib.tpnet_pair_feature(rp._st_ref(), u.data_ptr(), v.data_ptr(), n, rp._now_host, float(c["lam"]),
                                                          0, prep[2], None, out.data_ptr(), rp._stream()))
            gram = rp.pair_gram(u, v)
            t_mlp = timeit(lambda: ff.mlp_f32(rp.mlp, gram))
            line += f"; ONE kernel fp32 ({'matrix cores' if n >= 2048 else 'vector ALUs'}) {t_f32:.1f}; dense layers alone on the fp32 matrix cores {t_mlp:.1f}"
        print(line, flush=True)
