import os, sys, time
import numpy as np, torch, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import tpnet_amd, bench
from tpnet_amd import _lib
from tpnet_amd import random_projection as R
from tpnet_amd.stream import CONFIGS
c = CONFIGS["C2"]; B = c["B"]; K = 20
src, dst, neg, t, N = bench.make_workload(c, K + 5, 0)
dev = torch.device("cuda:0")
rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
        device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
d = [torch.from_numpy(np.ascontiguousarray(x[:K * B])).to(dev) for x in (src, dst, neg, t)]
NG = rp.pair_wise_feature_dim
o_p = torch.empty((K * B, NG), dtype=torch.float32, device=dev); o_n = torch.empty_like(o_p)
rp.run_stream(d[0], d[1], d[2], d[3], B, out_pos=o_p, out_neg=o_n, t_end=1.0, replay=False); torch.cuda.synchronize()
def T(name, f, n=20000):
    t0 = time.perf_counter()
    for _ in range(n): f()
    print(f"{name:28s} {(time.perf_counter() - t0) / n * 1e6:6.2f} us")
s_, d_, n_, t_ = d
E = K * B
def checks():
    for name, x, dt in (("src", s_, torch.int64), ("dst", d_, torch.int64), ("t", t_, torch.float64)):
        if x.device != dev or x.dtype != dt or not x.is_contiguous() or x.numel() != E: raise ValueError
    if n_ is not None and (n_.device != dev or n_.dtype != torch.int64 or n_.numel() != E or not n_.is_contiguous()): raise ValueError
    for name, o, want in (("out_pos", o_p, True), ("out_neg", o_n, True)):
        if want and o is not None and (o.dtype != torch.float32 or o.device != dev or not o.is_contiguous() or tuple(o.shape) != (E, NG)): raise ValueError
T("checks", checks)
T("_ensure_engine", rp._ensure_engine)
T("_dev", rp._dev)
T("numel", lambda: int(s_.numel()))
T("pair_wise_feature_dim", lambda: rp.pair_wise_feature_dim)
T("_workspace", lambda: rp._workspace(E, B, stream=True, keep_plan=True))
T("_st_ref", rp._st_ref)
T("_next_launch_ids", lambda: rp._next_launch_ids(20))
T("hash", lambda: hash((s_.data_ptr(), s_._version, d_.data_ptr(), d_._version, t_.data_ptr(), t_._version, E)))
T("_drop_plan", rp._drop_plan)
T("_lib.fast", _lib.fast)
T("_raw_stream", lambda: R._raw_stream(rp._eng["dev_index"]))
T("6 data_ptr", lambda: (s_.data_ptr(), d_.data_ptr(), n_.data_ptr(), t_.data_ptr(), o_p.data_ptr(), o_n.data_ptr()))
T("float(twd)", lambda: float(rp.time_decay_weight))
T("_table_written", rp._table_written)
def sets():
    rp.last_stream_replayed = False; rp._now_host = 1.0; rp._params_valid = False; rp._now_dirty = True
T("4 setattr", sets)
T("slice x6 + float", lambda: (s_[0:E], d_[0:E], n_[0:E], t_[0:E], o_p[:E], o_n[:E], float(t[E - 1])))
T("c_double", lambda: C.c_double(0.0))
