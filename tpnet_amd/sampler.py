"""Device-side 'recent' neighbour sampler (SURVEY.md §8 f-3, C ABI: tpnet_sampler_build / tpnet_sample_recent).

Drop-in for what TPNet's encoder asks of the reference's `NeighborSampler` built by `get_neighbor_sampler(data,
'recent')` (utils/utils.py:82-224, 293-312; call site models/TPNet.py:291-294): the same
`get_historical_neighbors(node_ids, node_interact_times, num_neighbors)` -> three [n, K] arrays.  The adjacency lives
in HBM as one CSR; a query batch is one kernel launch instead of a Python loop over the nodes."""
import ctypes as C

import numpy as np
import torch

from . import _lib


class GpuRecentNeighborSampler:
    sample_neighbor_strategy = "recent"      # attributes TPNet.set_neighbor_sampler looks at (models/TPNet.py:223-232)
    seed = None

    def __init__(self, src_node_ids, dst_node_ids, node_interact_times, edge_ids=None, device="cuda:0",
                 num_nodes: int = None):
        lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.TPNetHipError("GpuRecentNeighborSampler needs a cuda device (no CPU fallback)")
        to = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt).to(self.device) \
            if not isinstance(a, torch.Tensor) else a.to(self.device, dt).contiguous()
        src, dst = to(src_node_ids, torch.int64), to(dst_node_ids, torch.int64)
        t = to(node_interact_times, torch.float64)
        eid = to(edge_ids, torch.int64) if edge_ids is not None else None
        self.E = int(src.numel())
        if num_nodes is None:
            num_nodes = (int(max(src.max().item(), dst.max().item())) + 1) if self.E else 1
        self.num_nodes = int(num_nodes)
        nbytes = lib.tpnet_sampler_bytes(self.E, self.num_nodes)
        self._buf = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(lib.tpnet_sampler_build(self._buf.data_ptr(), nbytes, src.data_ptr(), dst.data_ptr(), t.data_ptr(),
                                           eid.data_ptr() if eid is not None else None, self.E, self.num_nodes, stream),
                   "sampler_build")
        self._keep = (src, dst, t, eid)      # inputs must outlive the asynchronous build

    def sample_device(self, node_ids: torch.Tensor, times: torch.Tensor, num_neighbors: int, with_edges: bool = True):
        """node_ids int64 [n], times float64 [n] on the device -> (ids, edge_ids, times), each [n, K] on the device."""
        assert num_neighbors > 0, "Number of sampled neighbors for each node should be greater than 0!"
        n = int(node_ids.numel())
        ids = torch.empty((n, num_neighbors), dtype=torch.int64, device=self.device)
        eids = torch.empty((n, num_neighbors), dtype=torch.int64, device=self.device) if with_edges else None
        ts = torch.empty((n, num_neighbors), dtype=torch.float64, device=self.device) if with_edges else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(_lib.load().tpnet_sample_recent(self._buf.data_ptr(), self.E, self.num_nodes, node_ids.data_ptr(),
                                                   times.data_ptr(), n, num_neighbors, ids.data_ptr(),
                                                   eids.data_ptr() if with_edges else None,
                                                   ts.data_ptr() if with_edges else None, stream), "sample_recent")
        return ids, eids, ts

    def get_historical_neighbors(self, node_ids: np.ndarray, node_interact_times: np.ndarray, num_neighbors: int = 20):
        """The reference's signature and return types (numpy [n, K] x 3): utils/utils.py:160-224."""
        nid = torch.as_tensor(np.ascontiguousarray(node_ids), dtype=torch.int64).to(self.device)
        tq = torch.as_tensor(np.ascontiguousarray(node_interact_times), dtype=torch.float64).to(self.device)
        ids, eids, ts = self.sample_device(nid, tq, num_neighbors)
        return ids.cpu().numpy(), eids.cpu().numpy(), ts.cpu().numpy()
