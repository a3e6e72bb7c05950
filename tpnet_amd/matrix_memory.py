"""Drop-in for the reference's `MatrixMemory` (models/MemoryModel.py:364-420), PINT's walk-matrix state -- the sibling
of TPNet's random projections: a dense [N, N, hop+1] tensor whose rows receive shifted copies of
their partners' rows for every observed edge.

Same constructor, attributes (`matrix`, `P`: the state-dict keys), methods and results as the reference class.  The
update runs on the TPNet engine unchanged: with hop h stored as layer H-h of an N x N table,

    matrix[u, :, k] += matrix[v, :, k+1]   (k = H-1 .. 0, pre-batch values, MemoryModel.py:392-394)

IS  `P[i][u] += w * P[i-1][v]`  (i = 1..H, models/TPNet.py:87-97) with every time weight and decay factor equal to 1
(time_decay_weight = 0 makes them exp(0) exactly), so `tpnet_update` does it with its plan, its ping-pong bundles and no
atomics; `get_memory` is one element gather (`tpnet_gather_elems`).  The Parameter `matrix` is the reference-visible
copy: it is read into the engine when somebody wrote it (reload, reset, load_state_dict) and written back when somebody
looks at it (backup, state_dict, attribute access).  update / get_memory need a GPU (no CPU fallback).
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .random_projection import RandomProjectionModule


class MatrixMemory(nn.Module):
    def __init__(self, num_node: int, num_hop: int, device: str):
        super().__init__()
        if not 1 <= num_hop <= _lib.TPNET_MAX_LAYERS:
            raise ValueError(f"num_hop must be in 1..{_lib.TPNET_MAX_LAYERS} (got {num_hop})")
        self.num_node = num_node
        self.num_hop = num_hop
        self.device = device
        self.matrix = nn.Parameter(torch.zeros((num_node, num_node, num_hop + 1)), requires_grad=False)
        self.P = nn.Parameter(torch.zeros(num_hop + 1, num_hop + 1), requires_grad=False)
        self.P.data[1:, :-1] = torch.eye(self.num_hop)                 # MemoryModel.py:376-377
        # engine side (plain attributes: the state dict stays {matrix, P})
        self.__dict__["_rp"] = None
        self.__dict__["_engine_valid"] = False       # the engine holds the truth
        self.__dict__["_matrix_valid"] = True        # the Parameter holds the truth
        self.__dict__["_sig"] = None
        self.reset_memory()

    # ---- plumbing -------------------------------------------------------------------------------------------------
    def _mparam(self) -> nn.Parameter:
        return self._parameters["matrix"]

    def __getattr__(self, name):
        if name == "matrix" and "_parameters" in self.__dict__:
            # a handed-out Parameter may be written in place through `.data` (the reference's own idiom,
            # MemoryModel.py:381-384), which neither data_ptr nor _version shows: re-import before the next engine call
            self.__dict__["_exposed"] = True
            if not self.__dict__.get("_matrix_valid", True):
                self._materialize()
        return super().__getattr__(name)

    def _signature(self):
        p = self._mparam()
        return (p.data_ptr(), p._version, p.device)

    def _ensure_engine(self):
        p = self._mparam()
        if p.device.type != "cuda":
            raise _lib.TPNetHipError("MatrixMemory (tpnet_amd) computes only on a GPU: move the module to a cuda device "
                                     "first; there is no CPU fallback")
        if self._engine_valid and (not self._matrix_valid or (self._sig == self._signature() and not self.__dict__.get("_exposed", False))):
            return
        N, H = self.num_node, self.num_hop
        rp = self.__dict__["_rp"]
        if rp is None or rp._plist()[0].device != p.device:
            rp = RandomProjectionModule.__new__(RandomProjectionModule)
            _bare_table(rp, N, H, p.device)
            self.__dict__["_rp"] = rp
        with torch.no_grad():
            for i in range(H + 1):                                     # layer i = hop H - i
                rp._plist()[i].data = p.data[:, :, H - i].contiguous()
        rp._engine_valid = False                                       # import on next use
        rp._params_valid = True
        self.__dict__["_engine_valid"] = True
        self.__dict__["_matrix_valid"] = True
        self.__dict__["_exposed"] = False
        self.__dict__["_sig"] = self._signature()

    def _materialize(self):
        if self._matrix_valid:
            return
        rp, H = self.__dict__["_rp"], self.num_hop
        p = self._mparam()
        with torch.no_grad():
            layers = rp.random_projections                              # materialises the engine's layers
            for i in range(H + 1):
                p.data[:, :, H - i].copy_(layers[i].data)
        self.__dict__["_matrix_valid"] = True
        self.__dict__["_sig"] = self._signature()

    def state_dict(self, *args, **kwargs):
        self._materialize()
        self.__dict__["_exposed"] = True       # the returned tensors alias the Parameter's storage
        return super().state_dict(*args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)                 # writes the Parameter: it is the truth again
        self.__dict__["_matrix_valid"] = True
        self.__dict__["_engine_valid"] = False

    def _apply(self, fn, *args, **kwargs):
        self._materialize()
        out = super()._apply(fn, *args, **kwargs)
        self.__dict__["_engine_valid"] = False
        return out

    # ---- the reference's methods ----------------------------------------------------------------------------------
    def reset_memory(self):
        """MemoryModel.py:379-385: hop 0 becomes the identity (the other hops are left as they are)."""
        self._materialize()
        self._mparam().data[:, :, 0] = torch.eye(self.num_node, device=self._mparam().device)
        self.__dict__["_engine_valid"] = False

    def update(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray):
        """MemoryModel.py:387-394."""
        if len(src_node_ids) != len(dst_node_ids):
            raise ValueError("src_node_ids and dst_node_ids must have the same length")
        if len(src_node_ids) == 0:
            return
        self._ensure_engine()
        rp = self.__dict__["_rp"]
        rp.update(src_node_ids=src_node_ids, dst_node_ids=dst_node_ids,
                  node_interact_times=np.zeros(len(src_node_ids), dtype=np.float64))
        self.__dict__["_matrix_valid"] = False

    def get_memory(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray):
        """MemoryModel.py:396-405: matrix[src, dst] / (its sum over the hops + 1e-4), shape [n, hop+1]."""
        if len(src_node_ids) != len(dst_node_ids):
            raise ValueError("src_node_ids and dst_node_ids must have the same length")
        self._ensure_engine()
        rp = self.__dict__["_rp"]
        rp._ensure_engine()
        u, v = rp._to_device(rp._check_ids(src_node_ids, "src_node_ids"), rp._check_ids(dst_node_ids, "dst_node_ids"))
        n = u.numel()
        out = torch.empty((n, self.num_hop + 1), dtype=torch.float32, device=rp._dev())
        st = rp._state()
        _lib.check(_lib.load().tpnet_gather_elems(C.byref(st), u.data_ptr(), v.data_ptr(), n, rp._now_host, 0.0,
                                                  out.data_ptr(), rp._stream()), "gather_elems")
        m = out.flip(1)                                                  # layer i -> hop H - i
        return m / (torch.sum(m, dim=1, keepdim=True) + 1e-4)

    def backup_memory(self):
        """MemoryModel.py:407-412."""
        return self.matrix.data.clone()

    def reload_memory(self, data):
        """MemoryModel.py:414-419."""
        self._mparam().data = data.clone()
        self.__dict__["_matrix_valid"] = True
        self.__dict__["_engine_valid"] = False


def _bare_table(rp: RandomProjectionModule, N: int, H: int, device):
    """Initialise `rp` as an N x N table of H+1 zero layers on `device` WITHOUT the constructor's random draw of layer 0
    (N*N normals on the host): time_decay_weight = 0, so every time weight and decay factor is exp(0) = 1."""
    nn.Module.__init__(rp)
    rp.node_num, rp.edge_num, rp.dim, rp.num_layer = N, 1, N, H
    rp.time_decay_weight = 0.0
    rp.begging_time = nn.Parameter(torch.tensor(np.float64(0.0)), requires_grad=False)
    rp.now_time = nn.Parameter(torch.tensor(np.float64(0.0)), requires_grad=False)
    rp.device = str(device)
    rp.use_matrix = False
    rp.node_feature_dim = 128
    rp.not_scale = True
    rp.exact = False
    rp.fused_mlp = False
    rp.random_projections = nn.ParameterList(
        [nn.Parameter(torch.zeros((1, 1), device=device), requires_grad=False) for _ in range(H + 1)])
    rp.pair_wise_feature_dim = (2 * H + 2) ** 2
    rp.mlp = nn.Identity()
    rp._eng = None
    rp._engine_valid = False
    rp._params_valid = True
    rp._param_sig = None
    rp._now_host = 0.0
    rp._launch_id = 1
    rp.to(device)
