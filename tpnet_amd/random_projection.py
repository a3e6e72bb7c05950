"""Drop-in `RandomProjectionModule` backed by the gfx950 HIP kernels (C ABI: include/tpnet_hip.h).

Mirrors the reference operator interface of /root/reference/models/TPNet.py:9-157 -- same constructor
arguments, method names, argument meaning, attributes and state-dict keys -- so that the reference's callers
(train_link_prediction.py:137-145,248,372,403-494; evaluate_models_utils.py:183; models/TPNet.py:313;
models/modules.py:112) work unchanged.  There is no CPU fallback: every compute method needs the HIP library
and a GPU-resident module, and raises otherwise.

How the state is held:
  * `random_projections[0]` (P[0]) IS the kernels' layer-0 buffer (no copy).
  * layers 1..L live in the engine's own layout (`_q`: ping-pong per-node bundles + `_meta`), where the
    reference's dense per-batch decay (TPNet.py:83-85) is carried lazily per row.  The `random_projections[1..L]`
    Parameters are materialised from it (one export pass, decay applied) whenever somebody looks at them
    (attribute access, state_dict(), backup, .to()), and imported back when somebody writes them
    (reload, load_state_dict, .to(), direct `.data` assignment -- detected through data_ptr/_version).
"""
import ctypes as C
import math
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import fused_feature as _ff

_MAX_LAUNCH_ID = 0x7FFFFFFF - (1 << 24)


# the current stream's raw handle without building a torch.cuda.Stream object (the hot methods need it every call)
_BIG_SLOT_BYTES = 1 << 20      # one encoder call from host arrays: up to 100 000 neighbour ids + their anchors
# where the per-batch staging ring of the host-array calls lives (tpnet_stage_create_ex): -1 = device memory written through the
# large BAR where the device has one (the kernels' first loads are local: ~1.5 us less at the head of each of a batch's three
# kernels, ~1.6 us more host time per call -- the per-batch loop is GPU-bound), 0 = pinned host memory, 1 = device memory or fail
STAGE_MODE = -1


def _ptr_or_null(t):
    return t.data_ptr() if t is not None else None


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or (lambda i: torch.cuda.current_stream(i).cuda_stream)


_OWNERS = weakref.WeakValueDictionary()      # id(module) -> module, for the lazy layer list below
_WARMED = set()                               # devices whose HIP runtime pools were warmed (tpnet_runtime_warmup)


class _LazyLayers(nn.ParameterList):
    """`random_projections` as the reference has it (an nn.ParameterList of L+1 [N, d] Parameters, models/TPNet.py:39-62)
    whose entries 1..L are brought up to date only when somebody actually reads one of THEM: the engine keeps those layers
    in its own layout, and the dense export (N*L*d floats) is not something a caller that touches `random_projections[0]`,
    `len(...)` or `.device` once per batch should pay for.  Reading an entry >= 1 (or iterating) materialises them and marks
    them as handed out (they may then be written in place through `.data`: the next engine call re-imports them)."""

    def _tp_owner(self):
        return _OWNERS.get(self.__dict__.get("_tp_owner_key"))

    def _tp_touch(self, idx):
        if isinstance(idx, int) and (idx == 0 or idx == -len(self)):
            return                                   # layer 0 IS the kernels' buffer: always current
        o = self._tp_owner()
        if o is not None:
            o._layers_read()

    def __getitem__(self, idx):
        self._tp_touch(idx)
        return super().__getitem__(idx)

    def __iter__(self):
        self._tp_touch(None)
        return super().__iter__()


class _Stage:
    """The pinned, device-mapped staging ring of the host-array entry points (tpnet_stage_*): 8 slots of 256 KB, i.e. up to
    16 384 pairs or one batch of up to 2 048 edges per call.  The only thing the C library allocates; freed with the engine."""

    def __init__(self, lib, dev, slots: int = 8, slot_bytes: int = 256 * 1024, mode: int = 0):
        self._lib = lib
        h = C.c_void_p()
        with torch.cuda.device(dev):
            _lib.check(lib.tpnet_stage_create_ex(slots, slot_bytes, int(mode), C.byref(h)), "stage_create")
        self.handle = h
        self.max_pairs = int(lib.tpnet_stage_max_pairs(h))
        self.max_batch = int(lib.tpnet_stage_max_batch(h))      # batches planned by one workgroup, read straight from the slot
        self.max_host_batch = slot_bytes // 24                  # larger ones: staged, copied to the workspace, chunk planner

    def __del__(self):
        try:
            if self.handle:
                self._lib.tpnet_stage_destroy(self.handle)
                self.handle = None
        except Exception:       # interpreter shutdown: the runtime may be gone already
            pass


class PreparedStream:
    """A run_stream call whose arguments have been checked and whose outputs exist (RandomProjectionModule.prepare_stream): calling
    it runs the stream.  Attributes: the tensors it reads (`src`, `dst`, `neg`, `t`: their contents are read at every call) and
    writes (`out_pos`, `out_neg`: what the call returns)."""
    __slots__ = ("rp", "src", "dst", "neg", "t", "E", "batch_size", "nb", "want_pos", "want_neg", "out_pos", "out_neg", "t_end",
                 "flags", "replay", "exact", "dev", "ptrs", "ws_need", "ws_cap")

    def __call__(self):
        return self.rp._run_prepared(self)

    def __setattr__(self, name, value):
        # the call's device pointers were taken when it was prepared: a tensor swapped in afterwards would not be the one that is read
        # (write INTO the held tensor -- call.neg.copy_(new_negatives) -- or prepare the stream again)
        if hasattr(self, name):
            raise AttributeError(f"PreparedStream.{name} is fixed once the call is prepared: copy into the held tensor or prepare again")
        object.__setattr__(self, name, value)


class RandomProjectionModule(nn.Module):
    # plan-replay bookkeeping (class-level defaults: tpnet_amd/matrix_memory.py builds instances without this constructor)
    _table_sig = 0
    # bound on the windowed schedule's version log per chunk when run_stream sizes its workspace (None: the library's 16 GiB); a
    # workspace that is already larger keeps its chunk length -- the C side takes the longest chunk the workspace holds
    stream_log_cap_bytes = None
    _sig_counter = 1
    _plan_tag = None

    def __init__(self, node_num: int, edge_num: int, dim_factor: int, num_layer: int, time_decay_weight: float,
                 device: str, use_matrix: bool, beginning_time: np.float64, not_scale: bool, enforce_dim: int,
                 exact: bool = False, alloc_device=None):
        """Arguments as in the reference (models/TPNet.py:10-26).  `exact=True` (extension, default off) selects
        the reference's literal arithmetic: an eager dense decay of every row per update (TPNet.py:83-85) and
        strictly index-ordered sums -- used by the parity tests; the default carries the decay lazily per row.
        `alloc_device` (extension, default None = the host, like the reference, whose callers then move the module:
        utils/utils.py:43) builds the tables directly on that device: a 10 M-node table is 41 GB that need not pass
        through host memory; P[0] is then drawn from that device's generator."""
        super().__init__()
        if not 1 <= num_layer <= _lib.TPNET_MAX_LAYERS:
            raise ValueError(f"num_layer must be in 1..{_lib.TPNET_MAX_LAYERS} (got {num_layer})")
        self.node_num = node_num
        self.edge_num = edge_num
        if enforce_dim != -1:
            self.dim = enforce_dim
        else:
            self.dim = min(int(math.log(self.edge_num * 2)) * dim_factor, node_num)
        self.num_layer = num_layer
        self.time_decay_weight = time_decay_weight
        self.begging_time = nn.Parameter(torch.tensor(beginning_time), requires_grad=False)
        self.now_time = nn.Parameter(torch.tensor(beginning_time), requires_grad=False)
        self.device = device
        self.random_projections = _LazyLayers()
        self.use_matrix = use_matrix
        self.node_feature_dim = 128
        self.not_scale = not_scale
        self.exact = bool(exact)
        self.fused_mlp = False               # opt-in: self.mlp on the bf16 matrix cores (tpnet_amd/fused_mlp.py)
        if self.use_matrix:
            self.dim = self.node_num
            for i in range(self.num_layer + 1):
                if i == 0:
                    self.random_projections.append(nn.Parameter(torch.eye(self.node_num, device=alloc_device),
                                                                requires_grad=False))
                else:
                    self.random_projections.append(
                        nn.Parameter(torch.zeros_like(self.random_projections[i - 1]), requires_grad=False))
        else:
            for i in range(self.num_layer + 1):
                if i == 0:
                    self.random_projections.append(
                        nn.Parameter(torch.normal(0, 1 / math.sqrt(self.dim), (self.node_num, self.dim),
                                                  device=alloc_device), requires_grad=False))
                else:
                    self.random_projections.append(
                        nn.Parameter(torch.zeros_like(self.random_projections[i - 1]), requires_grad=False))
        self.pair_wise_feature_dim = (2 * self.num_layer + 2) ** 2
        self.mlp = nn.Sequential(nn.Linear(self.pair_wise_feature_dim, self.pair_wise_feature_dim * 4), nn.ReLU(),
                                 nn.Linear(self.pair_wise_feature_dim * 4, self.pair_wise_feature_dim))
        # ---- engine side (plain attributes: not parameters/buffers, so the state-dict keys match the reference)
        self._eng = None                      # dict of device tensors, allocated on first use
        self._engine_valid = False            # engine holds the truth for layers 1..L
        self._params_valid = True             # the Parameters hold the truth for layers 1..L
        self._param_sig = None                # (data_ptr, _version) of the layer Parameters at the last sync
        self._now_host = float(beginning_time)
        self._launch_id = 1
        self._now_dirty = False
        self._params_exposed = False          # the ParameterList was handed out since the last import (see __getattr__)
        self._table_sig = 0                   # identifies the table's per-node (copy, reference time) state (plan replay)
        self._sig_counter = 1
        self._plan_tag = None                 # _lib.PlanTag of the plan the stream workspace holds

    # ------------------------------------------------------------------------------------------------------
    # plumbing
    # ------------------------------------------------------------------------------------------------------
    def _plist(self):
        # the Parameter OBJECTS of the list are stable (`.data = ...`, `.to()`, load_state_dict keep them), so they are
        # looked up once: nn.ParameterList.__getitem__ costs microseconds and the hot methods need them every call
        refs = self.__dict__.get("_param_refs")
        if refs is None or len(refs) != self.num_layer + 1:
            refs = list(nn.ParameterList.__iter__(self._modules["random_projections"]))   # (not the lazy list's own iterator)
            self.__dict__["_param_refs"] = refs
        return refs

    # bookkeeping attributes of the hot methods (plain Python values, never Parameters / Modules / buffers): stored without
    # nn.Module.__setattr__'s type checks -- six assignments per update() call, ~0.7 us each, in a loop the host bounds
    _PLAIN_ATTRS = frozenset(("_engine_valid", "_params_valid", "_now_host", "_now_dirty", "_param_sig", "_launch_id", "_table_sig",
                              "_params_exposed", "_sig_counter", "last_stream_replayed"))

    def __setattr__(self, name, value):
        if name in RandomProjectionModule._PLAIN_ATTRS:
            self.__dict__[name] = value
        else:
            super().__setattr__(name, value)

    def __getattr__(self, name):
        # external readers of `random_projections` / `now_time` see the reference's eager values
        if name == "random_projections" and "_modules" in self.__dict__:
            # the list itself is handed out as is; its entries 1..L are materialised when one of them is read (_LazyLayers)
            pl = self.__dict__["_modules"].get("random_projections")
            if isinstance(pl, _LazyLayers):
                pl.__dict__["_tp_owner_key"] = id(self)
                _OWNERS[id(self)] = self
            else:                                    # (a plain list put there by somebody else: the eager behaviour)
                self._layers_read()
        elif name == "now_time" and self.__dict__.get("_now_dirty", False):
            self._sync_now_time()
        return super().__getattr__(name)

    def _layers_read(self):
        """An entry 1..L of `random_projections` is being handed out: bring the Parameters up to date, and remember that
        whoever got one may write it in place through `.data` (p.data.copy_(), p.data[i] = ...: the idiom of the reference's
        MatrixMemory), which changes neither data_ptr nor _version -- the next engine call re-imports the layers."""
        self.__dict__["_params_exposed"] = True
        if not self.__dict__.get("_params_valid", True):
            self._materialize()

    def _sync_now_time(self):
        """The clock Parameter is written lazily: the hot methods keep the clock on the host (`_now_host`) and mark the
        Parameter stale instead of launching a fill kernel per call; whoever reads it (attribute access, state_dict, backup,
        copy, .to()) gets it filled first."""
        if self.__dict__.get("_now_dirty", False):
            self.__dict__["_now_dirty"] = False
            self._parameters["now_time"].data.fill_(self._now_host)

    def _sig(self):
        return tuple((p.data_ptr(), p._version, p.device) for p in self._plist())

    def _dev(self) -> torch.device:
        dev = self._plist()[0].device
        if dev.type != "cuda":
            raise _lib.TPNetHipError(
                "RandomProjectionModule (tpnet_amd) computes only on a GPU: move the module to a cuda device "
                "first (the reference script does this in utils/utils.py:43); there is no CPU fallback")
        return dev

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._dev()).cuda_stream)

    def _engine(self):
        """Allocate the engine buffers on the Parameters' device (once per device) and return them."""
        dev = self._dev()
        lib = _lib.load()
        if self._eng is None or self._eng["dev"] != dev:
            N, d, L = self.node_num, self.dim, self.num_layer
            with torch.cuda.device(dev):
                q = torch.empty(lib.tpnet_q_bytes(N, d, L) // 4, dtype=torch.float32, device=dev)
                meta = torch.empty(lib.tpnet_meta_bytes(N), dtype=torch.uint8, device=dev)
                err = torch.zeros(4, dtype=torch.int32, device=dev)
            # (the per-batch ring in device memory behind the large BAR where there is one: STAGE_MODE, tpnet_stage_create_ex)
            self._eng = dict(dev=dev, dev_index=dev.index if dev.index is not None else torch.cuda.current_device(), q=q,
                             meta=meta, err=err, ws=None, stage=_Stage(lib, dev, mode=STAGE_MODE))
            if dev not in _WARMED:
                # once per process and device: the HIP runtime's first-use costs are paid here, not inside the first long
                # call -- a burst of launches behind a busy kernel, then ONE synchronise (tools/first_call.py, first 20-batch
                # call of a fresh process: 255..260 us without, 257 with the burst alone, 193 with burst + synchronise;
                # every later call 185..190)
                _WARMED.add(dev)
                with torch.cuda.device(dev):
                    _lib.check(lib.tpnet_runtime_warmup(64, 100, C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)),
                               "runtime_warmup")
                    torch.cuda.synchronize(dev)
            self._engine_valid = False
            self.__dict__["_st_cache"] = None
        return self._eng

    def _st_ref(self):
        """byref of the tpnet_state struct, rebuilt only when P[0]'s storage or the engine buffers changed."""
        ptr = self._plist()[0].data_ptr()
        c = self.__dict__.get("_st_cache")
        if c is None or c[0] != ptr or c[1] is not self._eng:
            st = self._state()
            c = (ptr, self._eng, st, C.byref(st), C.addressof(st))
            self.__dict__["_st_cache"] = c
        return c[3]

    @staticmethod
    def _host_ids(ids, what):
        """Host ids as a contiguous one-dimensional int64 numpy array (no copy when they already are: the reference's callers
        pass exactly that), or None for a torch tensor.  The range check happens in the C call that stages them."""
        if type(ids) is not np.ndarray:
            if isinstance(ids, torch.Tensor):
                return None
            ids = np.asarray(ids)
        if ids.dtype != np.int64 or not ids.flags.c_contiguous:
            ids = np.ascontiguousarray(ids, dtype=np.int64)
        if ids.ndim != 1:
            raise ValueError(f"{what} must be one-dimensional")
        return ids

    def _host_readout(self, u, v, n, flags, width, mlp_ref=None, out_gram=None):
        """One FFI call: ids checked + staged on the host, ONE kernel (readout, or readout + self.mlp when mlp_ref)."""
        eng = self._eng
        out = torch.empty((n, width), dtype=torch.float32, device=eng["dev"])
        if n:
            fast = _lib.fast()
            if fast is not None:
                self._st_ref()
                rc = fast.pair_feature(self.__dict__["_st_cache"][4], eng["stage"].handle.value, u, v, self._now_host,
                                       float(self.time_decay_weight), flags, C.addressof(mlp_ref._obj) if mlp_ref is not None else 0,
                                       out_gram.data_ptr() if out_gram is not None else 0, out.data_ptr(),
                                       _raw_stream(eng["dev_index"]))
            else:
                rc = _lib.load().tpnet_host_pair_feature(
                    self._st_ref(), eng["stage"].handle, u.ctypes.data, v.ctypes.data, n, self._now_host,
                    float(self.time_decay_weight), flags, mlp_ref, out_gram.data_ptr() if out_gram is not None else None,
                    out.data_ptr(), _raw_stream(eng["dev_index"]))
            if rc:
                _lib.check(rc, "host_pair_feature")
        return out

    def _state(self) -> _lib.State:
        eng = self._engine()
        p0 = self._plist()[0]
        if not p0.is_contiguous() or p0.dtype != torch.float32:
            raise _lib.TPNetHipError("random_projections[0] must be a contiguous float32 tensor")
        return _lib.State(p0=p0.data_ptr(), q=eng["q"].data_ptr(), meta=eng["meta"].data_ptr(), N=self.node_num,
                          d=self.dim, L=self.num_layer, err=eng["err"].data_ptr())

    def _workspace(self, max_edges: int, batch: int, stream: bool = False, tail: int = 0, keep_plan: bool = False):
        """Plan workspace.  `stream`: sized for tpnet_run_stream (the windowed schedule's plan + version log where it applies),
        capped at one chunk of the stream -- the C side walks longer streams chunk by chunk -- or, up to 64 chunks, the chunks'
        plans side by side in front of one version log (what lets a multi-chunk stream replay its plan).  `stream_log_cap_bytes`
        (None: the library's 16 GiB) bounds the version log, i.e. the chunk, for a caller short of memory."""
        eng = self._engine()
        cache = self.__dict__.setdefault("_ws_need", {})
        cap = int(self.stream_log_cap_bytes or 0) if stream else 0
        need = cache.get((max_edges, batch, stream, tail, cap))
        if need is None:
            if stream and cap:
                need = _lib.load().tpnet_stream_workspace_bytes_capped(self.node_num, self.dim, self.num_layer, max_edges, batch, cap)
            elif stream:
                need = _lib.load().tpnet_stream_workspace_bytes(self.node_num, self.dim, self.num_layer, max_edges, batch)
            else:
                need = _lib.load().tpnet_workspace_bytes(max_edges, batch)
            need = (need + 255) // 256 * 256 + tail
            if len(cache) > 64:
                cache.clear()
            cache[(max_edges, batch, stream, tail, cap)] = need
        if eng["ws"] is None or eng["ws"].numel() < need:
            try:
                eng["ws"] = torch.empty(need, dtype=torch.uint8, device=eng["dev"])
            except torch.OutOfMemoryError:
                if not stream:
                    raise
                # a GPU short of memory: halve the version log's cap (shorter chunks, more pipeline fills and drains, and beyond
                # TPNET_ARENA_MAX_RATIO no replay) until the workspace fits -- the C side takes whatever chunk it is given
                eng["ws"] = None
                lib, row = _lib.load(), 2 * self.num_layer * self.dim * 4
                cap_try = (cap or (16 << 30)) // 2
                while True:
                    need = (lib.tpnet_stream_workspace_bytes_capped(self.node_num, self.dim, self.num_layer, max_edges, batch,
                                                                    cap_try) + 255) // 256 * 256 + tail
                    try:
                        eng["ws"] = torch.empty(need, dtype=torch.uint8, device=eng["dev"])
                        break
                    except torch.OutOfMemoryError:
                        if cap_try < 8 * batch * row:
                            raise
                        cap_try //= 2
            self._drop_plan()
        if not keep_plan:
            self._drop_plan()              # (whoever asks for the workspace may write it: only run_stream keeps its plan)
        return eng["ws"]

    # plan replay (tpnet_run_stream_tagged): the table signature names the per-node (current copy, reference time) state --
    # uniform (copy 0, one reference time) right after a reset or an import, unique after anything else wrote the state
    def _table_uniform(self, tref: float):
        import struct
        self._table_sig = (struct.unpack("<Q", struct.pack("<d", float(tref)))[0] ^ 0x9E3779B97F4A7C15) | 1

    def _table_written(self):
        self._sig_counter += 1
        self._table_sig = (self._sig_counter << 1) & 0x7FFFFFFFFFFFFFFE or 2      # even: never equals a uniform signature

    def _drop_plan(self):
        """The stream workspace is about to be used by something else (or was reallocated): no plan to replay."""
        tag = self.__dict__.get("_plan_tag")
        if tag is not None:
            C.memset(C.byref(tag), 0, C.sizeof(tag))
        self.__dict__["_rows_plan_sig"] = None          # (the row-sharded runner's per-batch plan of a stream: tpnet_amd/sharded.py)

    def _layer_ptrs(self):
        arr = (C.c_void_p * self.num_layer)()
        for i in range(1, self.num_layer + 1):
            p = self._plist()[i]
            if not p.is_contiguous() or p.dtype != torch.float32 or p.device != self._dev():
                raise _lib.TPNetHipError(f"random_projections[{i}] must be a contiguous float32 tensor on the GPU")
            arr[i - 1] = p.data_ptr()
        return arr

    def _ensure_engine(self):
        """Make the engine state current: import the Parameters if somebody wrote them since the last sync."""
        if self._engine_valid and not self._params_valid and self._eng is not None \
                and self._eng["dev"] == self._plist()[0].device:
            return                              # steady state of the batch loop: the engine is the only truth
        self._engine()
        if self._engine_valid and self._params_valid and (self._param_sig != self._sig() or self._params_exposed):
            # Parameters were (or may have been) written behind our back: `.data = ...` / `p.copy_()` show in the signature,
            # in-place writes through `.data` of a list that was handed out do not -- so a hand-out alone forces the re-import
            self._engine_valid = False
        if not self._engine_valid:
            if not self._params_valid:
                raise _lib.TPNetHipError("internal error: neither the engine nor the Parameters hold the state")
            self._now_host = float(self._parameters["now_time"].item())   # the Parameters are the truth here
            st = self._state()
            _lib.check(_lib.load().tpnet_import_layers(C.byref(st), self._layer_ptrs(), self._now_host, self._stream()),
                       "import_layers")
            self._engine_valid = True
            self._param_sig = self._sig()
            self._params_exposed = False
            self._launch_id = 1
            self._table_uniform(self._now_host)

    def _materialize(self):
        """Write the eager matrices P[1..L] (decay applied) into the Parameters (in place)."""
        self._sync_now_time()
        if self._params_valid:
            return
        st = self._state()
        _lib.check(_lib.load().tpnet_export_layers(C.byref(st), self._layer_ptrs(), self._now_host,
                                                   float(self.time_decay_weight), self._stream()), "export_layers")
        self._params_valid = True
        self._param_sig = self._sig()

    def _next_launch_ids(self, n: int) -> int:
        if self._launch_id + n >= _MAX_LAUNCH_ID:       # consolidate: export + import resets every node's version
            self._materialize()
            self._engine_valid = False
            self._ensure_engine()
        first = self._launch_id
        self._launch_id += n
        return first

    def _check_ids(self, ids, what):
        if isinstance(ids, torch.Tensor) and ids.is_cuda:
            # extension: ids already on the device (e.g. from the device-side sampler) are used in place; the kernels
            # check the range themselves (bad ids are skipped, counted, and give NaN features)
            if ids.dtype != torch.int64 or ids.dim() != 1:
                raise ValueError(f"{what}: device ids must be a one-dimensional int64 tensor")
            return ids.contiguous()
        ids = np.ascontiguousarray(np.asarray(ids), dtype=np.int64)
        if ids.ndim != 1:
            raise ValueError(f"{what} must be one-dimensional")
        if ids.size and (ids.min() < -self.node_num or ids.max() >= self.node_num):
            raise IndexError(f"{what}: index out of range for {self.node_num} nodes")
        if ids.size and ids.min() < 0:
            ids = np.where(ids < 0, ids + self.node_num, ids)     # python-style negative ids, as ATen indexing
        return ids

    def _to_device(self, *arrays):
        """ONE asynchronous host->device copy for several equally long 8-byte arrays (int64 ids, float64 times)
        through a small ring of pinned staging buffers.  The reference issues one pageable (blocking) copy per array
        (TPNet.py:74-77); here the host only memcpy's into pinned memory and moves on."""
        if all(isinstance(a, torch.Tensor) for a in arrays):
            return list(arrays)                                       # already resident
        if any(isinstance(a, torch.Tensor) for a in arrays):
            arrays = [a.cpu().numpy() if isinstance(a, torch.Tensor) else a for a in arrays]
        n = arrays[0].size
        k = len(arrays)
        dev = self._dev()
        ring = self.__dict__.setdefault("_pin_ring", {"bufs": [], "pos": 0})
        if not ring["bufs"]:
            ring["bufs"] = [[None, None] for _ in range(8)]          # [pinned tensor, event of its last copy]
        slot = ring["bufs"][ring["pos"]]
        ring["pos"] = (ring["pos"] + 1) % len(ring["bufs"])
        if slot[1] is not None:
            slot[1].synchronize()                                     # the copy that last used this buffer is done
        if slot[0] is None or slot[0].numel() < k * n:
            slot[0] = torch.empty(max(k * n, 4096), dtype=torch.int64).pin_memory()
        host = slot[0][: k * n].view(k, n)
        hv = host.numpy()
        for i, a in enumerate(arrays):
            hv[i] = a.view(np.int64)
        out = host.to(dev, non_blocking=True)
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(dev))
        return [out[i] if arrays[i].dtype == np.int64 else out[i].view(torch.float64) for i in range(k)]

    def _to_device_multi(self, *arrays):
        """_to_device for host arrays of DIFFERENT lengths (8-byte items): one pinned buffer, one asynchronous copy, a device view
        per array."""
        sizes = [int(a.size) for a in arrays]
        tot = sum(sizes)
        dev = self._dev()
        ring = self.__dict__.setdefault("_pin_ring", {"bufs": [], "pos": 0})
        if not ring["bufs"]:
            ring["bufs"] = [[None, None] for _ in range(8)]
        slot = ring["bufs"][ring["pos"]]
        ring["pos"] = (ring["pos"] + 1) % len(ring["bufs"])
        if slot[1] is not None:
            slot[1].synchronize()
        if slot[0] is None or slot[0].numel() < tot:
            slot[0] = torch.empty(max(tot, 4096), dtype=torch.int64).pin_memory()
        host = slot[0][:tot]
        hv = host.numpy()
        o = 0
        for a, k in zip(arrays, sizes):
            hv[o:o + k] = a.reshape(-1).view(np.int64)
            o += k
        out = host.to(dev, non_blocking=True)
        if slot[1] is None:
            slot[1] = torch.cuda.Event()
        slot[1].record(torch.cuda.current_stream(dev))
        res, o = [], 0
        for a, k in zip(arrays, sizes):
            v = out[o:o + k]
            res.append(v if a.dtype == np.int64 else v.view(torch.float64))
            o += k
        return res

    def _ids_to_device(self, ids, what):
        return self._to_device(self._check_ids(ids, what))[0]

    # copy / pickle: the engine buffers, pinned staging ring and cached refs are per-process plumbing, not state ------
    def __getstate__(self):
        if self._eng is not None and not self._params_valid:
            self._materialize()                                    # the Parameters carry the state into the copy
        d = dict(self.__dict__)
        for k in ("_eng", "_pin_ring", "_param_refs", "_st_cache"):
            d.pop(k, None)
        d["_eng"] = None
        d["_engine_valid"] = False
        d["_params_valid"] = True
        d["_param_sig"] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__dict__.update(copy.deepcopy(self.__getstate__(), memo))
        return new

    # nn.Module hooks that read or write the Parameters wholesale ---------------------------------------------
    def state_dict(self, *args, **kwargs):
        self._materialize()
        self._params_exposed = True            # the returned tensors alias the Parameters' storage
        return super().state_dict(*args, **kwargs)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self._params_valid = True
        self._engine_valid = False

    def _apply(self, fn, *args, **kwargs):
        if self._eng is not None and not self._params_valid:
            self._materialize()
        out = super()._apply(fn, *args, **kwargs)
        self._params_valid = True
        self._engine_valid = False
        return out

    # ------------------------------------------------------------------------------------------------------
    # reference interface
    # ------------------------------------------------------------------------------------------------------
    def update(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, node_interact_times: np.ndarray):
        """models/TPNet.py:67-99: decay to t[-1], then P[i][src] += w*P[i-1][dst]; P[i][dst] += w*P[i-1][src]
        for i = L..1 on pre-batch values; now_time <- t[-1]."""
        t = np.ascontiguousarray(np.asarray(node_interact_times), dtype=np.float64)
        if t.size == 0:
            raise IndexError("update() with an empty batch (the reference indexes node_interact_times[-1])")
        if not (len(src_node_ids) == len(dst_node_ids) == len(t)):
            raise ValueError("src_node_ids, dst_node_ids and node_interact_times must have the same length")
        self._ensure_engine()
        lib = _lib.load()
        next_time = float(t[-1])
        B = int(t.size)
        lam = float(self.time_decay_weight)
        flags = 0
        src_h, dst_h = self._host_ids(src_node_ids, "src_node_ids"), self._host_ids(dst_node_ids, "dst_node_ids")
        stage = self._eng["stage"]
        host = src_h is not None and dst_h is not None and B <= stage.max_host_batch
        if host and self.exact:
            # range check BEFORE anything is enqueued (the exact mode's decay below is a state change; otherwise the C call
            # checks the ids on the host before it launches anything)
            for ids, what in ((src_h, "src_node_ids"), (dst_h, "dst_node_ids")):
                if ids.min() < -self.node_num or ids.max() >= self.node_num:
                    raise IndexError(f"{what}: index out of range for {self.node_num} nodes")
        elif not host:
            src, dst, t_dev = self._to_device(self._check_ids(src_node_ids, "src_node_ids"),
                                              self._check_ids(dst_node_ids, "dst_node_ids"), t)
        if self.exact:
            # the factor exactly as the reference forms it: f64 numpy, rounded to f32 once (TPNet.py:84-85)
            g = np.exp(-self.time_decay_weight * (np.float64(next_time) - np.float64(self._now_host)))
            fac = (C.c_float * self.num_layer)(*[np.float32(np.power(g, i)) for i in range(1, self.num_layer + 1)])
            _lib.check(lib.tpnet_decay(self._st_ref(), fac, next_time, self._stream()), "decay")
            flags |= _lib.FLAG_SEQUENTIAL
        ws = self._workspace(B, B, tail=24 * B + 512 if (host and B > stage.max_batch) else 0)
        lid = self._next_launch_ids(1)
        if host:
            # host arrays (what the reference's loop passes, train_link_prediction.py:372): one FFI call = staging + one
            # single-workgroup plan kernel + the step kernel
            fast = _lib.fast()
            if fast is not None:
                self._st_ref()
                rc = fast.update(self.__dict__["_st_cache"][4], self._eng["stage"].handle.value, src_h, dst_h, t, self._now_host,
                                 lam, lid, flags, ws.data_ptr(), ws.numel(), _raw_stream(self._eng["dev_index"]))
            else:
                rc = lib.tpnet_host_update(self._st_ref(), self._eng["stage"].handle, src_h.ctypes.data, dst_h.ctypes.data,
                                           t.ctypes.data, B, self._now_host, lam, lid, flags, ws.data_ptr(), ws.numel(),
                                           _raw_stream(self._eng["dev_index"]))
            if rc:
                _lib.check(rc, "host_update")
        else:
            _lib.check(lib.tpnet_update(self._st_ref(), src.data_ptr(), dst.data_ptr(), t_dev.data_ptr(), B, self._now_host,
                                        lam, lid, flags, ws.data_ptr(), ws.numel(), self._stream()),
                       "update")
        self._now_host = next_time
        self._params_valid = False
        self._now_dirty = True
        self._table_written()

    def get_random_projections(self, node_ids: np.ndarray):
        """models/TPNet.py:101-110: [P[i][node_ids] for i in 0..L]."""
        self._ensure_engine()
        ids = self._ids_to_device(node_ids, "node_ids")
        n = ids.numel()
        out = torch.empty((self.num_layer + 1, n, self.dim), dtype=torch.float32, device=self._dev())
        st = self._state()
        _lib.check(_lib.load().tpnet_gather_rows(C.byref(st), ids.data_ptr(), n, self._now_host,
                                                 float(self.time_decay_weight), out.data_ptr(), self._stream()),
                   "gather_rows")
        return [out[i] for i in range(self.num_layer + 1)]

    @property
    def packed_feature_dim(self) -> int:
        """Distinct entries of the symmetric (2L+2) x (2L+2) Gram: the row length of `packed` readouts."""
        nn_ = 2 * self.num_layer + 2
        return nn_ * (nn_ + 1) // 2

    def pair_gram(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray, raw: bool = False,
                  packed: bool = False) -> torch.Tensor:
        """The pairwise feature BEFORE self.mlp (models/TPNet.py:119-128): [n, (2L+2)^2] on the GPU.  `raw` leaves
        out the x<0 -> 0, log(x+1) tail whatever `not_scale` says (partial inner products of a column shard);
        `packed` (implies raw) returns only the distinct entries a <= b: [n, packed_feature_dim]."""
        self._ensure_engine()
        if len(src_node_ids) != len(dst_node_ids):
            raise ValueError("src_node_ids and dst_node_ids must have the same length")
        uh, vh = self._host_ids(src_node_ids, "src_node_ids"), self._host_ids(dst_node_ids, "dst_node_ids")
        if uh is not None and vh is not None and uh.size <= self._eng["stage"].max_pairs:
            flags = _lib.FLAG_NOT_SCALE if (self.not_scale or raw or packed) else 0
            if packed:
                flags |= _lib.FLAG_PACKED
            return self._host_readout(uh, vh, uh.size, flags,
                                      self.packed_feature_dim if packed else self.pair_wise_feature_dim)
        u, v = self._to_device(self._check_ids(src_node_ids, "src_node_ids"),
                               self._check_ids(dst_node_ids, "dst_node_ids"))
        n = u.numel()
        out = torch.empty((n, self.packed_feature_dim if packed else self.pair_wise_feature_dim),
                          dtype=torch.float32, device=self._dev())
        st = self._state()
        flags = _lib.FLAG_NOT_SCALE if (self.not_scale or raw or packed) else 0
        if packed:
            flags |= _lib.FLAG_PACKED
        _lib.check(_lib.load().tpnet_pair_gram(C.byref(st), u.data_ptr(), v.data_ptr(), n, self._now_host,
                                               float(self.time_decay_weight), flags, out.data_ptr(), self._stream()),
                   "pair_gram")
        return out

    def pair_gram_shared(self, node_ids: np.ndarray, first_ids: np.ndarray, second_ids: np.ndarray):
        """Two readouts that share their first node, whose rows are fetched once: (G(node, first), G(node, second)),
        each [n, (2L+2)^2] before self.mlp."""
        self._ensure_engine()
        if not (len(node_ids) == len(first_ids) == len(second_ids)):
            raise ValueError("node_ids, first_ids and second_ids must have the same length")
        u, v1, v2 = self._to_device(self._check_ids(node_ids, "node_ids"), self._check_ids(first_ids, "first_ids"),
                                    self._check_ids(second_ids, "second_ids"))
        n = u.numel()
        out = torch.empty((2, n, self.pair_wise_feature_dim), dtype=torch.float32, device=self._dev())
        st = self._state()
        flags = _lib.FLAG_NOT_SCALE if self.not_scale else 0
        _lib.check(_lib.load().tpnet_pair_gram_shared(C.byref(st), u.data_ptr(), v1.data_ptr(), v2.data_ptr(), n,
                                                      self._now_host, float(self.time_decay_weight), flags,
                                                      out[0].data_ptr(), out[1].data_ptr(), self._stream()),
                   "pair_gram_shared")
        return out[0], out[1]

    def get_pair_wise_feature(self, src_node_ids: np.ndarray, dst_node_ids: np.ndarray):
        """models/TPNet.py:112-129.  No gradient flows into the projections (requires_grad=False in the
        reference, :49-62); self.mlp stays a trainable torch module.
        The encoder calls this with src_node_ids = tile(neighbours, 2) (models/TPNet.py:313-316): when the two halves
        of src_node_ids are equal, each neighbour's rows are fetched once for both of its pairs."""
        if self.fused_mlp and self._plist()[0].device.type == "cuda":
            # opt-in: self.mlp on the bf16 matrix cores INSIDE the readout kernel (the features never leave the chip)
            from . import fused_mlp as fm
            # (rows of < 256 floats on long lists: the 512-thread workgroups of the one-kernel version cost the readout its
            # occupancy -- 80 000 pairs at d=128: 89 us against 69 us for readout kernel + mlp kernel -- tools/feature_rate.py)
            if fm.readout_supported(self) and (self.dim >= 256 or len(src_node_ids) <= 16384):
                if len(src_node_ids) != len(dst_node_ids):
                    raise ValueError("src_node_ids and dst_node_ids must have the same length")
                self._ensure_engine()
                u, v = self._to_device(self._check_ids(src_node_ids, "src_node_ids"),
                                       self._check_ids(dst_node_ids, "dst_node_ids"))
                return fm.fused_readout_mlp(self, u, v)
        if isinstance(src_node_ids, torch.Tensor):
            return self._apply_mlp(self.pair_gram(src_node_ids, dst_node_ids))
        src = np.asarray(src_node_ids)
        n = len(src)
        if not self.fused_mlp and not isinstance(dst_node_ids, torch.Tensor):
            # readout AND self.mlp in one launch, fp32: the decoder's call (models/modules.py:112: n = batch size) and, where the
            # fp32 matrix-core kernel serves the shape (L = 3, rows of >= 36 floats in 16-byte vectors), lists of any length
            fused = self._fused_feature(src, dst_node_ids, n)
            if fused is not None:
                return fused
        if n > 8192 and n % 2 == 0 and not self.fused_mlp and self._plist()[0].device.type == "cuda":
            # the encoder's call as the reference issues it (models/TPNet.py:311-316: src = tile(neigh, 2), dst = concat(repeat(a1, K),
            # repeat(a2, K)) on the host): recognised in one pass over the two arrays in C; n / 2 neighbour ids + 2 n / (2 K) anchors
            # go up instead of 2 n ids, the anchored readout and the dense layers run as one call
            feats = self._encoder_pattern_features(src, dst_node_ids, n)
            if feats is not None:
                return feats
        # (rows of <= 128 floats: the generic kernel's 16-lane geometry is as fast on long lists; measured)
        if self.dim > 128 and n >= 4 and n % 2 == 0 and n > 8192 and np.array_equal(src[: n // 2], src[n // 2:]):
            # the encoder's pattern: neighbours tiled twice, each half of dst a np.repeat of the row's anchor
            # (models/TPNet.py:313-316): one lane group per row with the anchors in registers (rows of <= 128 floats: the
            # generic kernel is as fast on long lists -- measured, tools/encoder_readout.py)
            dst = np.asarray(dst_node_ids)
            if self._plist()[0].device.type == "cuda":
                self._ensure_engine()
                r1 = self._anchor_runs(dst[: n // 2])
                r2 = self._anchor_runs(dst[n // 2:]) if r1 is not None else None
                if r1 is not None and r2 is not None and _lib.load().tpnet_pair_gram_anchored_supported(self._st_ref()):
                    K = int(np.gcd(r1[1], r2[1]))
                    if K >= 4:
                        m = (n // 2) // K
                        g = self.pair_gram_anchored(src[: n // 2].reshape(m, K), dst[: n // 2: K], dst[n // 2:: K])
                        return self._apply_mlp(g.view(-1, self.pair_wise_feature_dim))
        if self.dim > 128 and n >= 2 and n % 2 == 0 and np.array_equal(src[: n // 2], src[n // 2:]):
            dst = np.asarray(dst_node_ids)
            g1, g2 = self.pair_gram_shared(src[: n // 2], dst[: n // 2], dst[n // 2:])
            return self._apply_mlp(torch.cat([g1, g2], dim=0))
        return self._apply_mlp(self.pair_gram(src_node_ids, dst_node_ids))

    def _encoder_pattern_features(self, src, dst, n):
        """get_pair_wise_feature for the encoder's tile / repeat pattern on host arrays (None: not that pattern, or a shape the
        one-call path does not serve)."""
        if type(src) is not np.ndarray or type(dst) is not np.ndarray or len(dst) != n:
            return None
        if src.dtype != np.int64 or dst.dtype != np.int64 or not src.flags.c_contiguous or not dst.flags.c_contiguous \
                or src.ndim != 1 or dst.ndim != 1:
            return None
        prep = self._overlapped_mlp()
        if prep is None:
            return None
        self._ensure_engine()
        lib = _lib.load()
        if not lib.tpnet_pair_gram_anchored_supported(self._st_ref()):
            return None
        NG = self.pair_wise_feature_dim
        flags = _lib.FLAG_NOT_SCALE if self.not_scale else 0
        grad = _ff.needs_grad(prep[4])
        if grad and int(lib.tpnet_host_encoder_pattern(src.ctypes.data, dst.ctypes.data, n, self.node_num)) < 4:
            # (training: the cheap host check FIRST -- a declined call would already have allocated an n x 64 feature buffer and an
            # autograd node over it, for the general path to do the same work again)
            return None
        # ONE crossing (tpnet_host_anchored_features): pattern and range check, the neighbours + anchors staged through a pinned
        # ring of its own (4 slots of 1 MB, created at the first call of this kind; the kernel reads the slot, no copy is enqueued)
        # and the launch -- 82 us of host time per 80 000-pair call before (C check, two numpy reductions, a pinned copy and an
        # enqueued host-to-device copy from Python), against 46 us of GPU time
        if (n // 2 + 2 * (n // 8)) * 8 <= _BIG_SLOT_BYTES:
            eng = self._eng
            if eng.get("stage_big") is None:
                eng["stage_big"] = _Stage(lib, eng["dev"], slots=4, slot_bytes=_BIG_SLOT_BYTES)
            served = C.c_int32(0)

            def launch_host(gram):
                out = torch.empty((n, NG), dtype=torch.float32, device=eng["dev"])
                _lib.check(lib.tpnet_host_anchored_features(self._st_ref(), eng["stage_big"].handle, src.ctypes.data, dst.ctypes.data, n,
                                                            self._now_host, float(self.time_decay_weight), flags, prep[2],
                                                            _ptr_or_null(gram), out.data_ptr(), C.byref(served),
                                                            _raw_stream(eng["dev_index"])), "host_anchored_features")
                return out
            # (no backward pass: no feature buffer; the call declines -- served stays 0 -- where readout and dense layers are two
            # launches, and where the arrays are not the pattern or hold an id out of range: the path below then answers)
            try:
                res = _ff.apply_with_grad(self.mlp, launch_host, n, NG) if grad else launch_host(None)
            except _lib.NeedGramBuffer:
                res = launch_host(torch.empty((n, NG), dtype=torch.float32, device=eng["dev"]))
            if served.value >= 4:
                return res
        K = int(lib.tpnet_host_encoder_pattern(src.ctypes.data, dst.ctypes.data, n, self.node_num))
        if K < 4:
            return None
        h = n // 2
        m = h // K
        wd, a1, a2 = self._to_device_multi(src[:h], np.ascontiguousarray(dst[:h:K]), np.ascontiguousarray(dst[h::K]))

        def launch(gram):
            out = torch.empty((n, NG), dtype=torch.float32, device=self._eng["dev"])
            _lib.check(lib.tpnet_anchored_features(self._st_ref(), wd.data_ptr(), a1.data_ptr(), a2.data_ptr(), m, K, self._now_host,
                                                   float(self.time_decay_weight), flags, prep[2], _ptr_or_null(gram), out.data_ptr(),
                                                   _raw_stream(self._eng["dev_index"])), "anchored_features")
            return out
        if grad:
            return _ff.apply_with_grad(self.mlp, launch, n, NG)
        return self._launch_no_grad(launch, m, K, prep, n, NG)

    def pair_gram_anchored(self, neighbor_ids, first_anchor_ids, second_anchor_ids, matrix_cores=True):
        """The encoder's readout before self.mlp (models/TPNet.py:311-324): neighbor_ids [n, K] (the sampled neighbours of n
        rows), two anchors per row (the edge's src and dst).  Returns [2, n*K, (2L+2)^2]: G(neighbour, first anchor) for every
        (row, neighbour), then G(neighbour, second anchor) -- viewed as [2*n*K, .] this is the reference's
        get_pair_wise_feature(tile(neighbours, 2), concat(repeat(first, K), repeat(second, K))) pair order.  One lane group per
        row keeps both anchors' rows in registers for its K neighbours (tpnet_pair_gram_anchored); rows of 64 / 128 floats with
        L = 3 and K >= 4 take the matrix cores (csrc/encoder_mfma.hip) unless matrix_cores=False."""
        self._ensure_engine()
        if isinstance(neighbor_ids, torch.Tensor):
            if neighbor_ids.dim() != 2:
                raise ValueError("neighbor_ids must be [n, K]")
            n, K = neighbor_ids.shape
            w = neighbor_ids.reshape(-1)
        else:
            nb = np.asarray(neighbor_ids)
            if nb.ndim != 2:
                raise ValueError("neighbor_ids must be [n, K]")
            n, K = nb.shape
            w = nb.reshape(-1)
        if len(first_anchor_ids) != n or len(second_anchor_ids) != n:
            raise ValueError("one first and one second anchor per row of neighbor_ids")
        lib = _lib.load()
        if not lib.tpnet_pair_gram_anchored_supported(self._st_ref()):
            raise _lib.TPNetHipError(f"pair_gram_anchored needs dim in (64, 128, 256, 512), not {self.dim}: use pair_gram_shared")
        wd = self._to_device(self._check_ids(w, "neighbor_ids"))[0]
        a1, a2 = self._to_device(self._check_ids(first_anchor_ids, "first_anchor_ids"),
                                 self._check_ids(second_anchor_ids, "second_anchor_ids"))
        out = torch.empty((2, n * K, self.pair_wise_feature_dim), dtype=torch.float32, device=self._dev())
        flags = (_lib.FLAG_NOT_SCALE if self.not_scale else 0) | (0 if matrix_cores else _lib.FLAG_NO_MFMA_READOUT)
        _lib.check(lib.tpnet_pair_gram_anchored(self._st_ref(), wd.data_ptr(), a1.data_ptr(), a2.data_ptr(), n, K,
                                                self._now_host, float(self.time_decay_weight), flags, out[0].data_ptr(),
                                                out[1].data_ptr(), self._stream()), "pair_gram_anchored")
        return out

    def _launch_no_grad(self, launch, n_rows, K, prep, n, NG):
        """An encoder call without a backward pass: no feature buffer where readout and dense layers are one launch; if the runtime
        refuses that launch after all (TPNET_ERR_NEED_GRAM), once more with a scratch buffer."""
        try:
            return launch(self._gram_buffer(n_rows, K, prep, n, NG))
        except _lib.NeedGramBuffer:
            return launch(torch.empty((n, NG), dtype=torch.float32, device=self._eng["dev"]))

    def _gram_buffer(self, n_rows, K, prep, n, NG):
        """Where the pre-mlp features of an encoder call go when no backward pass needs them: nowhere (None) if readout and dense
        layers are ONE launch (tpnet_encoder_fused_supported), else a scratch tensor between the two launches."""
        if _lib.load().tpnet_encoder_fused_supported(self._st_ref(), n_rows, K, prep[2]):
            return None
        return torch.empty((n, NG), dtype=torch.float32, device=self._eng["dev"])

    def _overlapped_mlp(self):
        """The prepared fp32 weights of self.mlp if the encoder's one-call path applies (readout chunks and their dense layers side
        by side, tpnet_anchored_features): L = 3, the reference's Linear-ReLU-Linear on this GPU, not the opt-in bf16 layers."""
        if self.fused_mlp or self.num_layer != 3:
            return None
        prep = _ff.prepared(self.mlp, self.pair_wise_feature_dim)
        return prep if (prep is not None and prep[1].w1) else None

    def get_pair_wise_feature_anchored(self, neighbor_ids, first_anchor_ids, second_anchor_ids):
        """Extension: the encoder's call (models/TPNet.py:313-316) from its natural arguments; [2*n*K, (2L+2)^2] in the
        reference's row order, self.mlp applied."""
        prep = self._overlapped_mlp() if self._plist()[0].device.type == "cuda" else None
        if prep is not None and isinstance(neighbor_ids, torch.Tensor) and neighbor_ids.is_cuda:
            self._ensure_engine()
            lib = _lib.load()
            if lib.tpnet_pair_gram_anchored_supported(self._st_ref()):
                n, K = neighbor_ids.shape
                wd = self._check_ids(neighbor_ids.reshape(-1), "neighbor_ids")
                a1, a2 = self._to_device(self._check_ids(first_anchor_ids, "first_anchor_ids"),
                                         self._check_ids(second_anchor_ids, "second_anchor_ids"))
                NG = self.pair_wise_feature_dim
                flags = _lib.FLAG_NOT_SCALE if self.not_scale else 0

                def launch(gram):
                    out = torch.empty((2 * n * K, NG), dtype=torch.float32, device=self._eng["dev"])
                    _lib.check(lib.tpnet_anchored_features(self._st_ref(), wd.data_ptr(), a1.data_ptr(), a2.data_ptr(), n, K,
                                                           self._now_host, float(self.time_decay_weight), flags, prep[2],
                                                           _ptr_or_null(gram), out.data_ptr(), _raw_stream(self._eng["dev_index"])),
                               "anchored_features")
                    return out
                if _ff.needs_grad(prep[4]):
                    return _ff.apply_with_grad(self.mlp, launch, 2 * n * K, NG)
                return self._launch_no_grad(launch, n, K, prep, 2 * n * K, NG)
        g = self.pair_gram_anchored(neighbor_ids, first_anchor_ids, second_anchor_ids)
        return self._apply_mlp(g.view(-1, self.pair_wise_feature_dim))

    def encoder_pair_features(self, sampler, src_ids: torch.Tensor, other_ids: torch.Tensor, times: torch.Tensor,
                              num_neighbors: int):
        """Extension: the encoder's whole readout for one (src, other) batch with the ids resident on the device
        (models/TPNet.py:280-324): `sampler` = a GpuRecentNeighborSampler, src_ids / other_ids int64 [B] and times float64 [B] on
        the device, or all three as host numpy arrays (the reference's batch slices: staged, no copy enqueued).  Returns (features [4*B*K, (2L+2)^2] in the reference's row order with self.mlp applied, neighbour ids
        [2B, K] on the device).  One FFI call = row set-up + neighbour sampling + anchored readout; self.mlp behind it."""
        self._ensure_engine()
        dev = self._dev()
        host = isinstance(src_ids, np.ndarray)
        if host:
            # the batch's arrays as the reference's loop holds them (numpy slices of the edge list): staged through the pinned ring
            # and read there by the row set-up kernel -- no copy is enqueued
            src_ids = self._host_ids(src_ids, "src_ids")
            other_ids = self._host_ids(other_ids, "other_ids")
            times = np.ascontiguousarray(np.asarray(times), dtype=np.float64)
            B, K = int(src_ids.size), int(num_neighbors)
            if other_ids is None or other_ids.size != B or times.size != B:
                raise ValueError("encoder_pair_features: src_ids, other_ids and times must be host arrays of one length")
            if B > self._eng["stage"].max_host_batch:
                s_d, o_d, t_d = self._to_device(self._check_ids(src_ids, "src_ids"), self._check_ids(other_ids, "other_ids"), times)
                return self.encoder_pair_features(sampler, s_d, o_d, t_d, num_neighbors)
        else:
            B, K = int(src_ids.numel()), int(num_neighbors)
            for name, x, dt in (("src_ids", src_ids, torch.int64), ("other_ids", other_ids, torch.int64), ("times", times, torch.float64)):
                if x.device != dev or x.dtype != dt or x.dim() != 1 or x.numel() != B or not x.is_contiguous():
                    raise ValueError(f"encoder_pair_features: {name} must be a contiguous {dt} tensor of {B} elements on {dev}")
        lib = _lib.load()
        NG = self.pair_wise_feature_dim
        nbytes = lib.tpnet_encoder_scratch_bytes(B, K)
        scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        flags = _lib.FLAG_NOT_SCALE if self.not_scale else 0
        off = (-scratch.data_ptr()) % 256 + 64 * B
        neigh = scratch[off: off + 16 * B * K].view(torch.int64).view(2 * B, K)
        prep = self._overlapped_mlp()
        if prep is not None and lib.tpnet_pair_gram_anchored_supported(self._st_ref()):
            # ONE call, self.mlp included: the dense layers of a chunk of rows run beside the readout of the next one
            def launch(gram):
                out = torch.empty((4 * B * K, NG), dtype=torch.float32, device=dev)
                if host:
                    _lib.check(lib.tpnet_host_encoder_features(
                        self._st_ref(), self._eng["stage"].handle, sampler._buf.data_ptr(), sampler.E, sampler.num_nodes,
                        src_ids.ctypes.data, other_ids.ctypes.data, times.ctypes.data, B, K, self._now_host,
                        float(self.time_decay_weight), flags, prep[2], scratch.data_ptr(), nbytes, _ptr_or_null(gram), out.data_ptr(),
                        _raw_stream(self._eng["dev_index"])), "host_encoder_features")
                    return out
                _lib.check(lib.tpnet_encoder_features(self._st_ref(), sampler._buf.data_ptr(), sampler.E, sampler.num_nodes,
                                                      src_ids.data_ptr(), other_ids.data_ptr(), times.data_ptr(), B, K,
                                                      self._now_host, float(self.time_decay_weight), flags, prep[2],
                                                      scratch.data_ptr(), nbytes, _ptr_or_null(gram), out.data_ptr(),
                                                      _raw_stream(self._eng["dev_index"])), "encoder_features")
                return out
            if _ff.needs_grad(prep[4]):
                return _ff.apply_with_grad(self.mlp, launch, 4 * B * K, NG), neigh
            return self._launch_no_grad(launch, 2 * B, K, prep, 4 * B * K, NG), neigh
        if host:
            s_d, o_d, t_d = self._to_device(self._check_ids(src_ids, "src_ids"), self._check_ids(other_ids, "other_ids"), times)
            return self.encoder_pair_features(sampler, s_d, o_d, t_d, num_neighbors)
        out = torch.empty((2, 2 * B * K, NG), dtype=torch.float32, device=dev)
        _lib.check(lib.tpnet_encoder_gram(self._st_ref(), sampler._buf.data_ptr(), sampler.E, sampler.num_nodes, src_ids.data_ptr(),
                                          other_ids.data_ptr(), times.data_ptr(), B, K, self._now_host,
                                          float(self.time_decay_weight), flags, scratch.data_ptr(), nbytes, out.data_ptr(),
                                          _raw_stream(self._eng["dev_index"])), "encoder_gram")
        return self._apply_mlp(out.view(-1, NG)), neigh

    @staticmethod
    def _anchor_runs(half):
        """If `half` is repeat(anchors, K) for some K >= 2 (np.repeat of the encoder's call): (anchors, K); else None.
        Any K whose blocks are constant serves: the gcd of the positions where the value changes and of the length."""
        m = half.size
        if m < 2:
            return None
        ch = np.flatnonzero(half[1:] != half[:-1]) + 1
        K = int(np.gcd.reduce(ch, initial=m)) if ch.size else m
        if K < 2:
            return None
        return half[::K], K

    def get_pair_wise_feature_shared(self, node_ids, first_ids, second_ids):
        """Extension: get_pair_wise_feature(tile(node_ids, 2), concat(first_ids, second_ids)) -- the encoder's pattern
        (models/TPNet.py:313-316) -- with each node's rows fetched once.  Returns [2n, (2L+2)^2] in the reference's
        row order (all (node, first) pairs, then all (node, second) pairs)."""
        g1, g2 = self.pair_gram_shared(node_ids, first_ids, second_ids)
        return self._apply_mlp(torch.cat([g1, g2], dim=0))

    def _fused_feature(self, src, dst, n):
        """tpnet_host_pair_feature with self.mlp (None if self.mlp is not the reference's Linear-ReLU-Linear on this GPU)."""
        NG = self.pair_wise_feature_dim
        mlp = self._modules.get("mlp")              # (self.mlp without nn.Module.__getattr__)
        if mlp is None:
            mlp = self.mlp
        if len(dst) != n:
            raise ValueError("src_node_ids and dst_node_ids must have the same length")
        if self._plist()[0].device.type != "cuda":
            return None
        prep = _ff.prepared(mlp, NG)
        if prep is None:
            return None
        self._ensure_engine()
        mfma = bool(prep[1].w1) and self.dim % 4 == 0 and self.dim >= 36 and not self.use_matrix
        if n > (self._eng["stage"].max_pairs if mfma else _ff.MAX_PAIRS):
            # a long list: with the matrix-core kernel still one launch, from a device copy of the ids -- except the encoder's
            # pattern on wide rows, which the caller below serves with the anchored readout + the matrix-core mlp
            # (rows of <= 128 floats on long lists: the 512-thread workgroups of the one-kernel version cost the readout its
            # occupancy -- 80 000 pairs at d=128: 87 us against 42 + 34 us for readout kernel + dense-layer kernel)
            if not mfma or self.dim <= 128 or (n % 2 == 0 and np.array_equal(src[: n // 2], src[n // 2:])):
                return None
            u, v = self._to_device(self._check_ids(src, "src_node_ids"), self._check_ids(dst, "dst_node_ids"))
            flags = _lib.FLAG_NOT_SCALE if self.not_scale else 0

            def launch(gram):
                out = torch.empty((n, NG), dtype=torch.float32, device=self._eng["dev"])
                _lib.check(_lib.load().tpnet_pair_feature(self._st_ref(), u.data_ptr(), v.data_ptr(), n, self._now_host,
                                                          float(self.time_decay_weight), flags, prep[2],
                                                          gram.data_ptr() if gram is not None else None, out.data_ptr(),
                                                          self._stream()), "pair_feature")
                return out
            return _ff.apply_with_grad(mlp, launch, n, NG) if _ff.needs_grad(prep[4]) else launch(None)
        uh, vh = self._host_ids(src, "src_node_ids"), self._host_ids(dst, "dst_node_ids")
        flags = _lib.FLAG_NOT_SCALE if self.not_scale else 0
        if _ff.needs_grad(prep[4]):
            return _ff.apply_with_grad(mlp, lambda gram: self._host_readout(uh, vh, n, flags, NG, prep[2], gram), n, NG)
        return self._host_readout(uh, vh, n, flags, NG, prep[2])

    def _apply_mlp(self, feats: torch.Tensor) -> torch.Tensor:
        if self.fused_mlp:
            from . import fused_mlp as fm
            if fm.supported(self.mlp):
                return fm.fused_mlp(self.mlp, feats)
        y = _ff.mlp_f32(self.mlp, feats) if self.num_layer == 3 else None      # the fp32 matrix-core kernel where it applies
        return y if y is not None else self.mlp(feats)

    def reset_random_projections(self):
        """models/TPNet.py:131-139."""
        for i in range(1, self.num_layer + 1):
            nn.init.zeros_(self._plist()[i])
        self._parameters["now_time"].data = self._parameters["begging_time"].clone()
        self._now_dirty = False
        if not self.use_matrix:
            nn.init.normal_(self._plist()[0], mean=0, std=1 / math.sqrt(self.dim))
        self._now_host = float(self._parameters["begging_time"].item())
        self._params_valid = True
        if self._plist()[0].device.type == "cuda":
            st = self._state()
            _lib.check(_lib.load().tpnet_state_init(C.byref(st), self._now_host, self._stream()), "state_init")
            self._engine_valid = True
            self._param_sig = self._sig()
            self._launch_id = 1
            self._table_uniform(self._now_host)
        else:
            self._engine_valid = False

    def backup_random_projections(self):
        """models/TPNet.py:141-147: (now_time.clone(), [P[1..L] clones])."""
        self._materialize()
        return self._parameters["now_time"].clone(), [self._plist()[i].clone() for i in range(1, self.num_layer + 1)]

    def reload_random_projections(self, random_projections):
        """models/TPNet.py:149-157."""
        now_time, layers = random_projections
        self._parameters["now_time"].data = now_time.clone()
        self._now_dirty = False
        for i in range(1, self.num_layer + 1):
            self._plist()[i].data = layers[i - 1].clone()
        self._params_valid = True
        self._engine_valid = False

    # ------------------------------------------------------------------------------------------------------
    # extension: device-resident edge stream (the reference's batch loop, train_link_prediction.py:253-373)
    # ------------------------------------------------------------------------------------------------------
    default_schedule = "auto"      # run_stream: "auto" | "windowed" | "batch" (see include/tpnet_hip.h, TPNET_FLAG_SCHED_*)
    plan_replay = True             # run_stream: replay the plan of a stream that is run again on the same table state
    last_stream_replayed = False

    def reserve_stream(self, max_edges: int, batch_size: int):
        """Extension: size the stream workspace for run_stream calls of up to `max_edges` edges in batches of `batch_size` now, so
        that the first call of that size neither asks the library for the size nor grows the buffer."""
        self._ensure_engine()
        self._workspace(int(max_edges), int(batch_size), stream=True)

    def run_stream(self, src: torch.Tensor, dst: torch.Tensor, neg, t: torch.Tensor, batch_size: int,
                   want_pos: bool = True, want_neg: bool = True, out_pos=None, out_neg=None, t_end: float = None,
                   raw: bool = False, packed: bool = False, schedule: str = None, replay: bool = None):
        """For each chronological batch: pre-mlp pairwise features of (src,dst) and (src,neg) on the pre-batch
        state, then update().  src/dst/neg: int64 [E] on the GPU, t: float64 [E] on the GPU.  Returns
        (feat_pos, feat_neg) of shape [E, (2L+2)^2] (None where not requested).  `t_end` = t[-1] if the caller
        already has it on the host (avoids one 8-byte device->host copy).  `raw` / `packed`: features without the
        x<0 -> 0, log(x+1) tail / only their distinct entries, rows of packed_feature_dim (see pair_gram).
        = prepare_stream(...)(): an epoch loop that runs the same stream again holds the prepared call instead."""
        return self._run_prepared(self.prepare_stream(src, dst, neg, t, batch_size, want_pos, want_neg, out_pos, out_neg, t_end, raw,
                                                      packed, schedule, replay))

    def prepare_stream(self, src: torch.Tensor, dst: torch.Tensor, neg, t: torch.Tensor, batch_size: int,
                       want_pos: bool = True, want_neg: bool = True, out_pos=None, out_neg=None, t_end: float = None,
                       raw: bool = False, packed: bool = False, schedule: str = None, replay: bool = None) -> "PreparedStream":
        """run_stream's arguments checked ONCE (devices, dtypes, shapes, the schedule's name), its outputs allocated: the returned
        PreparedStream runs the stream every time it is called -- `call()` = run_stream(...) without the ~10 us of argument
        checking per call that a 20-batch stream of ~130 us notices.  What `train_link_prediction.py:234-253` does every epoch (the
        same chronological arrays from a reset table) is one prepared call per split, called once per epoch (a new `neg` every epoch:
        write it into the tensor the call holds -- `call.neg.copy_(...)` -- or prepare again).  The call holds references to its
        tensors; their CONTENTS are read when it runs (and torch's version counters decide whether the plan of an earlier run may be
        replayed, exactly as in run_stream); replacing a tensor's storage (`resize_`, `set_`) behind a prepared call is not supported."""
        self._ensure_engine()
        dev = self._dev()
        E = int(src.numel())
        for name, x, dt in (("src", src, torch.int64), ("dst", dst, torch.int64), ("t", t, torch.float64)):
            if x.device != dev or x.dtype != dt or not x.is_contiguous() or x.numel() != E:
                raise ValueError(f"run_stream: {name} must be a contiguous {dt} tensor of {E} elements on {dev}")
        if neg is not None and (neg.device != dev or neg.dtype != torch.int64 or neg.numel() != E or not neg.is_contiguous()):
            raise ValueError("run_stream: neg must be a contiguous int64 tensor of the same length on the same device")
        want_neg = want_neg and neg is not None
        NG = self.packed_feature_dim if packed else self.pair_wise_feature_dim
        for name, o, want in (("out_pos", out_pos, want_pos), ("out_neg", out_neg, want_neg)):
            if want and o is not None and (o.dtype != torch.float32 or o.device != dev or not o.is_contiguous()
                                           or tuple(o.shape) != (E, NG)):
                raise ValueError(f"run_stream: {name} must be a contiguous float32 tensor of shape ({E}, {NG}) on {dev}")
        if want_pos and out_pos is None:
            out_pos = torch.empty((E, NG), dtype=torch.float32, device=dev)
        if want_neg and out_neg is None:
            out_neg = torch.empty((E, NG), dtype=torch.float32, device=dev)
        batch_size = int(batch_size)
        if batch_size < 1:
            raise ValueError("run_stream: batch_size must be positive")
        flags = (_lib.FLAG_NOT_SCALE if (self.not_scale or raw or packed) else 0)
        if packed:
            flags |= _lib.FLAG_PACKED
        if self.exact:
            flags |= _lib.FLAG_EAGER_DECAY | _lib.FLAG_SEQUENTIAL
        schedule = schedule or self.default_schedule
        if schedule not in ("auto", "windowed", "batch", "windowed-sorted", "windowed-hashed"):
            raise ValueError("schedule must be 'auto', 'windowed', 'windowed-sorted', 'windowed-hashed' or 'batch'")
        flags |= {"auto": 0, "windowed": _lib.FLAG_SCHED_WINDOWED, "batch": _lib.FLAG_SCHED_BATCH,
                  "windowed-sorted": _lib.FLAG_SCHED_WINDOWED | _lib.FLAG_PLAN_SORTED,
                  "windowed-hashed": _lib.FLAG_SCHED_WINDOWED | _lib.FLAG_PLAN_HASHED}[schedule]
        p = PreparedStream()
        p.rp, p.src, p.dst, p.neg, p.t = self, src, dst, neg, t
        p.E, p.batch_size, p.nb = E, batch_size, (E + batch_size - 1) // batch_size
        p.want_pos, p.want_neg = bool(want_pos), bool(want_neg)
        p.out_pos, p.out_neg = (out_pos if want_pos else None), (out_neg if want_neg else None)
        p.t_end = None if t_end is None else float(t_end)
        p.flags, p.replay, p.exact, p.dev = flags, replay, self.exact, dev
        p.ptrs = (src.data_ptr(), dst.data_ptr(), neg.data_ptr() if neg is not None else 0, t.data_ptr(),
                  out_pos.data_ptr() if want_pos else 0, out_neg.data_ptr() if want_neg else 0)
        p.ws_cap = self.stream_log_cap_bytes
        need = 0
        if E > 0:
            w0 = self._workspace(E, batch_size, stream=True, keep_plan=True)
            need = self.__dict__["_ws_need"].get((E, batch_size, True, 0, int(self.stream_log_cap_bytes or 0)), w0.numel())
        p.ws_need = need
        return p

    def _run_prepared(self, p: "PreparedStream"):
        if p.E == 0:
            return p.out_pos, p.out_neg
        self._ensure_engine()
        if p.exact != self.exact or p.dev != self._dev():
            raise RuntimeError("this prepared stream was made for another mode / device of the module: prepare it again")
        E, batch_size = p.E, p.batch_size
        src, dst, t = p.src, p.dst, p.t
        ws = self._eng["ws"]                             # (the stream's workspace: sized when the call was prepared; looked up again
        if ws is None or ws.numel() < p.ws_need or p.ws_cap != self.stream_log_cap_bytes:   #  only if somebody replaced it since)
            ws = self._workspace(E, batch_size, stream=True, keep_plan=True)
        self._st_ref()                                   # (the cached tpnet_state struct: rebuilt only when a buffer moved)
        st = self.__dict__["_st_cache"][2]
        lid = self._next_launch_ids(p.nb)
        flags = p.flags
        t_out = C.c_double(0.0)
        # a stream that is run again on the same table state (every epoch of train_link_prediction.py:234-253: reset, then the
        # same chronological batches) replays its plan: the tag tells the C side that src / dst / t hold what they held when
        # the plan in the workspace was built (torch bumps a tensor's _version on every in-place write; a write through a raw
        # pointer or .data is not seen -- pass replay=False then) and names the table's per-node state
        tag = None
        ps, pd, pn, pt, pop, pon = p.ptrs
        if p.replay is not False and self.plan_replay:
            tag = self.__dict__.get("_plan_tag")
            if tag is None:
                tag = self.__dict__["_plan_tag"] = _lib.PlanTag()
            tag.table_sig = self._table_sig
            tag.stream_sig = (hash((ps, src._version, pd, dst._version, pt, t._version, E)) & 0xFFFFFFFFFFFFFFFF) | 1
        else:
            self._drop_plan()
        t_end = p.t_end
        fast = _lib.fast()
        if fast is not None:
            rc, t_got = fast.run_stream(C.addressof(st), ps, pd, pn, pt, E, batch_size, self._now_host, float(self.time_decay_weight),
                                        lid, flags, pop, pon, ws.data_ptr(), ws.numel(), 0 if t_end is not None else 1,
                                        _raw_stream(self._eng["dev_index"]), C.addressof(tag) if tag is not None else 0)
            t_out.value = t_got
            _lib.check(rc, "run_stream")
        else:
            _lib.check(_lib.load().tpnet_run_stream_tagged(
                C.byref(st), ps, pd, pn or None, pt, E, batch_size, self._now_host, float(self.time_decay_weight), lid, flags,
                pop or None, pon or None, ws.data_ptr(), ws.numel(), None if t_end is not None else C.byref(t_out), self._stream(),
                C.byref(tag) if tag is not None else None), "run_stream")
        self.last_stream_replayed = bool(tag is not None and tag.replayed)
        self._now_host = t_end if t_end is not None else float(t_out.value)
        self._params_valid = False
        self._now_dirty = True
        self._table_written()
        return p.out_pos, p.out_neg

    def check_device_errors(self):
        """Raise IndexError if a kernel met a node id outside [0, node_num) since the last check."""
        st = self._state()
        _lib.check(_lib.load().tpnet_check_errors(C.byref(st), self._stream()), "check_errors")
