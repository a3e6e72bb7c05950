"""ctypes binding of the C ABI in include/tpnet_hip.h (tpnet_amd/libtpnet_hip.so).

There is NO CPU fallback: if the HIP library is missing, or no GPU is present when a compute entry point is
called, the product path raises.  PyTorch is used only as plumbing (device memory, streams).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TPNET_DEV_LIB") or os.path.join(_HERE, "libtpnet_hip.so")   # TPNET_DEV_LIB: diagnostic builds

TPNET_MAX_LAYERS = 4
FLAG_NOT_SCALE = 1
FLAG_EAGER_DECAY = 2
FLAG_SEQUENTIAL = 4
FLAG_PACKED = 8
FLAG_SCHED_WINDOWED = 16
FLAG_SCHED_BATCH = 32
FLAG_PLAN_SORTED = 64
FLAG_PLAN_HASHED = 128
FLAG_NO_MFMA_READOUT = 256

ERR_INDEX = -4
ERR_NEED_GRAM = -6


class NodeMeta(C.Structure):
    _fields_ = [("ver", C.c_uint32), ("pad0", C.c_uint32), ("tref", C.c_double * 2), ("pad1", C.c_uint64)]


class State(C.Structure):
    _fields_ = [("p0", C.c_void_p), ("q", C.c_void_p), ("meta", C.c_void_p), ("N", C.c_int64), ("d", C.c_int32),
                ("L", C.c_int32), ("err", C.c_void_p)]


class PlanTag(C.Structure):
    """tpnet_plan_tag: what lets tpnet_run_stream_tagged replay the plan a stream left in the workspace (include/tpnet_hip.h)."""
    _fields_ = [("table_sig", C.c_uint64), ("stream_sig", C.c_uint64), ("replayed", C.c_uint64), ("built", C.c_uint64 * 20)]


class Mlp(C.Structure):
    """tpnet_mlp: device arrays of self.mlp in the fused kernel's layout (include/tpnet_hip.h)."""
    _fields_ = [("w1t", C.c_void_p), ("b1", C.c_void_p), ("w2t", C.c_void_p), ("b2", C.c_void_p), ("F", C.c_int32),
                ("H", C.c_int32), ("w1", C.c_void_p), ("w2f", C.c_void_p), ("wimg", C.c_void_p)]


# name -> (restype, argtypes); must list every symbol include/tpnet_hip.h declares (tests check this)
_P = C.c_void_p
_SP = C.POINTER(State)
SIGNATURES = {
    "tpnet_strerror": (C.c_char_p, [C.c_int]),
    "tpnet_abi_version": (C.c_int, []),
    "tpnet_last_hip_error": (C.c_int, []),
    "tpnet_device_count": (C.c_int, []),
    "tpnet_runtime_warmup": (C.c_int, [C.c_int32, C.c_int32, _P]),
    "tpnet_q_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "tpnet_meta_bytes": (C.c_size_t, [C.c_int64]),
    "tpnet_state_init": (C.c_int, [_SP, C.c_double, _P]),
    "tpnet_import_layers": (C.c_int, [_SP, C.POINTER(_P), C.c_double, _P]),
    "tpnet_export_layers": (C.c_int, [_SP, C.POINTER(_P), C.c_double, C.c_double, _P]),
    "tpnet_decay": (C.c_int, [_SP, C.POINTER(C.c_float), C.c_double, _P]),
    "tpnet_gather_rows": (C.c_int, [_SP, _P, C.c_int64, C.c_double, C.c_double, _P, _P]),
    "tpnet_pair_gram": (C.c_int, [_SP, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_uint32, _P, _P]),
    "tpnet_pair_gram_shared": (C.c_int, [_SP, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_uint32, _P, _P, _P]),
    "tpnet_pair_gram_anchored": (C.c_int, [_SP, _P, _P, _P, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_uint32, _P, _P,
                                           _P]),
    "tpnet_pair_gram_anchored_supported": (C.c_int, [_SP]),
    "tpnet_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "tpnet_stream_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64]),
    "tpnet_stream_workspace_bytes_capped": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_size_t]),
    "tpnet_stream_schedule": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_uint32, C.c_size_t]),
    "tpnet_update": (C.c_int, [_SP, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_uint32, C.c_uint32, _P,
                               C.c_size_t, _P]),
    "tpnet_run_stream": (C.c_int, [_SP, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_uint32,
                                   C.c_uint32, _P, _P, _P, C.c_size_t, C.POINTER(C.c_double), _P]),
    "tpnet_run_stream_tagged": (C.c_int, [_SP, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_uint32,
                                          C.c_uint32, _P, _P, _P, C.c_size_t, C.POINTER(C.c_double), _P, C.POINTER(PlanTag)]),
    "tpnet_plan_stream": (C.c_int, [_SP, _P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_uint32, _P,
                                    C.c_size_t, _P]),
    "tpnet_step_batch": (C.c_int, [_SP, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_uint32,
                                   C.c_uint32, C.c_int32, C.c_int32, _P, _P, _P, C.c_size_t, _P]),
    "tpnet_pack_rows": (C.c_int, [_SP, _P, C.c_int64, C.c_double, C.c_double, _P, _P]),
    "tpnet_unpack_rows": (C.c_int, [_SP, _P, C.c_int64, C.c_double, _P, _P]),
    "tpnet_pack_bundles": (C.c_int, [_SP, _P, C.c_int64, C.c_double, C.c_double, _P, _P]),
    "tpnet_unpack_bundles": (C.c_int, [_SP, _P, C.c_int64, C.c_double, _P, C.c_int64, _P, C.c_int32, _P]),
    "tpnet_unpack_gathered": (C.c_int, [_SP, _P, C.c_int64, C.c_double, _P, C.c_int64, _P, C.c_int32, C.c_int32, _P]),
    "tpnet_rccl_unique_id": (C.c_int, [C.c_char_p, _P]),
    "tpnet_rccl_comm_create": (C.c_int, [C.c_char_p, _P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    "tpnet_rccl_comm_destroy": (C.c_int, [_P]),
    "tpnet_rows_step": (C.c_int, [_SP, _P, _P, C.c_int64, _P, _P, C.c_int64, _P, C.c_int64, _P, C.c_int32, C.c_double, _P, _P,
                                  _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_uint32, C.c_uint32, C.c_int32, _P,
                                  _P, _P, C.c_size_t, _P]),
    "tpnet_pack_split": (C.c_int, [_SP, _P, C.c_int64, C.c_double, C.c_double, _P, _P, C.c_int64, C.c_int64, _P]),
    "tpnet_rows_step_targeted": (C.c_int, [_SP, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_double, _P, _P, _P, _P,
                                           C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_uint32, C.c_uint32, C.c_int32, _P,
                                           _P, _P, C.c_size_t, _P]),
    "tpnet_rows_stream_targeted": (C.c_int, [_SP, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_double, _P, _P, _P, _P, _P,
                                             C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_uint32, C.c_uint32,
                                             C.c_int32, _P, _P, _P, C.c_size_t, _P]),
    "tpnet_sampler_bytes": (C.c_size_t, [C.c_int64, C.c_int64]),
    "tpnet_sampler_build": (C.c_int, [_P, C.c_size_t, _P, _P, _P, _P, C.c_int64, C.c_int64, _P]),
    "tpnet_sample_recent": (C.c_int, [_P, C.c_int64, C.c_int64, _P, _P, C.c_int64, C.c_int32, _P, _P, _P, _P]),
    "tpnet_encoder_scratch_bytes": (C.c_size_t, [C.c_int64, C.c_int32]),
    "tpnet_encoder_gram": (C.c_int, [_SP, _P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int64, C.c_int32, C.c_double, C.c_double,
                                     C.c_uint32, _P, C.c_size_t, _P, _P]),
    "tpnet_encoder_rows": (C.c_int, [_SP, _P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int64, C.c_int32, _P, C.c_size_t, _P]),
    "tpnet_anchored_features": (C.c_int, [_SP, _P, _P, _P, C.c_int64, C.c_int32, C.c_double, C.c_double, C.c_uint32, C.POINTER(Mlp),
                                          _P, _P, _P]),
    "tpnet_encoder_features": (C.c_int, [_SP, _P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int64, C.c_int32, C.c_double, C.c_double,
                                         C.c_uint32, C.POINTER(Mlp), _P, C.c_size_t, _P, _P, _P]),
    "tpnet_mlp64_bf16": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "tpnet_mlp64_bwd_partial_floats": (C.c_int64, []),
    "tpnet_mlp64_bwd_bf16": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, C.c_int32, _P]),
    "tpnet_mlp64_bwd_f32": (C.c_int, [_P, _P, C.c_int64, C.POINTER(Mlp), _P, C.c_int32, _P]),
    "tpnet_pair_feature_bf16": (C.c_int, [_SP, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_uint32, _P, _P, _P, _P, _P, _P,
                                          _P]),
    "tpnet_gather_elems": (C.c_int, [_SP, _P, _P, C.c_int64, C.c_double, C.c_double, _P, _P]),
    "tpnet_decoder_bf16": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int64, _P, _P, _P, C.c_float, C.c_int32, _P, _P]),
    "tpnet_gram_finish": (C.c_int, [_P, C.c_int64, _P]),
    "tpnet_gram_unpack": (C.c_int, [_P, C.c_int64, C.c_int32, C.c_uint32, _P, _P]),
    "tpnet_check_errors": (C.c_int, [_SP, _P]),
    "tpnet_pair_feature": (C.c_int, [_SP, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_uint32, C.POINTER(Mlp), _P, _P, _P]),
    "tpnet_mlp64_f32": (C.c_int, [_P, C.c_int64, C.POINTER(Mlp), _P, _P]),
    "tpnet_encoder_fused_supported": (C.c_int, [_SP, C.c_int64, C.c_int32, C.POINTER(Mlp)]),
    "tpnet_mlp_image_bytes": (C.c_size_t, []),
    "tpnet_mlp_prepare_image": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "tpnet_mlp_prepare": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P]),
    "tpnet_stage_create": (C.c_int, [C.c_int32, C.c_size_t, C.POINTER(_P)]),
    "tpnet_stage_create_ex": (C.c_int, [C.c_int32, C.c_size_t, C.c_int32, C.POINTER(_P)]),
    "tpnet_stage_in_device_memory": (C.c_int, [_P]),
    "tpnet_stage_destroy": (C.c_int, [_P]),
    "tpnet_stage_max_pairs": (C.c_int64, [_P]),
    "tpnet_stage_max_batch": (C.c_int64, [_P]),
    "tpnet_host_pair_feature": (C.c_int, [_SP, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_uint32, C.POINTER(Mlp),
                                          _P, _P, _P]),
    "tpnet_host_update": (C.c_int, [_SP, _P, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_uint32, C.c_uint32, _P,
                                    C.c_size_t, _P]),
    "tpnet_host_encoder_pattern": (C.c_int64, [_P, _P, C.c_int64, C.c_int64]),
    "tpnet_host_anchored_features": (C.c_int, [_SP, _P, _P, _P, C.c_int64, C.c_double, C.c_double, C.c_uint32, C.POINTER(Mlp), _P, _P,
                                              C.POINTER(C.c_int32), _P]),
    "tpnet_xplan_capacity": (C.c_int64, []),
    "tpnet_xplan_targeted": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P,
                                       _P, _P, _P]),
    "tpnet_xplan_large_bytes": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32]),
    "tpnet_xplan_targeted_large": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_size_t,
                                             _P, _P, _P, _P, _P, _P, _P]),
    "tpnet_wshard_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int64, C.c_int32, C.c_int32]),
    "tpnet_wshard_plan": (C.c_int, [_SP, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                    C.c_double, C.c_uint32, C.c_int32, C.c_int32, _P, C.c_size_t, _P, C.POINTER(_P)]),
    "tpnet_wshard_info": (C.c_int, [_P, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_int64)), C.POINTER(C.POINTER(C.c_int64))]),
    "tpnet_wshard_set_buffers": (C.c_int, [_P, _P, _P, _P, _P]),
    "tpnet_wshard_begin": (C.c_int, [_P, _P, C.c_uint32, _P]),
    "tpnet_wshard_step": (C.c_int, [_P, _P, C.c_int64, C.c_uint32, _P, _P, _P]),
    "tpnet_wshard_finish": (C.c_int, [_P, C.c_uint32, _P]),
    "tpnet_wshard_run": (C.c_int, [_P, _P, _P, _P, C.c_uint32, _P]),
    "tpnet_wshard_destroy": (None, [_P]),
    "tpnet_host_encoder_features": (C.c_int, [_SP, _P, _P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int64, C.c_int32, C.c_double,
                                              C.c_double, C.c_uint32, C.POINTER(Mlp), _P, C.c_size_t, _P, _P, _P]),
}

# measurement aids (tpnet_amd/csrc/tpnet_dev.h): exported by the library, not part of the drop-in boundary
DEV_SIGNATURES = {
    "tpnet_time_stream": (C.c_int, [_SP, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_uint32,
                                    C.c_uint32, _P, _P, _P, C.c_size_t, C.c_int, C.POINTER(C.c_float),
                                    C.POINTER(C.c_float), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P]),
    "tpnet_time_wshard_run": (C.c_int, [_P, _P, _P, _P, C.c_uint32, _P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tpnet_time_rows_stream_targeted": (C.c_int, [_SP, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_double, _P, _P, _P, _P, _P,
                                                  C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_uint32, C.c_uint32,
                                                  C.c_int32, _P, _P, _P, C.c_size_t, _P, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                                  C.POINTER(C.c_float)]),
}

_lib = None


class TPNetHipError(RuntimeError):
    pass


class NeedGramBuffer(TPNetHipError):
    """TPNET_ERR_NEED_GRAM: an encoder call without a feature buffer met a runtime that refuses the one-launch kernel -- the caller
    retries once with a scratch buffer (readout and dense layers as two launches)."""


def load():
    """Load libtpnet_hip.so (built by __graft_entry__.build() / tpnet_amd/csrc/Makefile).  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TPNetHipError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). tpnet_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in list(SIGNATURES.items()) + list(DEV_SIGNATURES.items()):
        try:
            fn = getattr(lib, name)
        except AttributeError:
            if os.environ.get("TPNET_DEV_LIB"):      # an older development build loaded on purpose (A/B runs): tolerate
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


_fast = False


def fast():
    """The CPython extension in front of the per-batch entry points (tpnet_amd/_tpnet_fast.so, csrc/pyext.c), or None when
    it was not built or a diagnostic library is loaded through TPNET_DEV_LIB (the ctypes binding above then serves every call)."""
    global _fast
    if _fast is False:
        _fast = None
        if not os.environ.get("TPNET_DEV_LIB") and os.path.exists(os.path.join(_HERE, "_tpnet_fast.so")):
            load()
            from . import _tpnet_fast as m          # (an import error here is a broken build: let it show)
            if m.abi_version() != load().tpnet_abi_version():
                raise TPNetHipError("_tpnet_fast.so and libtpnet_hip.so are different builds: run __graft_entry__.build()")
            _fast = m
    return _fast


def check(rc: int, what: str = ""):
    if rc == 0:
        return
    lib = load()
    msg = lib.tpnet_strerror(rc).decode()
    if rc == ERR_INDEX:
        raise IndexError(f"tpnet_hip {what}: {msg}")
    if rc == ERR_NEED_GRAM:
        raise NeedGramBuffer(f"tpnet_hip {what}: {msg}")
    extra = f" (hipError_t {lib.tpnet_last_hip_error()})" if rc == -3 else ""
    raise TPNetHipError(f"tpnet_hip {what}: {msg}{extra}")
