"""CPU tests of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/tpnet_hip.h declares; the ctypes binding lists exactly those; the Python module mirrors the reference
operator's surface.  No compute call is made here (no GPU in this tier)."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tpnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tpnet_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_exported_and_bound(hip_lib):
    from tpnet_amd import _lib
    syms = _declared_symbols()
    assert len(syms) >= 15
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/tpnet_hip.h but not exported"
    assert sorted(_lib.SIGNATURES.keys()) == syms
    assert hip_lib.tpnet_abi_version() == 7
    assert hip_lib.tpnet_strerror(-4) == b"node id out of range"


def test_struct_layouts_match_header():
    from tpnet_amd import _lib
    assert ctypes.sizeof(_lib.NodeMeta) == 32
    assert _lib.State.p0.offset == 0 and _lib.State.q.offset == 8 and _lib.State.meta.offset == 16
    assert _lib.State.N.offset == 24 and _lib.State.d.offset == 32 and _lib.State.L.offset == 36
    assert _lib.State.err.offset == 40 and ctypes.sizeof(_lib.State) == 48


def test_size_helpers(hip_lib):
    assert hip_lib.tpnet_q_bytes(9228, 128, 3) == 2 * 9228 * 3 * 128 * 4
    assert hip_lib.tpnet_meta_bytes(9228) == 9228 * 32


def test_stream_workspace_sizes_for_one_and_for_several_chunks(hip_lib):
    """tpnet_stream_workspace_bytes(_capped): a stream that fits one chunk gets that chunk's plan + version log; a longer one (here:
    the log capped) the plans of its chunks side by side + ONE chunk's log (what a replay of every chunk needs) -- more than one
    chunk's workspace, less than the uncapped stream's; beyond 64 chunks: one chunk's."""
    N, d, L, B = 9228, 128, 3, 1000
    row = 2 * L * d * 4                                                   # log bytes per edge
    full = hip_lib.tpnet_stream_workspace_bytes(N, d, L, 480 * B, B)
    assert full > 480 * B * row
    assert hip_lib.tpnet_stream_workspace_bytes_capped(N, d, L, 480 * B, B, 0) == full
    assert hip_lib.tpnet_stream_workspace_bytes_capped(N, d, L, 480 * B, B, 1000 * B * row) == full        # the cap does not bind
    one = hip_lib.tpnet_stream_workspace_bytes(N, d, L, 48 * B, B)
    four = hip_lib.tpnet_stream_workspace_bytes_capped(N, d, L, 480 * B, B, 120 * B * row)
    ten = hip_lib.tpnet_stream_workspace_bytes_capped(N, d, L, 480 * B, B, 48 * B * row)
    assert one < ten < four < full
    assert ten - one < 0.2 * 480 * B * row                                # (the ten plans without their logs: a fraction of the log)
    many = hip_lib.tpnet_stream_workspace_bytes_capped(N, d, L, 70 * 48 * B, B, 48 * B * row)              # 70 chunks: no arena
    assert many == one


def test_multi_gpu_planners_size_their_scratch_without_a_gpu(hip_lib):
    """ABI 7: the row shard on the windowed pipeline and the large exchange plan are sized by host arithmetic (no GPU call), and
    refuse nonsense."""
    import ctypes
    ws = hip_lib.tpnet_wshard_workspace_bytes(4614 + 9228, 128, 3, 20 * 2000, 2000, 2, 4614)
    assert ws > 20 * 2000 * 2 * 3 * 128 * 4                      # at least the version log of the call
    assert hip_lib.tpnet_wshard_workspace_bytes(0, 128, 3, 1000, 100, 2, 10) == 0
    assert hip_lib.tpnet_wshard_workspace_bytes(100, 128, 9, 1000, 100, 2, 10) == 0      # L > 4
    assert hip_lib.tpnet_wshard_workspace_bytes(100, 128, 3, 1000, 100, 65, 10) == 0     # more than 64 ranks
    big = hip_lib.tpnet_xplan_large_bytes(3 * 80000, 80000, 8)
    assert big > 6 * 3 * 80000 * 8 * 2 and hip_lib.tpnet_xplan_large_bytes(0, 100, 2) == 0
    h = ctypes.c_void_p()
    assert hip_lib.tpnet_stage_create_ex(8, 1 << 16, 7, ctypes.byref(h)) == -1          # no such mode
    assert hip_lib.tpnet_stage_in_device_memory(None) == 0
    assert hip_lib.tpnet_wshard_plan(None, None, None, None, None, 10, 5, 10, 2, 0, 5, 0.0, 0.0, 0, 1, 1, None, 0, None, ctypes.byref(h)) == -1
    assert hip_lib.tpnet_wshard_step(None, None, 0, 1, None, None, None) == -1 and hip_lib.tpnet_wshard_run(None, None, None, None, 1, None) == -1
    assert hip_lib.tpnet_strerror(-6).startswith(b"the one-launch encoder kernel")


def test_bad_arguments_are_rejected_without_a_gpu(hip_lib):
    from tpnet_amd import _lib
    st = _lib.State(p0=None, q=None, meta=None, N=10, d=16, L=3, err=None)
    assert hip_lib.tpnet_state_init(ctypes.byref(st), 0.0, None) == -1          # null pointers
    st = _lib.State(p0=8, q=8, meta=8, N=10, d=16, L=7, err=8)
    assert hip_lib.tpnet_pair_gram(ctypes.byref(st), None, None, 0, 0.0, 0.0, 0, None, None) == -1   # L > 4


def _mk(**kw):
    from tpnet_amd import RandomProjectionModule
    args = dict(node_num=50, edge_num=120, dim_factor=10, num_layer=3, time_decay_weight=1e-6, device="cpu",
                use_matrix=False, beginning_time=np.float64(12.5), not_scale=False, enforce_dim=32)
    args.update(kw)
    return RandomProjectionModule(**args)


def test_module_surface_matches_reference(golden_dir):
    """Constructor keywords, attributes, state-dict keys/dtypes/shapes as the reference (fixture G5 stores the
    reference's own state_dict keys)."""
    g = np.load(os.path.join(golden_dir, "g5_backup_reload.npz"))
    rp = _mk(node_num=int(g["N"]), enforce_dim=int(g["d"]))
    sd = rp.state_dict()
    assert sorted(sd.keys()) == [str(k) for k in g["state_dict_keys"]]
    got = [str(sd[k].dtype) + str(tuple(sd[k].shape)) for k in sorted(sd.keys())]
    assert got == [str(x) for x in g["state_dict_dtypes"]]
    assert rp.pair_wise_feature_dim == 64 and rp.dim == int(g["d"]) and rp.num_layer == 3
    assert rp.now_time.dtype == torch.float64 and rp.begging_time.dtype == torch.float64
    assert all(not p.requires_grad for p in rp.random_projections)
    assert isinstance(rp.mlp, torch.nn.Sequential) and rp.mlp[0].in_features == 64 and rp.mlp[2].out_features == 64
    for name in ("update", "get_random_projections", "get_pair_wise_feature", "reset_random_projections",
                 "backup_random_projections", "reload_random_projections"):
        assert callable(getattr(rp, name))


def test_dim_rule_and_use_matrix():
    rp = _mk(enforce_dim=-1, edge_num=157474 + 1, node_num=9228)     # Wikipedia: int(ln(2E))*10 = 120
    assert rp.dim == 120
    rp = _mk(use_matrix=True, node_num=30, enforce_dim=-1)
    assert rp.dim == 30 and torch.equal(rp.random_projections[0], torch.eye(30))


def test_triple_registration_state_dict_keys():
    """The reference registers the same module under three parents; keys must nest the same way."""
    rp = _mk()

    class Holder(torch.nn.Module):
        def __init__(self, r):
            super().__init__()
            self.random_projections = r

    model = torch.nn.Sequential(Holder(rp), Holder(rp))
    keys = model.state_dict().keys()
    assert "0.random_projections.random_projections.3" in keys and "1.random_projections.now_time" in keys
    model.load_state_dict(model.state_dict())


def test_no_cpu_fallback():
    from tpnet_amd import TPNetHipError
    rp = _mk()
    ids = np.array([1, 2, 3])
    with pytest.raises(TPNetHipError):
        rp.update(ids, ids, np.array([1.0, 2.0, 3.0]))
    with pytest.raises(TPNetHipError):
        rp.get_pair_wise_feature(ids, ids)
    with pytest.raises(TPNetHipError):
        rp.get_random_projections(ids)
    # state-management methods that need no kernel keep working on the CPU copy (checkpoint round trips)
    bk = rp.backup_random_projections()
    rp.reload_random_projections(bk)
    rp.reset_random_projections()
    assert float(rp.now_time) == 12.5


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "tpnet_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("no CPU fallback", ""), f"{f} mentions the oracle"


def test_module_copy_and_pickle_on_cpu():
    import copy, pickle
    rp = _mk()
    rp2 = copy.deepcopy(rp)
    assert torch.equal(rp2.random_projections[0], rp.random_projections[0])
    assert rp2.random_projections[0].data_ptr() != rp.random_projections[0].data_ptr()
    rp3 = pickle.loads(pickle.dumps(rp))
    assert sorted(rp3.state_dict().keys()) == sorted(rp.state_dict().keys())


def test_layer_list_is_lazy_about_layers_1_to_L():
    """random_projections[0], len() and .device do not count as reading layers 1..L (no dense export, no re-import);
    reading an entry >= 1 or iterating does."""
    rp = _mk()
    assert isinstance(rp.random_projections, torch.nn.ParameterList) and len(rp.random_projections) == 4
    rp._params_exposed = False
    rp.random_projections[0]
    rp.random_projections[-4]
    len(rp.random_projections)
    assert rp._params_exposed is False
    rp.random_projections[1]
    assert rp._params_exposed is True
    rp._params_exposed = False
    list(rp.random_projections)
    assert rp._params_exposed is True
    import copy
    rp2 = copy.deepcopy(rp)
    rp2._params_exposed = False
    rp2.random_projections[2]
    assert rp2._params_exposed is True and rp._params_exposed is True
    rp._params_exposed = False
    rp2.random_projections[3]
    assert rp._params_exposed is False                       # the copy's list belongs to the copy
    assert sorted(rp2.state_dict().keys()) == sorted(rp.state_dict().keys())


def test_encoder_pattern_detection_host_logic():
    """`_anchor_runs`: np.repeat(anchors, K) is recognised with any K whose blocks are constant (runs of equal anchors merge
    whole blocks only), anything else is refused."""
    from tpnet_amd import RandomProjectionModule as R
    a = np.array([5, 9, 9, 2, 7], dtype=np.int64)
    anchors, K = R._anchor_runs(np.repeat(a, 20))
    assert K == 20 and np.array_equal(anchors, a)
    anchors, K = R._anchor_runs(np.repeat(np.array([3, 3, 3], dtype=np.int64), 4))      # one long run: K = the whole length
    assert K == 12 and np.array_equal(np.repeat(anchors, K), np.repeat(3, 12))
    bad = np.repeat(a, 20).copy(); bad[21] = 4
    r = R._anchor_runs(bad)
    assert r is None or r[1] < 4                                                         # blocks of 1: not the encoder's call
    assert R._anchor_runs(np.arange(10, dtype=np.int64)) is None
    assert R._anchor_runs(np.array([1], dtype=np.int64)) is None


def test_fused_feature_applies_only_to_the_reference_mlp():
    """The one-launch readout + mlp serves exactly Linear(F, 4F) -> ReLU -> Linear(4F, F) with biases, fp32, on a GPU; anything
    else (the goldens' Identity, a wider layer, a CPU module) keeps the torch layers."""
    from tpnet_amd import fused_feature as ff
    mk = lambda a, b, c: torch.nn.Sequential(torch.nn.Linear(a, b), torch.nn.ReLU(), torch.nn.Linear(b, c))
    assert not ff.supported(mk(64, 256, 64), 64)                      # CPU weights
    assert not ff.supported(torch.nn.Identity(), 64)
    assert not ff.supported(mk(64, 128, 64), 64)
    assert ff.prepared(torch.nn.Identity(), 64) is None
    assert ff.prepared(mk(64, 256, 64), 64) is None                   # (CPU: not supported, no cache entry)
    assert ff.needs_grad(list(mk(4, 16, 4).parameters())) and not ff.needs_grad([torch.zeros(1)])


def test_product_library_has_no_developer_knobs(hip_lib):
    """Timing / tuning knobs (window length, thresholds, roles switched off ...) exist only in -DTPNET_DEV builds
    (libtpnet_hip_dev.so): the product library reads no environment variable of its own, and none of the knob names is
    compiled into it."""
    from tpnet_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"TPNET_DEV_" not in blob
    assert b"WIN_SKIP" not in blob and b"ROLE_MASK" not in blob


def test_plan_tag_layout(hip_lib):
    """tpnet_plan_tag as the header declares it: 3 words + 20 library-owned words."""
    from tpnet_amd import _lib
    assert ctypes.sizeof(_lib.PlanTag) == 23 * 8
    assert _lib.PlanTag.built.offset == 24


def test_mlp_and_state_structs_as_a_c_compiler_lays_them_out(tmp_path):
    """include/tpnet_hip.h compiled as plain C (gcc): sizeof / offsetof of tpnet_state and tpnet_mlp (ABI 6: + wimg) equal the
    ctypes mirrors a binding uses (tpnet_amd/_lib.py) -- a field added on one side only would shift every pointer behind it."""
    import shutil
    import subprocess
    from tpnet_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "layout.c"
    fields = ["w1t", "b1", "w2t", "b2", "F", "H", "w1", "w2f", "wimg"]
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "tpnet_hip.h"\nint main(void) {\n'
                   + 'printf("%zu %zu\\n", sizeof(tpnet_mlp), sizeof(tpnet_state));\n'
                   + "".join(f'printf("%zu\\n", offsetof(tpnet_mlp, {f}));\n' for f in fields)
                   + "return 0; }\n")
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(root, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == ctypes.sizeof(_lib.Mlp) and int(out[1]) == ctypes.sizeof(_lib.State)
    for f, off in zip(fields, out[2:]):
        assert getattr(_lib.Mlp, f).offset == int(off), f
