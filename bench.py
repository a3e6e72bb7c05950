#!/usr/bin/env python3
"""bench.py -- throughput of the temporal-walk-matrix hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch: pairwise readout of (src,dst) and (src,neg) on the pre-batch
state + update() of the batch (the decoder-level unit of SURVEY.md §8d).  Workload = BASELINE.json configs[1]
(C2: Wikipedia-shaped stream, d=128, batch=1000, L=3, fp32); inputs are resident in HBM when the timed region
starts.  Prints ONE JSON line (rank 0):
  value / ms_per_step   the K timed steps (ONE run_stream call, planning included; its argument views of the resident stream are
                        built before the clock starts), max over ranks
  roofline              the kernel those K steps ran on, timed live with HIP events over the same K batches
  epoch                 one epoch of the config's own stream (C2: 157 474 edges): wall clock, cold and with the plan replayed
  long_stream           2 048 batches of the same stream: the regime of streams of millions of edges
  cpu_baseline, dropin  the torch-CPU port on the host cores; the per-batch module API from host arrays
"""
import argparse
import os
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")   # kernel arguments in device memory (the ROCm 7 default; 2 us per launch otherwise)
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def make_workload(cfg, n_batches, rank):
    from tpnet_amd.stream import synthetic_stream, synthetic_negatives
    B = cfg["B"]
    E = n_batches * B
    span = cfg["span"] * (E / cfg["E"])          # same edge rate as the dataset-shaped stream
    src, dst, t, N = synthetic_stream(cfg["U"], cfg["I"], E, span, seed=1000 * rank)
    neg = synthetic_negatives(cfg["U"], N, E, B, seed=1 + 1000 * rank)
    return src, dst, neg, t, N


def _cpu_port_times(cfg, src, dst, neg, t, P0, threads, nb, with_mlp=False):
    """One pass of the torch-CPU port over batches [2, 2+nb) (2 warm-up batches before): seconds spent in the readouts and
    in the updates."""
    from oracle.torch_port import TorchPort
    torch.set_num_threads(threads)
    B = cfg["B"]
    port = TorchPort(P0, 3, cfg["lam"], 0.0)
    mlp = None
    if with_mlp:
        torch.manual_seed(0)
        mlp = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 64))
    tr = tu = 0.0
    with torch.no_grad():
        for b in range(2 + nb):
            s = slice(b * B, (b + 1) * B)
            t0 = time.perf_counter()
            f1 = port.pair_gram(src[s], dst[s])
            f2 = port.pair_gram(src[s], neg[s])
            if mlp is not None:
                mlp(torch.as_tensor(f1)); mlp(torch.as_tensor(f2))
            t1 = time.perf_counter()
            port.update(src[s], dst[s], t[s])
            t2 = time.perf_counter()
            if b >= 2:
                tr += t1 - t0
                tu += t2 - t1
    return tr, tu


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def _physical_cores():
    """Physical cores of the host (distinct (package, core) pairs of /proc/cpuinfo; logical CPUs / 2 as a fallback)."""
    try:
        seen, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        if seen:
            return len(seen)
    except OSError:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(cfg, src, dst, neg, t, N, P0, reps=3, quick=False):
    """the torch-CPU port of the reference ops (oracle/torch_port.py, eager dense decay included) on
    this box's host cores, on a bounded prefix of the same workload: 2 warm-up batches, `reps` repetitions, median.  Settings: ALL
    physical cores this process may use, 16 threads, and the reference's own 3 intra-op threads
    (train_link_prediction.py:124); update-only, readout-only and combined rates, with and without rp.mlp.  `value` = the FASTEST
    of the three, `cores` = the threads it used (more threads are not faster on these small ops: 128 threads of a 2 x 64-core host
    run 30x slower than 16 -- all three are reported).  quick: one repetition, no mlp leg (the N > 1 line)."""
    try:
        usable = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        usable = os.cpu_count() or 1
    threads_all = max(1, min(_physical_cores(), usable))
    threads16 = min(usable, 16)
    B = cfg["B"]
    nb = max(3, min(len(src) // B - 2, max(3, (150000 if quick else 400000) // B)))   # ~400 000 edges per repetition
    # (the all-cores setting on a shorter sample, once: 128 threads on these small ops run at ~0.013 M edges/s -- 30x slower than
    # 16 -- and the full sample would take minutes)
    nb_all = max(3, min(nb, 24000 // B))
    med = lambda xs: float(np.median(xs))
    settings = [("16thr", threads16, False, nb, reps), ("3thr", 3, False, nb, reps)]
    if threads_all != threads16:
        settings.append(("all", threads_all, False, nb_all, 1))
    if not quick:
        settings.append(("16thr_mlp", threads16, True, nb, reps))
    out = {}
    for name, th, mlp, nb_, reps_ in settings:
        runs = [_cpu_port_times(cfg, src, dst, neg, t, P0, th, nb_, mlp) for _ in range(reps_)]
        n = nb_ * B
        out[name] = {"combined": n / med([a + b for a, b in runs]), "readout_only": n / med([a for a, _ in runs]),
                     "update_only": n / med([b for _, b in runs]), "threads": th, "batches": nb_, "repetitions": reps_}
    if "all" not in out:
        out["all"] = out["16thr"]
    torch.set_num_threads(threads16)
    best = max(("16thr", "3thr", "all"), key=lambda k: out[k]["combined"])
    res = {"value": out[best]["combined"], "unit": "edges/s", "cores": out[best]["threads"], "cpu_model": _cpu_model(),
           "host_logical_cpus": os.cpu_count(), "host_physical_cores": _physical_cores(), "kind": "port",
           "value_is": f"the fastest of the three thread settings ({best})",
           "readout_only": out[best]["readout_only"], "update_only": out[best]["update_only"],
           "value_all_physical_cores": out["all"]["combined"], "threads_all_physical_cores": out["all"]["threads"],
           "value_16_threads": out["16thr"]["combined"], "value_3_threads": out["3thr"]["combined"],
           "readout_only_3_threads": out["3thr"]["readout_only"], "update_only_3_threads": out["3thr"]["update_only"],
           "repetitions": reps,
           "sample": f"{nb} batches of {B} edges after 2 warm-up batches, median of {reps} repetitions per setting (all physical cores: "
                     f"{out['all']['batches']} batches, once); torch-CPU port of the reference ops incl. its eager dense decay; pre-mlp "
                     f"features unless with_mlp"}
    if "16thr_mlp" in out:
        res["with_mlp"] = out["16thr_mlp"]["combined"]
    return res


def dropin_rate(cfg, rp, src, dst, neg, t, nb=30):
    """The per-batch module API the reference's loop calls (train_link_prediction.py:325-373): per batch two
    get_pair_wise_feature calls (rp.mlp included) and one update, from host numpy arrays.  Never `value`."""
    B = cfg["B"]
    nb = max(1, min(nb, len(src) // B - 3))
    def decoder_pass():
        rp.reset_random_projections()
        with torch.no_grad():
            for b in range(nb + 3):
                if b == 3:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                s = slice(b * B, (b + 1) * B)
                rp.get_pair_wise_feature(src[s], dst[s])
                rp.get_pair_wise_feature(src[s], neg[s])
                rp.update(src[s], dst[s], t[s])
            torch.cuda.synchronize()
        return time.perf_counter() - t0
    el = min(decoder_pass(), decoder_pass())         # (the faster of two passes, as for the encoder-level loops below)
    res = {"value": nb * B / el, "unit": "edges/s", "us_per_batch": el / nb * 1e6,
           "what": "module API per batch from host arrays: 2 x get_pair_wise_feature (with rp.mlp) + update; the faster of two "
                   "passes of %d batches" % nb}
    # the same loop as a TRAINING step issues it (train_link_prediction.py:321-386): gradients recorded, a scalar loss of the
    # features, zero_grad / backward / Adam step on rp.mlp every batch -- the prepared weight layouts are rebuilt after every step
    dev_ = rp._dev()
    opt = torch.optim.Adam(rp.mlp.parameters(), lr=1e-4)
    labels = torch.cat([torch.ones(B, device=dev_), torch.zeros(B, device=dev_)])
    lossf = torch.nn.BCEWithLogitsLoss()

    def train_step(feats):
        loss = lossf(torch.cat([f.sum(1) for f in feats]), labels)
        opt.zero_grad()
        loss.backward()
        opt.step()

    def decoder_train_pass():
        rp.reset_random_projections()
        for b in range(nb + 3):
            if b == 3:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            s = slice(b * B, (b + 1) * B)
            if len(src[s]) != B:
                break
            f1 = rp.get_pair_wise_feature(src[s], dst[s])
            f2 = rp.get_pair_wise_feature(src[s], neg[s])
            rp.update(src[s], dst[s], t[s])
            train_step((f1, f2))
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    def torch_only_step():
        """The same loss / zero_grad / backward / Adam step on plain torch layers over RESIDENT features (no readout, no update):
        what the training machinery around the hot path costs on this GPU whatever serves the features."""
        x = torch.rand(B, rp.pair_wise_feature_dim, device=dev_) * 8.0
        def go(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                train_step((rp.mlp(x), rp.mlp(x)))
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n
        go(5)
        return go(20)
    try:
        elt = min(decoder_train_pass(), decoder_train_pass())
        tonly = torch_only_step()
        res["train"] = {"value": nb * B / elt, "unit": "edges/s", "us_per_batch": elt / nb * 1e6,
                        "ratio_to_no_grad": elt / el,
                        "torch_step_alone_us": tonly * 1e6, "ratio_to_torch_step_alone": (elt / nb) / tonly,
                        "what": "the decoder-level loop as a training step issues it: gradients recorded, BCE loss of the two feature "
                                "sums, zero_grad / backward / Adam step on rp.mlp every batch (train_link_prediction.py:321-386)"}
    except Exception as ex:                       # noqa: BLE001 -- a secondary figure must not cost the main line
        res["train"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    # encoder-level unit (secondary): the encoder's two calls per batch on top -- 4*B*K pairs each in the
    # reference's tile / repeat pattern (models/TPNet.py:311-316), K = 20 synthetic neighbours per node
    K = 20
    # (24 batches per pass since round 4, 8 before: the loop's start -- the host is a call ahead before the GPU has anything to do --
    # and its final synchronisation are ~40 us that 8 batches of ~140 us did not amortise: 138-144 us per batch where the kernel
    # timeline of the same loop shows 131, tools/encoder_trace2.py)
    nbe = max(1, min(24, nb, len(src) // B - 4))
    rng = np.random.RandomState(7)
    N = rp.node_num
    calls = []                                   # the index arrays of every call, built before the clock starts (that part is the
    for b in range(nbe + 4):                     # reference encoder's own host work, not the module API)
        s = slice(b * B, (b + 1) * B)
        nodes = np.concatenate([src[s], dst[s]])
        per = []
        for anchors in ((src[s], dst[s]), (src[s], neg[s])):
            neigh = rng.randint(1, N, (len(nodes), K)).astype(np.int64)
            per.append((np.tile(neigh.reshape(-1), 2),
                        np.concatenate([np.repeat(np.tile(anchors[0], 2), K), np.repeat(np.tile(anchors[1], 2), K)])))
        calls.append(per)
    def encoder_pass():
        rp.reset_random_projections()
        with torch.no_grad():
            for b in range(nbe + 4):             # (4 warm-up batches = 8 long calls: every slot of the pinned staging ring exists)
                if b == 4:
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                s = slice(b * B, (b + 1) * B)
                for u_, v_ in calls[b]:
                    rp.get_pair_wise_feature(u_, v_)
                rp.get_pair_wise_feature(src[s], dst[s])
                rp.get_pair_wise_feature(src[s], neg[s])
                rp.update(src[s], dst[s], t[s])
            torch.cuda.synchronize()
        return time.perf_counter() - t0_
    # (four passes, the fastest one: a pass of 24 batches is ~4-6 ms, and one stray host stall -- a pinned allocation, a page fault in a
    # fresh buffer -- has shown up as 3x the figure; all four are listed)
    el_all = [encoder_pass() for _ in range(4)]
    el = min(el_all)
    res["encoder_level"] = {"value": nbe * B / el, "unit": "edges/s", "us_per_batch": el / nbe * 1e6,
                            "passes_us_per_batch": [x / nbe * 1e6 for x in el_all],
                            "what": f"+ 2 x get_pair_wise_feature on 4*B*K = {4 * B * K} pairs (K = {K}) per batch, host index arrays in the "
                                    "reference's tile / repeat layout (built before the clock starts; the module recognises the pattern and ships neighbours + anchors only)"}
    # the same unit with the ids resident on the device end to end (SURVEY section 8 f-3 -> f-2 -> f-1): the batch's src / dst /
    # neg / t are staged through the pinned ring (read in place by the row set-up kernel), the device sampler draws the K most recent neighbours of the 2B nodes (the neighbour ids
    # never visit the host), the anchored readout pairs every neighbour with the edge's two endpoints (no index arrays at
    # all), self.mlp runs behind it; then the decoder-level calls as above.  fp32 class; `bf16_mlp`: self.mlp on the bf16 matrix
    # cores (opt-in RandomProjectionModule.fused_mlp, 1e-2 class).
    try:
        from tpnet_amd.sampler import GpuRecentNeighborSampler
        dev = rp._dev()
        nall = (nbe + 4) * B
        smp = GpuRecentNeighborSampler(src[:nall], dst[:nall], t[:nall], device=str(dev), num_nodes=N)

        def device_loop(label):
            rp.reset_random_projections()
            with torch.no_grad():
                for b in range(nbe + 4):
                    if b == 4:
                        torch.cuda.synchronize()
                        t0_ = time.perf_counter()
                    s = slice(b * B, (b + 1) * B)
                    for other in (dst[s], neg[s]):
                        rp.encoder_pair_features(smp, src[s], other, t[s], K)
                    rp.get_pair_wise_feature(src[s], dst[s])
                    rp.get_pair_wise_feature(src[s], neg[s])
                    rp.update(src[s], dst[s], t[s])
                torch.cuda.synchronize()
            el_ = time.perf_counter() - t0_
            return {"value": nbe * B / el_, "unit": "edges/s", "us_per_batch": el_ / nbe * 1e6, "what": label}
        def device_train_loop():
            rp.reset_random_projections()
            for b in range(nbe + 4):
                if b == 4:
                    torch.cuda.synchronize()
                    t0_ = time.perf_counter()
                s = slice(b * B, (b + 1) * B)
                fe = [rp.encoder_pair_features(smp, src[s], other, t[s], K) for other in (dst[s], neg[s])]
                f1 = rp.get_pair_wise_feature(src[s], dst[s])
                f2 = rp.get_pair_wise_feature(src[s], neg[s])
                rp.update(src[s], dst[s], t[s])
                # (the encoder's features enter the loss too: their gradient reaches rp.mlp through 4*B*K rows per call)
                fx = [x if torch.is_tensor(x) else x[0] for x in fe]
                loss = lossf(torch.cat([f1.sum(1), f2.sum(1)]), labels) + sum(x.mean() for x in fx)
                opt.zero_grad()
                loss.backward()
                opt.step()
            torch.cuda.synchronize()
            el_ = time.perf_counter() - t0_
            return {"value": nbe * B / el_, "unit": "edges/s", "us_per_batch": el_ / nbe * 1e6,
                    "what": "encoder_level_device as a training step issues it (gradients through self.mlp of every call, Adam step per batch)"}
        if _lib_anchored_ok(rp):
            # (the fastest of four passes -- a pass is ~3 ms and the first two still run on a GPU that is clocking up; all four are listed)
            passes_ = [device_loop(
                f"neighbour ids resident on the device: 2 x encoder_pair_features on the batch's host arrays (staged, no copy; device sampler, K = {K}; "
                f"anchored readout of 4*B*K = {4 * B * K} pairs; self.mlp fp32) + the decoder-level calls; the fastest of four passes of {nbe} batches")
                for _ in range(4)]
            res["encoder_level_device"] = min(passes_, key=lambda r_: r_["us_per_batch"])
            res["encoder_level_device"]["passes_us_per_batch"] = [r_["us_per_batch"] for r_ in passes_]
            try:
                device_train_loop()
                tr_ = device_train_loop()
                tr_["ratio_to_no_grad"] = tr_["us_per_batch"] / res["encoder_level_device"]["us_per_batch"]
                res["encoder_level_device"]["train"] = tr_
            except Exception as ex:               # noqa: BLE001
                res["encoder_level_device"]["train"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
            if rp.num_layer == 3:
                rp.fused_mlp = True
                try:
                    res["encoder_level_device"]["bf16_mlp"] = device_loop("the same with self.mlp on the bf16 matrix cores (opt-in)")
                finally:
                    rp.fused_mlp = False
    except Exception as ex:                       # noqa: BLE001 -- a secondary figure must not cost the main line
        res["encoder_level_device"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    return res


def _lib_anchored_ok(rp):
    from tpnet_amd import _lib
    return bool(_lib.load().tpnet_pair_gram_anchored_supported(rp._st_ref()))


def copy_bandwidth_gbs(dev):
    """Measured device-to-device copy rate (read + write bytes / time) of a 1 GiB buffer: the practical HBM ceiling
    next to the 8 TB/s spec peak."""
    n = 1 << 28
    x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    y = torch.empty_like(x)
    y.copy_(x)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        y.copy_(x)
    e1.record()
    torch.cuda.synchronize()
    return 5 * 2 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9


METRIC = "temporal edges/sec (proj-update + pairwise readout)"


def _error_line(n_gpus, msg, **extra):
    line = {"metric": METRIC, "value": None, "unit": "edges/s", "n_gpus": n_gpus, "error": msg}
    line.update(extra)
    return json.dumps(line)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launcher_command(n, argv, port, python=None, script=None):
    """The child `python bench.py --gpus N ...` starts when nobody launched its ranks: the driver's own launch line
    (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <argv>),
    one rank per GPU.  Returns (argv list, environment)."""
    cmd = [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}",
           "--master-addr", "127.0.0.1", "--master-port", str(int(port)), script or os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    env["TPNET_BENCH_LAUNCHED_BY"] = str(os.getpid())
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return cmd, env


def pick_result_line(stdout_text):
    """Rank 0's JSON line among whatever the ranks and the launcher printed (the last line that parses and names the metric)."""
    found = None
    for ln in stdout_text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            try:
                obj = json.loads(ln)
            except ValueError:
                continue
            if isinstance(obj, dict) and "metric" in obj:
                found = obj
    return found


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher (WORLD_SIZE unset): start the N ranks as a CHILD process -- this parent has
    touched no GPU and never replaces itself -- relay rank 0's single JSON line, and fail unless all N ranks reported.  Returns
    the exit code."""
    import subprocess
    backend = os.environ.get("TPNET_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()           # (counts devices without creating a context on this image)
    if ndev < 1:
        print(_error_line(n, "bench.py needs a GPU (no CPU fallback): no HIP device visible"), flush=True)
        return 2
    if backend == "nccl" and n > ndev:
        print(_error_line(n, f"--gpus {n} but only {ndev} GPU(s) visible: one rank per GPU over RCCL (TPNET_BENCH_BACKEND=gloo "
                             f"rehearses the N > 1 path with the ranks sharing a GPU)", gpus_visible=ndev), flush=True)
        return 2
    cmd, env = launcher_command(n, argv, free_port())
    # the launcher and its ranks in a process group of their own: a run that outlives the limit is ended as a whole (the group this
    # call created, by its id -- a launcher killed alone would leave its ranks holding the GPUs)
    import signal
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, _ = proc.communicate(timeout=float(os.environ.get("TPNET_BENCH_LAUNCH_TIMEOUT", "1500")))
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        for sig, grace in ((signal.SIGTERM, 10.0), (signal.SIGKILL, 10.0)):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=grace)
                break
            except subprocess.TimeoutExpired:
                continue
        try:
            out = proc.communicate(timeout=5.0)[0] or ""
        except Exception:
            out = ""
        rc = 124
    line = pick_result_line(out)
    for ln in out.splitlines():                      # whatever else the ranks printed goes to stderr: stdout carries ONE line
        if not (ln.strip().startswith("{") and "metric" in ln):
            print(ln, file=sys.stderr)
    if line is None:
        print(_error_line(n, f"the {n} ranks printed no result line (launcher exit code {rc})"), flush=True)
        return rc or 1
    print(json.dumps(line), flush=True)
    if rc != 0:
        return rc
    if line.get("value") is None or line.get("n_gpus") != n or line.get("ranks_seen") != n:
        print(f"bench.py: expected {n} ranks, the line says n_gpus={line.get('n_gpus')} ranks_seen={line.get('ranks_seen')}",
              file=sys.stderr)
        return 1
    return 0


def csrc_fingerprint():
    """sha256 over the kernel sources (tpnet_amd/csrc/*.hip|hpp|h|c + include/*.h, names and contents): what a committed PMC entry
    is stamped with, and what bench.py recomputes at run time -- .git does not travel to the GPU box, file contents do."""
    import hashlib
    h = hashlib.sha256()
    files = []
    for dname in (os.path.join(ROOT, "tpnet_amd", "csrc"), os.path.join(ROOT, "include")):
        for fn in sorted(os.listdir(dname)):
            if fn.endswith((".hip", ".hpp", ".h", ".c")):
                files.append(os.path.join(dname, fn))
    for fp in files:
        h.update(os.path.basename(fp).encode() + b"\0")
        h.update(open(fp, "rb").read())
    return h.hexdigest()[:16]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None)
    ap.add_argument("--steps", type=int, default=4000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--config", default="C2")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropin", action="store_true")
    args = ap.parse_args()

    # --gpus N with nobody having launched the ranks (the way the driver runs --gpus 1): this process starts them as a child
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and (args.gpus or 1) > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if args.gpus is not None and args.gpus != world:
        # never a silent n_gpus: 1 -- the flag and the launch must agree
        if rank == 0:
            print(_error_line(world, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                                     f"(or run `python bench.py --gpus {args.gpus}` without a launcher)"), flush=True)
        raise SystemExit(2)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # TPNET_BENCH_BACKEND=gloo rehearses the N>1 path with several ranks sharing one GPU (development only)
    backend = os.environ.get("TPNET_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % ndev if backend != "nccl" else local_rank)
    torch.cuda.set_device(dev)
    dist = None
    # TPNET_BENCH_FORCE_DIST=1 (development): take the N>1 code path, collectives included, with a single rank
    force_dist = os.environ.get("TPNET_BENCH_FORCE_DIST") == "1"
    result_fd = None
    if world > 1 or force_dist:
        # stdout carries ONE line: whatever the communication library prints there (RCCL's version banner goes to stdout) is sent
        # to stderr for the life of the process; the result line is written to the original stdout
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)
        import torch.distributed as dist
        if env_world is None:                  # (the forced one-rank run: a rendezvous of its own)
            os.environ.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1",
                               "MASTER_PORT": str(free_port())})
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    if world > 1:
        # a collective that never returns (a rank lost, a communicator that cannot form) must not hang the run: after
        # TPNET_BENCH_WATCHDOG seconds (default 420) rank 0 reports the failure as its JSON line and every rank exits
        import threading

        def _give_up():
            if rank == 0:
                os.write(result_fd if result_fd is not None else 1,
                         (_error_line(world, "multi-GPU run did not finish in time (watchdog); set TPNET_ROWS_C_LOOP=0 to route the "
                                      "exchange through torch.distributed, TPNET_BENCH_EXCHANGE=allgather for the all-gather variant") + "\n").encode())
            os._exit(3)
        _wd = threading.Timer(float(os.environ.get("TPNET_BENCH_WATCHDOG", "420")), _give_up)
        _wd.daemon = True
        _wd.start()

    import tpnet_amd
    from tpnet_amd import _lib
    from tpnet_amd.stream import CONFIGS, bytes_per_edge
    cfg = CONFIGS[args.config]
    B, d, L = cfg["B"], cfg["d"], 3
    K, W = args.steps, args.warmup
    lib_ = _lib.load()

    # N = 1: configs[1] as is.  N > 1: the SAME graph and table cut over the ranks, weak scaling: the global batch is N*B
    # edges per step (B per GPU), every rank holds the whole edge stream (32 bytes per edge).
    #   rows (default, `value`): row sharding, BASELINE.json north_star's layout -- owner(n) = n % N, every rank holds ONLY
    #                   its rows (all layers) + a halo; one RCCL all-gather of the touched rows' bundles per step
    #                   (tpnet_amd/sharded.py: ShardedStreamRunner).
    #   cols (ablation, reported under "col_sharded"; TPNET_BENCH_SHARD=cols makes it the main leg): column sharding --
    #                   every rank keeps d/N columns of every row; the update needs no exchange, the raw Gram entries are
    #                   reduce-scattered per chunk of steps behind the next chunk's kernels (ColumnShardedRunner).
    Bg = B * world
    cfg_run = dict(cfg, B=Bg)
    # `long_stream` (rank 0, N = 1, beside the main figures) runs ROOF_STEPS batches of the same stream, so it is generated that long
    ROOF_STEPS = 2048
    Kr = max(K, ROOF_STEPS) if world == 1 else K
    src, dst, neg, t, N = make_workload(cfg_run, W + Kr, 0)
    shard = os.environ.get("TPNET_BENCH_SHARD", "rows") if (world > 1 or force_dist) else "single"
    cols_ok = not (d % world or (d // world) % 4)
    if shard == "cols" and not cols_ok:
        shard = "rows"
    to_dev = lambda x: torch.from_numpy(x).to(dev)
    d_src, d_dst, d_neg, d_t = to_dev(src), to_dev(dst), to_dev(neg), to_dev(t)
    P0 = None

    def make_full_module():
        nonlocal P0
        torch.manual_seed(0)
        P0 = torch.normal(0, 1 / np.sqrt(d), (N, d))
        m = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=cfg["E"], dim_factor=10, num_layer=L,
                                             time_decay_weight=cfg["lam"], device=str(dev), use_matrix=False,
                                             beginning_time=np.float64(0.0), not_scale=False, enforce_dim=d)
        m.random_projections[0].data = P0.clone()
        return m.to(dev)

    def write_line(text):
        if result_fd is None:
            print(text, flush=True)
        else:
            sys.stdout.flush()
            os.write(result_fd, (text + "\n").encode())

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()


    warm_windowed = False
    rp = None

    def time_leg(run, k_steps, prep=None):
        """W untimed steps, then exactly k_steps timed ones between barrier + synchronize; max over ranks.  `prep(a, b)` (optional)
        builds a call's ARGUMENTS -- tensor views of the resident stream, the last timestamp as a float -- and runs before the
        clock starts: the inputs of the timed steps exist when the region begins, the region holds the call itself."""
        if W > 1:
            # the W warm-up steps as up to FOUR calls: the first calls of a process pay one-time costs (lazy kernel loading,
            # allocator and stream set-up in the runtime, cold host caches) that a single short call does not absorb
            # (tools/sync_latency.py, 20 steps at C2: call 3 of a process 181-199 us, call 4 180-187, call 5 175-177, steady 172-175)
            # (where the timed steps take the windowed pipeline, every warm-up call must be long enough to take it too -- at least
            # four batches -- or the pipeline's kernels are first launched, i.e. loaded, inside the timed region: 2.5 ms instead of 0.15)
            ncall = min(W, 4) if not warm_windowed else max(1, min(4, W // 4))
            cuts = [W * i // ncall for i in range(ncall + 1)]
            for a_, b_ in zip(cuts[:-1], cuts[1:]):
                run(a_, b_)
        elif W > 0:
            run(0, W)
        pre = prep(W, W + k_steps) if prep is not None else None
        barrier()
        t0 = time.perf_counter()
        if pre is not None:
            run(W, W + k_steps, pre)
        else:
            run(W, W + k_steps)
        # the clock of a rank stops when ITS k_steps are done (synchronize); the barrier behind it closes the bracket, and the
        # figure is the MAX over the ranks -- a collective barrier inside every rank's clock was ~100-200 us of RCCL latency on top
        # of a 150-us region, and measured the barrier, not the steps (the ranks of a sharded stream meet at every step's exchange)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        barrier()
        if dist is not None:
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el

    CSRC_SHA = csrc_fingerprint()

    def pmc_traffic(cfg_name, kernel_short, edges_per_launch):
        """Memory-side bytes per launch of this kernel from the NEWEST committed rocprofv3 PMC passes of the config
        (tools/prof_round.sh -> profiles/r*_<config>_pmc.json), accepted only for the same kernel at the same work per launch
        (+-10 %) AND only when the entry was taken on exactly these kernel sources (`csrc_sha` = csrc_fingerprint(): .git does not
        travel to the GPU box, file contents do).  Otherwise traffic is None and traffic_source says why."""
        import glob
        import re
        if cfg_name is None:
            return None, {"stale": True, "reason": "no PMC pass for this workload"}
        paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{cfg_name}_pmc.json")),
                       key=lambda q: int(re.search(r"r(\d+)_", os.path.basename(q)).group(1)), reverse=True)
        why = "no committed PMC file for this config"
        for path in paths:
            try:
                ents = json.load(open(path)).get("kernels", [])
            except Exception:
                continue
            for ent in ents:
                if ent.get("kernel") != kernel_short or not ent.get("traffic_bytes_per_launch") or \
                        abs(ent.get("edges_per_launch", 0) - edges_per_launch) > 0.1 * edges_per_launch:
                    continue
                src_ = {"file": os.path.relpath(path, ROOT), "commit": ent.get("commit"), "csrc_sha": ent.get("csrc_sha"),
                        "command": ent.get("command"), "edges_per_launch": ent.get("edges_per_launch")}
                if ent.get("csrc_sha") == CSRC_SHA:
                    src_["stale"] = False
                    return ent["traffic_bytes_per_launch"], src_
                why = (f"newest matching entry ({src_['file']}, commit {src_['commit']}) was taken on other kernel sources "
                       f"(csrc_sha {ent.get('csrc_sha')} != {CSRC_SHA}): re-run tools/prof_round.sh")
            if why.startswith("newest"):
                break
        return None, {"stale": True, "reason": why, "csrc_sha_now": CSRC_SHA}

    def roof_pass(a_src, a_dst, a_neg, a_t, n_edges_in, t_now, t_last, o_pos, o_neg, flags=0, rp_=None, B_=None, d_=None, N_=None,
                  cfg_name="__main__"):
        rp_ = rp if rp_ is None else rp_
        B_ = B if B_ is None else B_
        d_ = d if d_ is None else d_
        N_ = N if N_ is None else N_
        cfg_name = args.config if cfg_name == "__main__" else cfg_name
        lib = _lib.load()
        st = rp_._state()
        ws = rp_._workspace(n_edges_in, B_, stream=True)
        total_ms, kern_ms = C.c_float(0), C.c_float(0)
        n_launch, n_edges = C.c_int64(0), C.c_int64(0)
        nbat = (n_edges_in + B_ - 1) // B_
        lid = rp_._next_launch_ids(3 * nbat + 8)
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.tpnet_time_stream(C.byref(st), a_src.data_ptr(), a_dst.data_ptr(), a_neg.data_ptr(),
                                         a_t.data_ptr(), n_edges_in, B_, t_now, float(rp_.time_decay_weight), lid, flags,
                                         o_pos.data_ptr(), o_neg.data_ptr(), ws.data_ptr(), ws.numel(), 1,
                                         C.byref(total_ms), C.byref(kern_ms), C.byref(n_launch), C.byref(n_edges), stream),
                   "time_stream")
        rp_._now_host = t_last
        rp_._params_valid = False
        rp_._now_dirty = True
        rp_._table_written()
        bpe = bytes_per_edge(d_, L)
        windowed = n_launch.value > 0 and n_launch.value < nbat                     # fewer launches than batches
        bytes_per_launch = bpe * n_edges.value / max(1, n_launch.value)
        achieved = bytes_per_launch / (kern_ms.value * 1e-3) / 1e9 if kern_ms.value > 0 else 0.0
        kshort = "k_wpipe" if windowed else "k_step"
        traffic, traffic_src = pmc_traffic(cfg_name, kshort, n_edges.value / max(1, n_launch.value))
        kname = ("k_wpipe (windowed schedule: one launch per pipeline step = layer i of window j-i+1, i=1..L, + the readouts of "
                 "window j-L)") if windowed else "k_step (fused readout + update of one batch; one launch per step)"
        # (cache-resident configs: the algorithmic rate can exceed what the memory side delivers; the memory-side rate from the
        # committed counters -- bytes that really crossed the Infinity Cache / HBM boundary per launch -- is the one to hold
        # against the 8 TB/s peak there)
        mem_gbs = traffic / (kern_ms.value * 1e-3) / 1e9 if (traffic and kern_ms.value > 0) else None
        # A table that sits in the 256 MB Infinity Cache is not served by HBM: where the section-8(d) byte count over time exceeds
        # the HBM peak (the fused kernel moves fewer bytes than the unfused count at such shapes), the line says bound = "cache" and
        # `frac` is the memory-side fraction (PMC bytes over time over 8 TB/s; null without counters) -- never a fraction above 1
        # under bound = "hbm".  The algorithmic figure stays beside it.
        frac_alg = achieved / HBM_PEAK_GBS
        cache_bound = (28.0 * N_ * d_ < 256e6) and frac_alg > 1.0
        return {"bound": "cache" if cache_bound else "hbm", "kernel": kname, "kernel_short": kshort, "windowed": windowed,
                "achieved": (mem_gbs if cache_bound else achieved), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ((mem_gbs / HBM_PEAK_GBS) if mem_gbs else None) if cache_bound else frac_alg,
                "algorithmic_gbs": achieved, "algorithmic_frac": frac_alg,
                "traffic": traffic, "traffic_source": traffic_src, "steps": nbat,
                "memory_side_gbs": mem_gbs, "memory_side_frac": (mem_gbs / HBM_PEAK_GBS) if mem_gbs else None,
                "algorithmic_bytes_per_launch": bytes_per_launch, "launches": n_launch.value,
                "edges_per_launch": n_edges.value / max(1, n_launch.value),
                "avg_launch_period_us": kern_ms.value * 1e3,
                "stream_ms_events": total_ms.value,
                "end_to_end_frac": (bpe * n_edges_in / (total_ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS) if total_ms.value > 0 else None}

    def sharded_rows(N_, d_, lam_, E_cfg, Bg_, arrs, t_host, k_steps, draw_on_device=False):
        """The row-sharded runner (tpnet_amd/sharded.py: ShardedStreamRunner, targeted exchange) over the resident stream `arrs` =
        (src, dst, neg, t) in global batches of Bg_: W warm-up steps, k_steps timed ones (time_leg), then an extra pass over the SAME
        timed batches with HIP events on the launch stream around every step (every rank: the exchange is collective).  Returns
        the elapsed seconds, the bytes of table a rank holds, and the kernel / exchange figures of THIS rank."""
        from tpnet_amd.sharded import ShardedStreamRunner
        a_src, a_dst, a_neg, a_t = arrs
        # halo rows: 3 per edge of a batch are always enough for the per-batch shard; the windowed shard keeps ONE row per remote node a
        # whole call touches -- on a table this small (C2: 9 228 rows) that is every remote node
        halo_ = max(3 * Bg_, min(N_, 65536)) if N_ <= 65536 else 3 * Bg_
        runner = ShardedStreamRunner.create(node_num=N_, edge_num=E_cfg, dim=d_, num_layer=L, time_decay_weight=lam_,
                                            device=dev, beginning_time=np.float64(0.0), halo_rows=halo_,
                                            draw_on_device=draw_on_device)
        # targeted exchange (every row only to the ranks that read it, received in place, grouped ncclSend / ncclRecv from C);
        # TPNET_BENCH_EXCHANGE=allgather: one all-gather of every touched row per step
        runner.exchange = os.environ.get("TPNET_BENCH_EXCHANGE", "targeted")
        rp_ = runner.rp
        rp_._ensure_engine()
        NG_ = rp_.pair_wise_feature_dim
        n_max = max(k_steps, W) * Bg_
        # outputs and workspaces exist, and were touched, before any clock starts (as on the single-GPU path)
        o_pos = torch.zeros((n_max, NG_), dtype=torch.float32, device=dev)
        o_neg = torch.zeros((n_max, NG_), dtype=torch.float32, device=dev)
        single_sched = runner.G == 1 and runner.single_rank_pipeline          # (a forced one-rank run: the module's own schedules)
        runner.reserve_stream(n_max, Bg_)
        if single_sched:
            rp_._eng["ws"].zero_()
            ws_b = rp_._eng["ws"].numel()
            will_window = bool(lib_.tpnet_stream_schedule(rp_.node_num, d_, L, k_steps * Bg_, Bg_, 0, ws_b) == 1)
        else:
            # (G > 1: from 16 batches the stream takes the windowed shard; the warm-up call takes it too, from 4 batches, so that its
            # kernels and buffers exist before the clock starts)
            will_window = bool(runner.G > 1 and runner.windowed and k_steps >= runner.windowed_min_batches)
        min_b = runner.windowed_min_batches

        def run(a, b_, timing=None):
            sl_ = slice(a * Bg_, b_ * Bg_)
            n_ = (b_ - a) * Bg_
            runner.windowed_min_batches = 4 if (will_window and b_ <= W) else min_b
            t_last = t_host[np.minimum(np.arange(a + 1, b_ + 1) * Bg_, len(t_host)) - 1]
            # (a warm-up call of >= 4 batches runs the schedule the timed call will run: its kernels are loaded before the clock starts)
            runner.schedule = "windowed" if (will_window and b_ <= W and b_ - a >= 4) else None
            # features stay sharded by the owner of the pair's src node (a sharded decoder consumes them in place)
            runner.run_stream(a_src[sl_], a_dst[sl_], a_neg[sl_], a_t[sl_], Bg_, t_host_last=t_last, merge_outputs=False,
                              out_pos=o_pos[:n_], out_neg=o_neg[:n_], timing=timing)
        nonlocal warm_windowed
        warm_windowed = will_window
        # (as on the single-GPU path: a short region is measured five times in this process, each from a reset table, and the
        # figure is the median; the first region -- the second call of its shape in the process: 1.5-3x slower -- stays beside it)
        regs = []
        for r_ in range(5 if k_steps <= 64 else 1):
            if r_ > 0:
                rp_.reset_random_projections()
            regs.append(time_leg(run, k_steps))
        el = float(np.median(regs))
        warm_windowed = False
        runner.check_device_errors()
        tb = runner.table_bytes()
        # ---- the kernel those steps ran on, timed live: an extra pass over the same batches
        bpe_ = bytes_per_edge(d_, L)
        info = {}
        if single_sched:
            runner.rp.reset_random_projections()
            sl_ = slice(W * Bg_, (W + k_steps) * Bg_)
            info["roof"] = roof_pass(a_src[sl_], a_dst[sl_], a_neg[sl_], a_t[sl_], k_steps * Bg_, 0.0, float(t_host[(W + k_steps) * Bg_ - 1]),
                                     o_pos[:k_steps * Bg_], o_neg[:k_steps * Bg_], rp_=rp_, B_=Bg_, d_=d_, N_=rp_.node_num,
                                     cfg_name=None)
        else:
            tm = {}
            rp_.reset_random_projections()
            run(W, W + k_steps, timing=tm)
            row_b = (L + 1) * d_ * 4
            transport = "rccl (C loop)" if backend == "nccl" else backend
            if tm.get("windowed"):
                # the shard ran on the windowed pipeline (csrc/wshard.hip): one k_wpipe launch + one exchange per window of batches
                nl = max(1, int(tm.get("launches") or 1))
                per_launch_edges = k_steps * Bg_ / world / nl   # this rank's share: the pairs / targets it owns
                ach = bpe_ * per_launch_edges / (tm["step_ms"] * 1e-3) / 1e9 if tm.get("step_ms") else 0.0
                info["roof"] = {"bound": "hbm", "kernel": "k_wpipe (windowed schedule on a row shard: one launch per pipeline step = layer i of "
                                "window j-i+1 + the readouts of window j-L, restricted to the chains / pairs this rank owns; one exchange "
                                "behind every launch)", "kernel_short": "k_wpipe", "windowed": True,
                                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                "traffic_source": {"stale": True, "reason": "no PMC pass of the sharded pipeline exists (one-GPU builder pool)"},
                                "algorithmic_bytes_per_launch": bpe_ * per_launch_edges, "edges_per_launch": per_launch_edges,
                                "launches": nl, "steps": k_steps, "avg_launch_period_us": tm.get("step_ms", 0.0) * 1e3,
                                "stream_ms_events": tm.get("total_ms"), "rank": rank,
                                "duration_note": "HIP events on the launch stream around every k_wpipe launch of rank 0 (tpnet_time_wshard_run), an "
                                                 "extra pass over the timed batches; achieved = section-8(d) bytes per edge x this rank's share "
                                                 "(1 / N) of the edges a launch covers / the average launch duration",
                                "exchange": {"what": "per launch: pack + ONE grouped ncclSend / ncclRecv + unpack of the launch's results that "
                                                     "other ranks read (log slots, layer by layer); once per call: the touched rows of every "
                                                     "rank to every other (the chunk's halo rows)",
                                             "avg_us_per_launch": tm.get("exchange_ms", 0.0) * 1e3,
                                             "rows_sent_per_launch": tm.get("rows_sent_per_launch"),
                                             "rows_received_per_launch": tm.get("rows_received_per_launch"),
                                             "bytes_sent_per_launch": (tm.get("rows_sent_per_launch") or 0.0) * d_ * 4,
                                             "bytes_received_per_launch": (tm.get("rows_received_per_launch") or 0.0) * d_ * 4,
                                             "bytes_per_launch_per_peer": ((tm.get("rows_sent_per_launch") or 0.0) * d_ * 4 / (world - 1)) if world > 1 else 0.0,
                                             "chunk_rows_sent": tm.get("chunk_rows_sent"), "chunk_rows_received": tm.get("chunk_rows_received"),
                                             "chunk_bytes_received": (tm.get("chunk_rows_received") or 0) * row_b,
                                             "row_bytes": d_ * 4, "transport": transport}}
            else:
                per_launch_edges = Bg_ / world                      # this rank's share of a step: the pairs / targets it owns
                ach = bpe_ * per_launch_edges / (tm["step_ms"] * 1e-3) / 1e9 if tm.get("step_ms") else 0.0
                xc = runner.__dict__.get("_xplan_cache")
                R_ = xc[1] if xc else None
                rows_s = float(np.mean(R_["stot"])) if R_ is not None else 0.0
                rows_r = float(np.mean(R_["rtot"])) if R_ is not None else 0.0
                info["roof"] = {"bound": "hbm", "kernel": "k_step (fused readout + update of one batch, restricted to the pairs / targets this rank "
                                "owns; one launch per step behind the step's exchange)", "kernel_short": "k_step", "windowed": False,
                                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                                "traffic_source": {"stale": True, "reason": "no PMC pass of the sharded step exists (one-GPU builder pool)"},
                                "algorithmic_bytes_per_launch": bpe_ * per_launch_edges, "edges_per_launch": per_launch_edges,
                                "launches": tm.get("batches"), "steps": tm.get("batches"),
                                "avg_launch_period_us": tm.get("step_ms", 0.0) * 1e3, "stream_ms_events": tm.get("total_ms"),
                                "rank": rank,
                                "duration_note": "HIP events on the launch stream around every step launch of rank 0 (tpnet_time_rows_stream_targeted), "
                                                 "an extra pass over the timed batches; achieved = section-8(d) bytes per edge x this rank's "
                                                 "share of the global batch (1 / N of it) / the average step duration",
                                "exchange": {"what": "per step: one pack launch + ONE grouped ncclSend / ncclRecv (rows received in place in the halo)",
                                             "avg_us_per_step": tm.get("exchange_ms", 0.0) * 1e3,
                                             "rows_sent_per_step": rows_s, "rows_received_per_step": rows_r,
                                             "bytes_sent_per_step": rows_s * row_b, "bytes_received_per_step": rows_r * row_b,
                                             "bytes_per_step_per_peer": (rows_s * row_b / (world - 1)) if world > 1 else 0.0,
                                             "row_bytes": row_b, "transport": transport}}
            info["schedule"] = "windowed" if tm.get("windowed") else "batch"
        info["regions"] = {"n": len(regs), "value_is": "median", "wall_us": [x * 1e6 for x in regs]}
        runner.close()
        del o_pos, o_neg
        return el, tb, info

    def rows_leg(k_steps):
        return sharded_rows(N, d, cfg["lam"], cfg["E"], Bg, (d_src, d_dst, d_neg, d_t), t, k_steps)
    halo_rows_main = max(3 * Bg, min(N, 65536)) if N <= 65536 else 3 * Bg

    def cols_leg(k_steps):
        from tpnet_amd.sharded import ColumnShardedRunner
        crun = ColumnShardedRunner.create(node_num=N, edge_num=cfg["E"], dim=d, num_layer=L,
                                          time_decay_weight=cfg["lam"], device=dev, beginning_time=np.float64(0.0))

        def run(a, b_):
            sl_ = slice(a * Bg, b_ * Bg)
            ends = [a * Bg + hi - 1 for _, hi in crun.chunk_bounds((b_ - a) * Bg, Bg)]
            crun.run_stream(d_src[sl_], d_dst[sl_], d_neg[sl_], d_t[sl_], Bg, merge="scatter", t_chunk_last=t[ends])
        el = time_leg(run, k_steps)
        crun.rp.check_device_errors()
        return el

    out_pos = out_neg = None
    regions = []
    if shard == "single":
        rp = make_full_module()
        NG = rp.pair_wise_feature_dim
        out_pos = torch.empty((K * Bg, NG), dtype=torch.float32, device=dev)
        out_neg = torch.empty((K * Bg, NG), dtype=torch.float32, device=dev)
        # buffers are touched once before any clock starts (torch.empty hands out pages the GPU has never mapped; their
        # first touch inside a 200 us timed region showed as 25 % run-to-run spread)
        rp.reserve_stream(K * Bg, Bg)                # (the timed call's workspace exists before the warm-up steps run)
        rp._eng["ws"].zero_()
        out_pos.zero_()
        out_neg.zero_()
        import gc
        gc.collect()
        gc.disable()                              # (no collector pause inside the timed region; re-enabled behind it)

        # the warm-up steps run the schedule the timed steps will run ("auto" picks the windowed pipeline from 16 batches of
        # <= 2048 edges, 56 larger ones; a warm-up call shorter than that would otherwise leave the pipeline's kernels to be
        # loaded inside the timed region: HIP resolves every kernel at its first launch, ~0.3 ms each)
        # (asked of the library -- tpnet_stream_schedule -- not restated here: when the C side's threshold moves, this follows)
        timed_windowed = lib_.tpnet_stream_schedule(N, d, L, K * Bg, Bg, 0, rp._eng["ws"].numel()) == 1
        warm_windowed = timed_windowed

        def prep_args(a, b_):
            sl_ = slice(a * Bg, b_ * Bg)
            sched = "windowed" if (timed_windowed and b_ <= W and b_ - a >= 4) else None
            return (d_src[sl_], d_dst[sl_], d_neg[sl_], d_t[sl_], out_pos[:(b_ - a) * Bg], out_neg[:(b_ - a) * Bg],
                    float(t[b_ * Bg - 1]), sched)

        # the timed steps go through the module's PREPARED call (RandomProjectionModule.prepare_stream: the arguments' devices / dtypes /
        # shapes checked and the call's flags formed once, before the clock -- what an epoch loop that runs the same stream every epoch
        # holds on to); the same steps through plain run_stream (arguments checked inside the region, ~10 us of a ~130-us region) are
        # timed behind them and reported beside `value` as timed_regions.plain_call
        def prep(a, b_):
            s_, d_, n_, t_, op_, on_, te_, sched = prep_args(a, b_)
            return rp.prepare_stream(s_, d_, n_, t_, Bg, out_pos=op_, out_neg=on_, t_end=te_, schedule=sched, replay=False)

        def run(a, b_, pre=None):
            (pre if pre is not None else prep(a, b_))()

        def run_plain(a, b_, pre=None):
            s_, d_, n_, t_, op_, on_, te_, sched = pre if pre is not None else prep_args(a, b_)
            rp.run_stream(s_, d_, n_, t_, Bg, out_pos=op_, out_neg=on_, t_end=te_, schedule=sched, replay=False)
        # a short timed region (the driver's 20 steps are ~150 us) is ONE sample of a quantity that scatters by 10-20 % with the
        # state the process and the GPU's clocks are in: up to 64 steps the region is measured nine times in this process (five until the end of round 5) -- each
        # time from a reset table: W warm-up steps, then the K timed steps -- and `value` is the median; the first region's number
        # stays beside it (`first_region`: the first timed call of a process is 25-40 us slower than every later one, eight runs of
        # the driver's line in profiles/r04_driver_line_runs.md)
        # (nine regions since the end of round 5, five before: the first region of a process is always the slow one -- host time, DESIGN.md
        # section 7 item 5 -- and one more in five is often disturbed by the host, so the median of five sat in the upper half of the quiet
        # ones; every region's time stays in the line)
        n_regions = 9 if K <= 64 else 1
        regions = []
        for r_ in range(n_regions):
            if r_ > 0:
                rp.reset_random_projections()
            regions.append(time_leg(run, K, prep))
        elapsed = float(np.median(regions))
        plain_regions = []
        copy_regions = []
        if n_regions > 1:
            for r_ in range(5):
                rp.reset_random_projections()
                plain_regions.append(time_leg(run_plain, K, prep_args))
            # the PCIe-inclusive rate (never `value`): the timed steps' four arrays start in pinned HOST memory and are copied to the
            # device inside the region, in front of the same prepared call
            sl_k = slice(W * Bg, (W + K) * Bg)
            h_pin = [torch.from_numpy(np.ascontiguousarray(x[sl_k])).pin_memory() for x in (src, dst, neg, t)]
            d_dst_views = [d_src[sl_k], d_dst[sl_k], d_neg[sl_k], d_t[sl_k]]

            def run_copy(a, b_, pre=None):
                if pre is not None:
                    for hv, dv in zip(h_pin, d_dst_views):
                        dv.copy_(hv, non_blocking=True)
                (pre if pre is not None else prep(a, b_))()
            for _ in range(2):                           # (the copy path's own first-use costs stay outside the regions)
                for hv, dv in zip(h_pin, d_dst_views):
                    dv.copy_(hv, non_blocking=True)
            torch.cuda.synchronize()
            for r_ in range(5):
                rp.reset_random_projections()
                copy_regions.append(time_leg(run_copy, K, prep))
        gc.enable()
        rp.check_device_errors()
    elif shard == "cols":
        elapsed = cols_leg(K)
    else:
        elapsed, row_bytes, rows_info = rows_leg(K)

    par = {"single": "single GPU",
           "cols": f"columns sharded over {world} GPUs ({d // world} of {d} per GPU), global batch {Bg} = {B} per GPU, no "
                   f"per-step collective; the 36 distinct raw Gram entries per pair are reduce-scattered (RCCL) per "
                   f"chunk of ~2M edges behind the next chunk's kernels",
           "rows": f"rows sharded over {world} GPUs (owner = id % {world}; every GPU holds only its {(N + world - 1) // world} "
                   f"rows of all {L + 1} layers + {halo_rows_main} halo rows), global batch {Bg} = {B} per GPU; from 16 batches on the windowed "
                   f"pipeline (one k_wpipe launch + one grouped RCCL send / recv per window of batches), else per step one grouped RCCL "
                   f"send / recv of the rows each peer reads, received in place ({os.environ.get('TPNET_BENCH_EXCHANGE', 'targeted')} exchange)"}[shard]

    def emit(row_info=None, roof=None, cpu=None, dropin=None, extra=None):
        line = {
            "metric": METRIC,
            "value": K * Bg / elapsed, "unit": "edges/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: {cfg['desc']}, L=3, synthetic S({cfg['U']},{cfg['I']},E,span) "
                                   f"stream of {(W + K) * Bg} edges, decoder-level unit (2 readouts + update per edge)",
                       "batch": Bg, "dim": d, "num_layer": L, "nodes": N, "parallelism": par},
            "roofline": roof, "cpu_baseline": cpu,
        }
        if row_info is not None:
            line["col_sharded"] = row_info
        if shard == "rows":
            line["config"]["table_bytes_per_gpu"] = row_bytes
            line["timed_regions"] = rows_info.get("regions")
            line["config"]["schedule"] = rows_info.get("schedule")
        if dropin is not None:
            line["dropin"] = dropin
        if shard == "single" and len(regions) > 1:
            line["timed_regions"] = {"n": len(regions), "value_is": "median", "wall_us": [r_ * 1e6 for r_ in regions],
                                     "first_region": {"value": K * Bg / regions[0], "ms_per_step": regions[0] * 1e3 / K},
                                     "call": "rp.prepare_stream(...) before the clock, the prepared call inside it",
                                     "plain_call": {"value": K * Bg / float(np.median(plain_regions)),
                                                    "wall_us": [r_ * 1e6 for r_ in plain_regions],
                                                    "what": "the same steps through rp.run_stream (its argument checks inside the "
                                                            "region), median of five regions timed behind the nine"},
                                     "with_host_copy": {"value": K * Bg / float(np.median(copy_regions)),
                                                        "wall_us": [r_ * 1e6 for r_ in copy_regions],
                                                        "what": "the PCIe-inclusive rate: src / dst / neg / t of the timed steps copied from "
                                                                "pinned host memory inside the region (four asynchronous copies, "
                                                                f"{K * Bg * 32} bytes), then the same prepared call; median of five regions"}}
        if extra:
            line.update(extra)
        write_line(json.dumps(line))

    # second leg at N > 1: the column-sharded layout (ablation) on the same workload, behind a watchdog (a collective that
    # never returns must not cost the main line)
    row_info = None
    if shard == "rows" and cols_ok and world > 1 and os.environ.get("TPNET_BENCH_COLS_LEG", "1") != "0":
        import threading
        state = {"done": False}

        def give_up():
            if not state["done"]:
                if rank == 0:
                    emit({"error": "column-sharded leg did not finish within 150 s"})
                os._exit(0)
        wd = threading.Timer(150.0, give_up)
        wd.daemon = True
        try:
            wd.start()
            el_c = cols_leg(K)
            row_info = {"value": K * Bg / el_c, "unit": "edges/s", "steps": K, "ms_per_step": el_c * 1e3 / K,
                        "parallelism": f"ablation: columns sharded over {world} GPUs ({d // world} of {d} per GPU), no per-step "
                                       f"collective; raw Gram entries reduce-scattered per chunk behind the next chunk's kernels"}
        except Exception as ex:                   # noqa: BLE001 -- report, keep the main line
            row_info = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        state["done"] = True
        wd.cancel()

    # kernel-level timing for the roofline object: HIP events on the stream the kernels run on (C side, tpnet_time_stream)
    # around the loop of launches of the dominant kernel, in an extra pass over the SAME batches [W, W + K) as the timed
    # region -- so `roofline` describes the kernel `value` was produced by (k_step below 16 batches, k_wpipe from there).
    roof = None
    extra = {}

    resident = 28.0 * N * d < 256e6
    if rank == 0 and shard == "single":
        sl = slice(W * Bg, (W + K) * Bg)
        rp.reset_random_projections()
        roof = roof_pass(d_src[sl], d_dst[sl], d_neg[sl], d_t[sl], K * B, 0.0, float(t[(W + K) * B - 1]), out_pos, out_neg)
        roof["duration_note"] = ("HIP events on the launch stream around the loop of launches of this kernel / launches: kernel "
                                 "duration + inter-kernel boundary (rocprofv3 per-kernel average of the same command: profiles/); "
                                 f"an extra pass over the {K} batches of the timed region, same schedule"
                                 + ("; the state is cache-resident at this config, so the algorithmic rate can exceed what "
                                    "HBM itself delivers: see traffic / memory_side_frac" if resident else
                                    "; the state is far larger than the 256 MB Infinity Cache: row accesses are HBM misses"))
        try:
            cbw = copy_bandwidth_gbs(dev)
            roof["copy_bandwidth_gbs_measured"] = cbw
            roof["frac_of_measured_copy_bandwidth"] = roof["algorithmic_gbs"] / cbw
        except Exception:
            pass

        # ---- ONE epoch of the config as BASELINE states it (C2: E = 157 474 edges, batches of 1 000, the last one partial):
        # wall clock around run_stream -- planning, pipeline fill and drain, write-back included -- after a reset, as
        # train_link_prediction.py:246-253 starts every epoch.  `cold`: the first epoch of a process (plans the stream);
        # `replay`: a later epoch (same stream, new negatives: the plan of the update is replayed, tpnet_run_stream_tagged).
        def epoch_figures():
            Ee = int(cfg["E"])
            if Ee * (2 * 64 * 4 + 40) > 40e9:
                return None
            e_src, e_dst, e_t, _ = synthetic_stream(cfg["U"], cfg["I"], Ee, cfg["span"], seed=0)
            negs = [synthetic_negatives(cfg["U"], N, Ee, B, seed=s_) for s_ in (1, 2, 3, 4)]
            g_src, g_dst, g_t = to_dev(e_src), to_dev(e_dst), to_dev(e_t)
            g_negs = [to_dev(x) for x in negs]
            o_p = torch.empty((Ee, out_pos.shape[1]), dtype=torch.float32, device=dev)
            o_n = torch.empty_like(o_p)
            o_p.zero_(); o_n.zero_()
            rp._workspace(Ee, B, stream=True).zero_()
            t_end = float(e_t[-1])

            def one(neg_dev, replay):
                rp.reset_random_projections()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                rp.run_stream(g_src, g_dst, neg_dev, g_t, B, out_pos=o_p, out_neg=o_n, t_end=t_end, replay=replay)
                torch.cuda.synchronize()
                return time.perf_counter() - t0, rp.last_stream_replayed
            one(g_negs[0], False)                                      # first use of this shape in the process: untimed
            cold = [one(g_negs[i % 4], False)[0] for i in range(1, 6)]
            one(g_negs[0], None)                                       # leaves the plan behind
            rep = [one(g_negs[i % 4], None) for i in range(1, 6)]
            # SURVEY section 8(d) asks for the figures with and without rp.mlp: the same epoch followed by self.mlp (fp32 class,
            # tpnet_mlp64_f32) over its 2 E feature rows -- what get_pair_wise_feature returns for every (src, dst) and (src, neg)
            with_mlp = None
            if rp.pair_wise_feature_dim == 64:
                from tpnet_amd import fused_feature as FF

                def one_mlp(neg_dev, replay):
                    rp.reset_random_projections()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    rp.run_stream(g_src, g_dst, neg_dev, g_t, B, out_pos=o_p, out_neg=o_n, t_end=t_end, replay=replay)
                    with torch.no_grad():
                        y1 = FF.mlp_f32(rp.mlp, o_p)
                        y2 = FF.mlp_f32(rp.mlp, o_n)
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t0) if (y1 is not None and y2 is not None) else None
                if one_mlp(g_negs[0], False) is not None:
                    cm = [one_mlp(g_negs[i % 4], False) for i in range(1, 4)]
                    one_mlp(g_negs[0], None)
                    rm = [one_mlp(g_negs[i % 4], None) for i in range(1, 4)]
                    with_mlp = {"what": "the same epoch + self.mlp (fp32 class, tpnet_mlp64_f32) over its 2 E feature rows, wall clock, median of 3",
                                "cold": {"wall_us": float(np.median(cm)) * 1e6, "value": Ee / float(np.median(cm))},
                                "replay": {"wall_us": float(np.median(rm)) * 1e6, "value": Ee / float(np.median(rm))}}
            rp.check_device_errors()
            # the dominant kernel of the epoch, timed like `roofline` (events around the loop of pipeline launches)
            rp.reset_random_projections()
            er = roof_pass(g_src, g_dst, g_negs[0], g_t, Ee, 0.0, t_end, o_p, o_n)
            med = lambda xs: float(np.median(xs))
            bpe = bytes_per_edge(d, L)
            out = {"edges": Ee, "batch": B, "batches": (Ee + B - 1) // B,
                   "what": "wall clock around ONE run_stream call over the config's whole stream after a reset (planning, "
                           "pipeline fill / drain and write-back included), median of 5",
                   "cold": {"wall_us": med(cold) * 1e6, "value": Ee / med(cold), "unit": "edges/s",
                            "end_to_end_frac": bpe * Ee / med(cold) / 1e9 / HBM_PEAK_GBS},
                   "replay": {"wall_us": med([r[0] for r in rep]) * 1e6, "value": Ee / med([r[0] for r in rep]), "unit": "edges/s",
                              "end_to_end_frac": bpe * Ee / med([r[0] for r in rep]) / 1e9 / HBM_PEAK_GBS,
                              "plan_replayed": bool(all(r[1] for r in rep)),
                              "what": "a later epoch: same stream, new negatives, the update's plan replayed"},
                   "with_mlp": with_mlp,
                   "kernel": {k: er[k] for k in ("kernel_short", "bound", "frac", "achieved", "algorithmic_frac", "launches", "edges_per_launch",
                                                 "avg_launch_period_us", "memory_side_frac", "traffic", "stream_ms_events",
                                                 "end_to_end_frac")}}
            del o_p, o_n
            return out
        if os.environ.get("TPNET_BENCH_EPOCH", "1") != "0" and B <= 4096:
            from tpnet_amd.stream import synthetic_stream, synthetic_negatives
            extra["epoch"] = epoch_figures()

        # ---- the long-stream regime: the same kernel schedule over ROOF_STEPS batches of the bench stream (what a stream of
        # millions of edges runs at); reported beside, never instead of, the figures above
        if Kr > K and B <= 4096:
            slr = slice(W * Bg, (W + Kr) * Bg)
            o_pos = torch.empty((Kr * Bg, out_pos.shape[1]), dtype=torch.float32, device=dev)
            o_neg = torch.empty_like(o_pos)
            args_r = (d_src[slr], d_dst[slr], d_neg[slr], d_t[slr], Kr * B, 0.0, float(t[(W + Kr) * B - 1]), o_pos, o_neg)
            rp.reset_random_projections()
            roof_pass(*args_r)                 # first use of this shape in the process: untimed
            rp.reset_random_projections()
            lr = roof_pass(*args_r)
            lr["value"] = Kr * B / (lr["stream_ms_events"] * 1e-3)
            lr["unit"] = "edges/s (HIP events around the whole call: planning and write-back included)"
            # the longest stream that is ONE chunk (256 windows of 24 batches / 16 GiB of version log), run again after a reset: the plan is replayed
            kwin = max(2, min(64, 24576 // B))                       # batches per window of a long stream
            Kc = min(Kr, min(256, ((16 << 30) // (2 * L * d * 4)) // (kwin * B)) * kwin)     # 256 windows, or what 16 GiB of version log hold
            slc = slice(W * Bg, (W + Kc) * Bg)
            bpe = bytes_per_edge(d, L)

            def one_chunk(replay):
                rp.reset_random_projections()
                torch.cuda.synchronize()
                t0_ = time.perf_counter()
                rp.run_stream(d_src[slc], d_dst[slc], d_neg[slc], d_t[slc], B, out_pos=o_pos[:Kc * B], out_neg=o_neg[:Kc * B],
                              t_end=float(t[(W + Kc) * B - 1]), replay=replay)
                torch.cuda.synchronize()
                return time.perf_counter() - t0_, rp.last_stream_replayed
            one_chunk(False)
            cold = float(np.median([one_chunk(False)[0] for _ in range(3)]))
            one_chunk(None)
            reps = [one_chunk(None) for _ in range(3)]
            rep_t = float(np.median([r_[0] for r_ in reps]))
            lr["one_chunk"] = {"batches": Kc, "what": "wall clock around run_stream on the longest stream that is one chunk, after a reset",
                               "cold": {"value": Kc * B / cold, "end_to_end_frac": bpe * Kc * B / cold / 1e9 / HBM_PEAK_GBS},
                               "replay": {"value": Kc * B / rep_t, "end_to_end_frac": bpe * Kc * B / rep_t / 1e9 / HBM_PEAK_GBS,
                                          "plan_replayed": bool(all(r_[1] for r_ in reps))}}
            # the same stream as SEVERAL chunks (the version log capped at half of it; without a cap: streams beyond 16 GiB of log or
            # 256 windows): every chunk's plan keeps a region of its own in front of the one log, so the replay covers them all
            half = (Kc // 2 // kwin) * kwin
            if half >= kwin and half < Kc:
                rp._eng["ws"] = None                                # (the workspace of the one-chunk runs would hold the whole log)
                rp._drop_plan()
                rp.stream_log_cap_bytes = half * B * 2 * L * d * 4
                one_chunk(False)
                cold2 = float(np.median([one_chunk(False)[0] for _ in range(3)]))
                one_chunk(None)
                reps2 = [one_chunk(None) for _ in range(3)]
                rep2_t = float(np.median([r_[0] for r_ in reps2]))
                lr["chunks"] = {"batches": Kc, "chunks": -(-Kc // half), "log_cap_batches": half,
                                "workspace_gb": rp._eng["ws"].numel() / 1e9,
                                "cold": {"value": Kc * B / cold2, "end_to_end_frac": bpe * Kc * B / cold2 / 1e9 / HBM_PEAK_GBS},
                                "replay": {"value": Kc * B / rep2_t, "end_to_end_frac": bpe * Kc * B / rep2_t / 1e9 / HBM_PEAK_GBS,
                                           "plan_replayed": bool(all(r_[1] for r_ in reps2))}}
                rp.stream_log_cap_bytes = None
                rp._eng["ws"] = None
                rp._drop_plan()
            extra["long_stream"] = lr
            del o_pos, o_neg

    # ---- N > 1 (or a forced one-rank run of that path): who took part, the kernel / exchange figures, and C4's law through the
    # same runner -- the config BASELINE.json names for 8 GPUs (10 000 001 rows x d = 256 cut over the ranks, 10 000 edges per
    # GPU and step) -- beside the C2 `value`
    if dist is not None:
        ids = torch.tensor([rank, dev.index if dev.index is not None else 0, os.getpid()], dtype=torch.int64, device=dev)
        got = [torch.empty_like(ids) for _ in range(world)]
        dist.all_gather(got, ids)
        seen = sorted({int(g_[0]) for g_ in got})
        extra["ranks_seen"] = len(seen)
        extra["devices_seen"] = [int(g_[1]) for g_ in got]
        extra["distinct_processes"] = len({int(g_[2]) for g_ in got})
        extra["backend"] = backend
    if shard == "rows":
        roof = rows_info.get("roof")
        if os.environ.get("TPNET_BENCH_C4_LEG", "1") != "0" and (world > 1 or os.environ.get("TPNET_BENCH_C4_LEG") == "1"):
            try:
                c4 = CONFIGS["C4"]
                B4 = int(os.environ.get("TPNET_BENCH_C4_BATCH", c4["B"]))
                Bg4 = B4 * world
                cfg4 = dict(c4, B=Bg4)
                s4, d4, n4, t4, N4 = make_workload(cfg4, W + K, 0)
                arrs4 = tuple(to_dev(x) for x in (s4, d4, n4, t4))
                torch.cuda.synchronize()
                el4, tb4, info4 = sharded_rows(N4, c4["d"], c4["lam"], c4["E"], Bg4, arrs4, t4, K, draw_on_device=True)
                extra["c4_rows"] = {"value": K * Bg4 / el4, "unit": "edges/s", "steps": K, "warmup": W, "ms_per_step": el4 * 1e3 / K,
                                    "config": {"workload": f"C4: {c4['desc']}, L=3, S(5000000,5000000,E,span) stream of {(W + K) * Bg4} edges",
                                               "nodes": N4, "dim": c4["d"], "batch": Bg4, "batch_per_gpu": B4,
                                               "rows_per_gpu": (N4 + world - 1) // world, "halo_rows": 3 * Bg4,
                                               "table_bytes_per_gpu": tb4},
                                    "roofline": info4.get("roof"), "timed_regions": info4.get("regions")}
                del arrs4
            except Exception as ex:               # noqa: BLE001 -- a secondary leg must not cost the main line
                extra["c4_rows"] = {"error": f"{type(ex).__name__}: {ex}"[:400]}
    if dist is not None:
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        dist = None
    if rank == 0:
        cpu = dropin = None
        if shard == "single" and not args.no_dropin:
            dropin = dropin_rate(cfg, rp, src, dst, neg, t)
        if not args.no_cpu_baseline:
            if shard == "single":
                cpu = cpu_baseline(cfg, src, dst, neg, t, N, P0.numpy())
            else:
                # rank 0's host cores, after the collectives are done: a bounded sample of the SAME (global-batch) workload
                torch.manual_seed(0)
                cpu = cpu_baseline(cfg_run, src, dst, neg, t, N, torch.normal(0, 1 / np.sqrt(d), (N, d)).numpy(), reps=1, quick=True)
        emit(row_info, roof, cpu, dropin, extra)


if __name__ == "__main__":
    main()
