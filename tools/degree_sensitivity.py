#!/usr/bin/env python3
"""Developer tool: is the launch tuning fitted to ONE degree law?  Rate of the device-resident stream at
C2's shape (d=128, B=1000) on streams with other degree laws -- S's exponents (users, items): (2,3) = the bench stream,
(1.5,4) heavier item hubs, (3,5) very heavy hubs, (1,1) uniform -- with the default launch parameters and with each
developer override, one process per setting (the overrides are read once per process).
    python tools/degree_sensitivity.py            # the table
    python tools/degree_sensitivity.py one PU PI SCHED   # (internal) one measurement, prints us per batch"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LAWS = [(2.0, 3.0), (1.5, 4.0), (3.0, 5.0), (1.0, 1.0)]


def one(pu, pi, sched):
    import numpy as np, torch
    import tpnet_amd
    from tpnet_amd.stream import CONFIGS, synthetic_stream, synthetic_negatives
    c = CONFIGS["C2"]; B = c["B"]
    nb = 2000 if sched == "windowed" else 400
    E = nb * B
    src, dst, t, N = synthetic_stream(c["U"], c["I"], E, c["span"] * E / c["E"], 0, pu, pi)
    neg = synthetic_negatives(c["U"], N, E, B, 1)
    dev = torch.device("cuda:0")
    rp = tpnet_amd.RandomProjectionModule(node_num=N, edge_num=c["E"], dim_factor=10, num_layer=3, time_decay_weight=c["lam"],
            device="cuda:0", use_matrix=False, beginning_time=np.float64(0.0), not_scale=False, enforce_dim=c["d"]).to(dev)
    D = lambda x: torch.from_numpy(x).to(dev)
    ds, dd, dn, dt = D(src), D(dst), D(neg), D(t)
    op = torch.empty((E, 64), device=dev); on = torch.empty((E, 64), device=dev)
    best = 1e9
    for r in range(4):
        rp.reset_random_projections()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rp.run_stream(ds, dd, dn, dt, B, out_pos=op, out_neg=on, t_end=float(t[-1]), schedule=sched)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        if r:
            best = min(best, el / nb * 1e6)
    # hub statistics of the stream: the largest number of contributions one node receives in a batch (median over batches)
    top = [int(np.bincount(np.concatenate([src[b * B:(b + 1) * B], dst[b * B:(b + 1) * B]])).max()) for b in range(0, nb, 37)]
    print(f"RESULT {best:.3f} {int(np.median(top))}")


def run(pu, pi, sched, env):
    e = dict(os.environ, HIP_FORCE_DEV_KERNARG="1", **env)
    out = subprocess.run([sys.executable, __file__, "one", str(pu), str(pi), sched], env=e, capture_output=True, text=True)
    for line in out.stdout.splitlines():
        if line.startswith("RESULT"):
            _, us, top = line.split()
            return float(us), int(top)
    raise RuntimeError(out.stdout[-500:] + out.stderr[-500:])


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "one":
        one(float(sys.argv[2]), float(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    variants = {
        "batch": [("default", {})] + [(f"HB={h}", {"TPNET_DEV_HEAVY_BLOCKS": str(h)}) for h in (int(x) for x in os.environ.get("SWEEP_HB", "24,64,96").split(","))]
                 + [(f"thr={x}", {"TPNET_DEV_HEAVY_THRESHOLD": str(x)}) for x in (4, 16, 32)],
        "windowed": [("default", {})] + [(f"WHB={h}", {"TPNET_DEV_WIN_HB": str(h)}) for h in (int(x) for x in os.environ.get("SWEEP_WHB", "64,256").split(","))]
                    + [(f"wthr={x}", {"TPNET_DEV_WIN_HEAVY": str(x)}) for x in (int(x) for x in os.environ.get("SWEEP_WTHR", "32,128").split(","))]
                    + [(f"win={k}", {"TPNET_DEV_WINDOW": str(k)}) for k in (int(x) for x in os.environ.get("SWEEP_WIN", "8,32").split(","))],
    }
    for sched in (sys.argv[1:] or ["windowed", "batch"]):
        print(f"## schedule = {sched} (us per batch of 1000 edges; best of 3 after a warm-up pass)")
        print("| degree law (users, items) | hub: median max contributions per batch | " + " | ".join(n for n, _ in variants[sched])
              + " | default vs best |")
        print("|---|---|" + "---|" * (len(variants[sched]) + 1))
        for pu, pi in LAWS:
            res = [run(pu, pi, sched, env) for _, env in variants[sched]]
            us = [r[0] for r in res]
            print(f"| ({pu}, {pi}) | {res[0][1]} | " + " | ".join(f"{u:.2f}" for u in us) + f" | {100 * (us[0] / min(us) - 1):+.1f} % |",
                  flush=True)
