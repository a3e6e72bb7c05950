#!/usr/bin/env python3
"""Summarise gpurun_out/<round>_<CFG>_<mode>/{trace,fetch,write} (tools/prof_round.sh; ROUND=r05 by default) into
profiles/<round>_<CFG>.md, profiles/<round>_<CFG>_<mode>_kernel_stats.csv and profiles/<round>_<CFG>_pmc.json (what bench.py reads `roofline.traffic` from: one
entry per (kernel, work per launch), stamped with the commit, the command and `csrc_sha` = bench.csrc_fingerprint() of the kernel sources the
library under the profiler was built from -- bench.py takes an entry only when that equals the sources it runs on).  FETCH_SIZE is doubled (gfx950 counts wide
coalesced reads at half their bytes, MI355X_MICROARCH.md HBM section); both counters are in KiB."""
import csv, glob, json, os, shutil, statistics as st, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tpnet_amd.stream import CONFIGS, bytes_per_edge
ROUND = os.environ.get("ROUND", "r05")

def newest(pat):
    f = sorted(glob.glob(pat), key=os.path.getmtime)
    return f[-1] if f else None

def main(cfg):
    c = CONFIGS[cfg]
    B, d = c["B"], c["d"]
    commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    # the sources the profiled library was built from: recorded by the GPU-side script (sha.txt) when present, else the tree's now
    import bench
    sha = bench.csrc_fingerprint()
    lines = [f"# rocprofv3 summary, {cfg} ({c['desc']}, L=3), round {ROUND[1:].lstrip('0')} (commit {commit}, csrc_sha {sha})", ""]
    entries = []
    for mode, what in (("timed20", "the driver's 20 timed batches (one run_stream call, schedule auto)"),
                       ("epoch", f"ONE epoch of the config's own stream ({c['E']} edges), cold plan"),
                       ("long", "2 048 batches of the stream (the long-stream regime)"),
                       ("batch", "a run of full batches (batches of 10 000 edges take the per-batch kernel)"),
                       ("b1000", "the config's rows in batches of 1 000 edges, 600 batches: wide rows on the windowed schedule")):
        base = os.path.join(ROOT, "gpurun_out", f"{ROUND}_{cfg}_{mode}")
        ks = newest(f"{base}/trace/*/*_kernel_stats.csv")
        tr_f = newest(f"{base}/trace/*/*_kernel_trace.csv")
        if not ks or not tr_f:
            continue
        args = open(f"{base}/args.txt").read().strip() if os.path.exists(f"{base}/args.txt") else ""
        sha_m = open(f"{base}/sha.txt").read().strip() if os.path.exists(f"{base}/sha.txt") else sha   # (as built on the GPU box)
        shutil.copy(ks, os.path.join(ROOT, "profiles", f"{ROUND}_{cfg}_{mode}_kernel_stats.csv"))
        lines += [f"## {mode}: {what}", "", f"`rocprofv3 --kernel-trace --stats -- python3 tools/profile_stream.py {args}`", "",
                  "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
        for r in list(csv.DictReader(open(ks)))[:9]:
            nm = r["Name"].replace("void ", "").replace("tpnet::", "")
            nm = "rocprim " + ("radix onesweep" if "onesweep" in nm else "merge sort" if "merge" in nm else "kernel") if "rocprim" in nm else nm
            lines.append(f"| `{nm[:60]}` | {r['Calls']} | {float(r['AverageNs']):.0f} | {r['MinNs']} | {r['MaxNs']} | {r['Percentage']} |")
        alltr = list(csv.DictReader(open(tr_f)))
        kname = "k_wpipe" if any("k_wpipe" in r["Kernel_Name"] for r in alltr) else "k_step"
        tr = [r for r in alltr if kname in r["Kernel_Name"]]
        dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tr]
        # edges those launches covered: every rep of profile_stream.py runs the whole stream
        toks = args.split()
        reps = int(toks[toks.index("--reps") + 1]) if "--reps" in toks else 3
        if "--edges" in toks:
            E = c["E"] if int(toks[toks.index("--edges") + 1]) < 0 else int(toks[toks.index("--edges") + 1])
        else:
            E = int(toks[toks.index("--batches") + 1]) * (int(toks[toks.index("--batch") + 1]) if "--batch" in toks else B)
        epl = E * reps / len(dur)
        avg = st.mean(dur)
        bpl = bytes_per_edge(d, 3) * epl
        res = {}
        for name in ("fetch", "write"):
            f = newest(f"{base}/{name}/*/*_counter_collection.csv")
            if f:
                vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if kname in r["Kernel_Name"]]
                if vals:
                    res[name] = st.mean(vals)
        fetch_b, write_b = res.get("fetch", 0) * 1024 * 2, res.get("write", 0) * 1024
        traffic = fetch_b + write_b if res else None
        lines += ["", f"{kname}: {len(dur)} launches, avg {avg:.0f} ns, median {st.median(dur):.0f} ns, {epl:.0f} edges per launch; grid "
                  f"{tr[0]['Grid_Size_X']} x {tr[0]['Workgroup_Size_X']}, VGPR {tr[0]['VGPR_Count']}, LDS {tr[0]['LDS_Block_Size']} B",
                  f"algorithmic bytes per launch = {bytes_per_edge(d, 3)} B/edge x {epl:.0f} = {bpl / 1e6:.2f} MB -> {bpl / avg:.0f} GB/s "
                  f"= {bpl / avg / 8000:.3f} of 8 TB/s" + (" (cache-resident state: the memory-side figure below is the HBM-roofline one)"
                                                           if (c["U"] + c["I"] + 1) * 7 * d * 4 < 256e6 else "")]
        if traffic:
            lines += [f"PMC (separate passes): FETCH_SIZE {res.get('fetch', 0):.0f} KiB x2 (gfx950) = {fetch_b / 1e6:.2f} MB, WRITE_SIZE "
                      f"{res.get('write', 0):.0f} KiB = {write_b / 1e6:.2f} MB per launch -> {traffic / 1e6:.2f} MB = {traffic / bpl:.2f} x algorithmic "
                      f"-> {traffic / avg:.0f} GB/s on the memory side = {traffic / avg / 8000:.3f} of 8 TB/s"]
        lines.append("")
        entries.append({"config": cfg, "mode": mode, "kernel": kname, "edges_per_launch": epl, "launches": len(dur), "avg_ns": avg,
                        "median_ns": st.median(dur), "algorithmic_bytes_per_launch": bpl, "fetch_size_kib_avg": res.get("fetch"),
                        "write_size_kib_avg": res.get("write"), "traffic_bytes_per_launch": traffic, "commit": commit, "csrc_sha": sha_m,
                        "command": f"rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 tools/profile_stream.py {args}"})
    json.dump({"config": cfg, "note": "FETCH_SIZE doubled (gfx950 wide-read correction, MI355X_MICROARCH.md HBM section); separate "
               "--pmc passes; one entry per (kernel, work per launch)", "kernels": entries},
              open(os.path.join(ROOT, "profiles", f"{ROUND}_{cfg}_pmc.json"), "w"), indent=1)
    open(os.path.join(ROOT, "profiles", f"{ROUND}_{cfg}.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))

if __name__ == "__main__":
    for cfg in sys.argv[1:] or ["C2"]:
        main(cfg)
