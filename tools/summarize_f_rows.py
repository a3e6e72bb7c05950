#!/usr/bin/env python3
"""Summarise gpurun_out/prof_f (tools/profile_f_rows.sh) into profiles/<tag>_f_rows.md."""
import csv, glob, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = [f"# rocprofv3 evidence for the rows either side of the path (f-1, f-2), {tag}", "",
       "`tools/profile_f_rows.sh`: kernel trace + stats of `tools/encoder_readout.py`, `feature_rate.py`, `mlp_rate.py`, `decoder_rate.py`; FETCH_SIZE of the encoder readout kernels in a separate `--pmc` pass (KiB; doubled for gfx950 wide reads as MI355X_MICROARCH.md prescribes).", ""]
def stats(tool, keep):
    f = sorted(glob.glob(f"gpurun_out/prof_f/{tool}/*/*_kernel_stats.csv"))[-1]
    return [r for r in csv.DictReader(open(f)) if any(k in r["Name"] for k in keep)]
def short(name):
    n = name.split("(")[0]
    if n.startswith("_ZN5tpnet19k_pair_feature_bf16"):
        t = n[len("_ZN5tpnet19k_pair_feature_bf16I"):]
        v = [x[2:].rstrip("E") for x in t.split("E")[:5]]
        return f"tpnet::k_pair_feature_bf16<{v[0]}, {v[1]}, {v[2]}, exact fit, {'fp32' if 'Lb1ELb1' in n else 'bf16'} matrix cores>"
    return n[:80]
out += ["## f-2: encoder readout at C3's shape (800 000 pairs = 20 000 rows x K = 20 neighbours x 2 anchors, d = 256) and C1 / C2", "",
        "| kernel (template = lanes, vectors, W, L, exact fit) | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
for r in stats("encoder_readout", ["k_pair_gram"]):
    out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
f = sorted(glob.glob("gpurun_out/prof_f/encoder_fetch/*/*_counter_collection.csv"))[-1]
acc = {}
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if "k_pair_gram" in n and r["Counter_Name"] == "FETCH_SIZE":
        acc.setdefault(n.split("(")[0], []).append(float(r["Counter_Value"]))
out += ["", "FETCH_SIZE per launch (memory side; the C3-shape launches are the large ones):", "",
        "| kernel | launches | max per launch: KiB -> MB after the x2 correction |", "|---|---|---|"]
for k, v in sorted(acc.items()):
    out.append(f"| `{k[:70]}` | {len(v)} | {max(v):.0f} KiB -> {max(v) * 2 * 1024 / 1e6:.1f} MB |")
g = {k.split('<')[0].split('::')[-1]: max(v) * 2 * 1024 / 1e9 for k, v in acc.items() if "<32, 2" in k}
out += ["", f"(800 000 pairs at d=256: {g.get('k_pair_gram', 0):.2f} GB fetched per launch by the generic kernel, {g.get('k_pair_gram_shared', 0):.2f} GB by the "
        f"shared-first-node kernel, **{g.get('k_pair_gram_anchored', 0):.2f} GB by the anchored kernel** -- it issues 4 + 8/K row loads per (neighbour, src, dst) unit against 12 and 16.)", ""]
out += ["## f-1: readout + self.mlp (`tools/feature_rate.py`: C2 / C3 / C5 shapes, random pairs)", "",
        "| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
for r in stats("feature_rate", ["k_pair_feature", "k_mlp64", "k_pair_gram<"]):
    out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
out += ["", "## f-1: the dense layers alone (`tools/mlp_rate.py`: n = 2 000 .. 800 000, forward and backward; `tools/decoder_rate.py`)", "",
        "| kernel | calls | avg us | min us | max us |", "|---|---|---|---|---|"]
for tool, keep in (("mlp_rate", ["k_mlp64"]), ("decoder_rate", ["k_decoder"])):
    for r in stats(tool, keep):
        out.append(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
for t in ("encoder_readout", "feature_rate", "mlp_rate", "decoder_rate"):
    out += ["", f"### `tools/{t}.py` (under the profiler)", "", "```"]
    out += [l.rstrip() for l in open(f"gpurun_out/prof_f.{t}.log") if l[:1] in "Cnb" and " us" in l][:14]
    out += ["```"]
open(f"profiles/{tag}_f_rows.md", "w").write("\n".join(out) + "\n")
print("\n".join(out)[-2500:])
