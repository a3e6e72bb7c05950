#!/usr/bin/env python3
"""Aggregate tools/encoder_pmc.sh's rocprofv3 outputs: per kernel the average duration (kernel trace) and the average of every
counter over its launches (the first launch of each kernel dropped)."""
import collections, csv, glob, json, os, sys
O = sys.argv[1]
KEYS = ("k_pair_gram_anchored", "k_pair_gram", "k_mlp64_x3", "k_pair_feature", "k_encoder_mfma", "k_encoder_gram_mfma")
def short(name):
    if "k_encoder_fused" in name:
        return "k_encoder_fused"
    if "k_encoder_gram_mfma" in name:
        return "k_encoder_gram_mfma"
    for k in KEYS:
        if k in name:
            return k
    return None
res = collections.defaultdict(dict)
for f in glob.glob(os.path.join(O, "trace", "*", "*_kernel_trace.csv")):
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in dur.items():
        v = v[1:] or v
        res[k]["avg_us"] = round(sum(v) / len(v), 2)
        res[k]["launches"] = len(v)
for f in glob.glob(os.path.join(O, "pmc*", "*", "*_counter_collection.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    extra = {}
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for col in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                if col in r and r[col] != "":
                    extra.setdefault(k, {})[col] = r[col]
    for k, cs in acc.items():
        for cn, v in cs.items():
            v = v[1:] or v
            res[k][cn] = round(sum(v) / len(v), 1)
        res[k].update(extra.get(k, {}))
print(json.dumps(res, indent=1, sort_keys=True))
