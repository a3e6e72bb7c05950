#!/usr/bin/env python3
"""Summarise the kernel timeline of bench.py's device-resident encoder-level loop from a rocprofv3 kernel trace
(tools/encoder_trace.sh): per batch of the no-grad loop the kernels, their durations and the gaps between them (medians)."""
import csv, glob, os, sys, statistics as st
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = sorted(glob.glob(os.path.join(R, "gpurun_out", "enc_trace", "*", "*_kernel_trace.csv")), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void ", "").replace("tpnet::", "").split("(")[0][:40]
# batches of the no-grad loop: sample, fused, sample, fused, feature, feature, plan, step with nothing else in between
pat = ["k_encoder_sample", "k_encoder_", "k_encoder_sample", "k_encoder_", "k_pair_feature", "k_pair_feature", "k_plan_one", "k_step"]
names = [short(r["Kernel_Name"]) for r in rows]
hits = []
for i in range(len(rows) - len(pat)):
    if all(p in names[i + j] for j, p in enumerate(pat)):
        hits.append(i)
per = []
for a, b in zip(hits[:-1], hits[1:]):
    if b - a == len(pat):
        t0 = int(rows[a]["Start_Timestamp"])
        per.append([(int(rows[a + j]["Start_Timestamp"]) - t0, int(rows[a + j]["End_Timestamp"]) - t0) for j in range(len(pat))] + [(int(rows[b]["Start_Timestamp"]) - t0, 0)])
print(f"{len(per)} consecutive batches of the no-grad loop")
prev_end = 0
for j in range(len(pat)):
    s = st.median(p[j][0] for p in per); e = st.median(p[j][1] for p in per)
    print(f"{names[hits[0] + j]:42s} start {s / 1e3:7.1f}  dur {(e - s) / 1e3:6.1f}  gap before {(s - prev_end) / 1e3:5.1f}")
    prev_end = e
print(f"next batch starts at {st.median(p[-1][0] for p in per) / 1e3:.1f} us")
