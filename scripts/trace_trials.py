#!/usr/bin/env python
"""Per LM iteration of a kernel trace (rocprofv3 --kernel-trace csv of bench.py): its length (linearisation start to the next one's), how many
trial steps (k_trial_apply, all streams) it held, and where the main stream waited.  Usage: trace_trials.py <kernel_trace.csv> [n_last]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("mcp::", "").replace("void ", ""), r["Stream_Id"]) for r in rows)
lin = [i for i, k in enumerate(ks) if k[2].startswith("k_linearize")]
main = ks[lin[-1]][3]
by = defaultdict(list)
for a, b in zip(lin[-nlast - 1:-1], lin[-nlast:]):
    seg = ks[a:b]
    t0 = seg[0][0]
    ntr = sum(1 for k in seg if k[2] == "k_trial_apply")
    msk = [k for k in seg if k[3] == main]
    first_trial_end = next((k[1] for k in msk if k[2].startswith("k_final_sums")), None)
    head_start = next((k[0] for k in msk if k[2].startswith("k_select_pass") or k[2].startswith("k_head")), None)
    by[ntr].append(((ks[b][0] - t0) / 1e3, (head_start - first_trial_end) / 1e3 if first_trial_end and head_start else float("nan")))
for ntr in sorted(by):
    v = by[ntr]
    print("trial steps %d: %2d iterations, length %.1f us (min %.1f, max %.1f), main trial's sums -> head %.1f us" % (
        ntr, len(v), sum(x[0] for x in v) / len(v), min(x[0] for x in v), max(x[0] for x in v), sum(x[1] for x in v) / len(v)))
