#!/usr/bin/env python3
"""Developer tool: per-kernel means of arbitrary rocprofv3 --pmc counters (any *counter_collection.csv under the given
directories), over each kernel's last `--last` launches. Usage: python3 tools/pmc_reduce.py DIR [DIR ...] [--last 56]
[--kernels k_fused_rows,k_fused_direction]"""
import argparse
import csv
import glob
import os
import re
from collections import defaultdict

ap = argparse.ArgumentParser()
ap.add_argument("dirs", nargs="+")
ap.add_argument("--last", type=int, default=56)
ap.add_argument("--kernels", type=str, default="k_fused_rows,k_fused_direction,k_fused_insert_choice")
a = ap.parse_args()
want = a.kernels.split(",")
vals = defaultdict(lambda: defaultdict(list))
for d in a.dirs:
    for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(p)), key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            m = re.search(r"\b(k_[A-Za-z0-9_]+)", r["Kernel_Name"])
            k = m.group(1) if m else r["Kernel_Name"].split("(")[0].split()[-1]
            if k in want:
                vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in want:
    print(k)
    for c, v in sorted(vals[k].items()):
        w = v[-a.last:]
        print(f"  {c:28s} {sum(w) / len(w):16.1f}   (n={len(w)})")
