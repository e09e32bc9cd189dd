#!/usr/bin/env python3
"""Developer tool: condense rocprofv3 CSV output into the summaries committed under profiles/.

  python tools/profile_summary.py stats  <dir>/run_kernel_stats.csv  "<header note>"  > profiles/<name>_kernel_stats.txt
  python tools/profile_summary.py pmc    <fetch_dir> <write_dir> "<note>" edges agents envs > profiles/<name>_pmc_traffic.json

PMC reduction follows MI355X_MICROARCH.md's HBM recipe: FETCH_SIZE and WRITE_SIZE from separate passes, units KiB,
gfx950 correction FETCH_SIZE x 2 (calibrated here with tools/pmc_calibrate.py: a 256 MiB copy reads FETCH_SIZE = 128 MiB,
WRITE_SIZE = 256 MiB): hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024. Medians over each kernel's launches after
the first five."""
import csv
import json
import statistics
import sys


def stats(path, note):
    rows = list(csv.DictReader(open(path)))
    print(f"# {note}")
    print("# name | calls | avg_us | total_ms | pct")
    for r in rows[:22]:
        print(f"{r['Name'][:70]:70s} | {int(r['Calls']):6d} | {float(r['AverageNs']) / 1e3:9.2f} | "
              f"{float(r['TotalDurationNs']) / 1e6:9.2f} | {float(r['Percentage']):6.2f}")


def counter_medians(path, name):
    per = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0]
        if k.startswith("k_fused_"):
            per.setdefault(k, []).append(float(r["Counter_Value"]))
    return {k: statistics.median(v[5:] if len(v) > 10 else v) for k, v in per.items()}


def pmc(fdir, wdir, note, edges, agents, envs):
    f = counter_medians(f"{fdir}/run_counter_collection.csv", "FETCH_SIZE")
    w = counter_medians(f"{wdir}/run_counter_collection.csv", "WRITE_SIZE")
    out = {"note": note, "config": {"edges": int(edges), "agents": int(agents), "envs": int(envs)}, "kernels": {}}
    for k in ("k_fused_choice", "k_fused_direction", "k_fused_rows", "k_fused_insert"):
        out["kernels"][k] = {"FETCH_SIZE_KiB": f[k], "WRITE_SIZE_KiB": w[k],
                             "hbm_bytes_per_launch": 2 * f[k] * 1024 + w[k] * 1024}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](*sys.argv[2:])
