#!/usr/bin/env python3
"""Developer tool: HBM traffic per launch of the rollout kernels, measured on bench.py's OWN rollout.

On the GPU box (two passes: FETCH_SIZE and WRITE_SIZE do not fit one; --pmc with --kernel-trace only, as
MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes):

  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o run -- \
      python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --congested-window 0 --no-kernel-timing
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o run -- \
      python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --congested-window 0 --no-kernel-timing
  python3 tools/pmc_bench.py gpurun_out/pmc_f gpurun_out/pmc_w --out profiles/r02_pmc_traffic.json

Reduction: for every rollout kernel the launches of the LAST iteration's frames >= --first-frame (default 200: the
episode has filled up; the first frames after a reset move almost nobody) are averaged. Units and the gfx950 correction
follow the guide: the counters are KiB, FETCH_SIZE reports half of the bytes of a wide coalesced read (calibrated on this
pool with tools/pmc_calibrate.py: a 256 MiB copy reads FETCH_SIZE = 128 MiB, WRITE_SIZE = 256 MiB), so
hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.
"""
import argparse
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tarl-simulator_amd")]

# kernel -> launches per iteration of T frames
ROLLOUT_KERNELS = {
    "k_fused_direction": lambda T: T,
    "k_fused_rows": lambda T: T,
    "k_fused_insert": lambda T: T,
}


def per_kernel(dirname, counter):
    paths = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
    if not paths:
        raise SystemExit(f"no *counter_collection.csv under {dirname}")
    rows = []
    for p in paths:
        rows += [r for r in csv.DictReader(open(p)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    per = {}
    for r in rows:
        m = re.search(r"\b(k_[A-Za-z0-9_]+)", r["Kernel_Name"])     # template kernels: "void k_fused_rows<4>(...)"
        name = m.group(1) if m else r["Kernel_Name"].split("(")[0].split()[-1]
        if name == "k_fused_insert2":      # several environments per wave: the same launch slot of a frame
            name = "k_fused_insert"
        per.setdefault(name, []).append(float(r["Counter_Value"]))
    return per


def window(vals, per_iter, T, first_frame):
    """The launches of the last iteration that belong to frames >= first_frame."""
    last = vals[-per_iter:]
    return last[first_frame:] if first_frame < len(last) else last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--edges", type=int, default=10000)
    ap.add_argument("--agents", type=int, default=16384)
    ap.add_argument("--envs", type=int, default=16384)
    ap.add_argument("--rollout-steps", type=int, default=256)
    ap.add_argument("--first-frame", type=int, default=200)
    ap.add_argument("--note", type=str, default="")
    ap.add_argument("--out", type=str, default=None)
    a = ap.parse_args()
    from tarl_hip.ops import FUSED_LAYOUT
    f = per_kernel(a.fetch_dir, "FETCH_SIZE")
    w = per_kernel(a.write_dir, "WRITE_SIZE")
    T = a.rollout_steps
    out = {"note": ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) of "
                    "`python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --congested-window 0 --no-kernel-timing`; counters are KiB; "
                    "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B); "
                    f"mean over the last iteration's frames >= {a.first_frame}. " + a.note).strip(),
           "config": {"edges": a.edges, "agents": a.agents, "envs": a.envs, "rollout_steps": T},
           "layout": FUSED_LAYOUT, "first_frame": a.first_frame, "kernels": {}}
    for k, per_iter in ROLLOUT_KERNELS.items():
        if k not in f or k not in w:
            print(f"warning: {k} not in the traces ({sorted(f)[:12]} ...)", file=sys.stderr)
            continue
        fv, wv = window(f[k], per_iter(T), T, a.first_frame), window(w[k], per_iter(T), T, a.first_frame)
        fm, wm = sum(fv) / len(fv), sum(wv) / len(wv)
        out["kernels"][k] = {"FETCH_SIZE_KiB": fm, "WRITE_SIZE_KiB": wm, "launches_averaged": len(fv),
                             "launches_total": len(f[k]), "hbm_bytes_per_launch": 2 * fm * 1024 + wm * 1024}
    txt = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
