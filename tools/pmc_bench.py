#!/usr/bin/env python3
"""Developer tool: HBM traffic per launch of the rollout kernels, measured on bench.py's OWN rollout.

On the GPU box (two passes: FETCH_SIZE and WRITE_SIZE do not fit one; --pmc with --kernel-trace only, as
MI355X_MICROARCH.md's HBM / rocprofv3 section prescribes):

  cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_f -o run -- \
      python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --congested-window 0 --policy-envs 0 --no-kernel-timing
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_w -o run -- \
      python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --congested-window 0 --policy-envs 0 --no-kernel-timing
  python3 tools/pmc_bench.py gpurun_out/pmc_f gpurun_out/pmc_w [more pass dirs] --out profiles/r05_pmc_traffic.json
  (tools/r05_pmc.sh runs the passes for the default line, the congested regime — --departure-window 600 — and config 5)

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


def per_kernel_all(dirnames):
    """{counter: {kernel: [values in dispatch order]}} over every *counter_collection.csv under the directories."""
    out = {}
    for dirname in dirnames:
        paths = glob.glob(os.path.join(dirname, "**", "*counter_collection.csv"), recursive=True)
        if not paths:
            raise SystemExit(f"no *counter_collection.csv under {dirname}")
        rows = []
        for p in paths:
            rows += list(csv.DictReader(open(p)))
        rows.sort(key=lambda r: int(r["Dispatch_Id"]))
        for r in rows:
            m = re.search(r"\b(k_[A-Za-z0-9_]+)", r["Kernel_Name"])     # template kernels: "void k_fused_rows<4>(...)"
            name = m.group(1) if m else r["Kernel_Name"].split("(")[0].split()[-1]
            if name == "k_fused_insert2":      # several environments per wave: the same launch slot of a frame
                name = "k_fused_insert"
            out.setdefault(r["Counter_Name"], {}).setdefault(name, []).append(float(r["Counter_Value"]))
    return out


def window(vals, per_iter, T, first_frame):
    """The launches of the last iteration that belong to frames >= first_frame."""
    last = vals[-per_iter:]
    return last[first_frame:] if first_frame < len(last) else last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+", help="rocprofv3 output directories, one per --pmc pass (FETCH_SIZE and WRITE_SIZE "
                                            "are required; any further counters are recorded beside them)")
    ap.add_argument("--edges", type=int, default=10000)
    ap.add_argument("--agents", type=int, default=16384)
    ap.add_argument("--envs", type=int, default=32768)
    ap.add_argument("--rollout-steps", type=int, default=256)
    ap.add_argument("--departure-window", type=int, default=0, help="the bench's --departure-window (0 = the default line)")
    ap.add_argument("--first-frame", type=int, default=200)
    ap.add_argument("--note", type=str, default="")
    ap.add_argument("--out", type=str, default=None)
    a = ap.parse_args()
    from tarl_hip.ops import FUSED_LAYOUT, frame_kernel_source_hash
    allc = per_kernel_all(a.dirs)
    if "FETCH_SIZE" not in allc or "WRITE_SIZE" not in allc:
        raise SystemExit("FETCH_SIZE and WRITE_SIZE passes are required")
    T = a.rollout_steps
    cfg = {"edges": a.edges, "agents": a.agents, "envs": a.envs, "rollout_steps": T}
    cmd = ("python3 bench.py --steps 1 --warmup 1 --cpu-seconds 0 --congested-window 0 --policy-envs 0 --config5-envs 0 "
           "--update-epochs 0 --no-kernel-timing")
    if (a.edges, a.agents, a.envs) != (10000, 16384, 32768):
        cmd += f" --edges {a.edges} --agents {a.agents} --envs {a.envs}"
    if a.departure_window:
        cfg["departure_window"] = a.departure_window
        cmd += f" --departure-window {a.departure_window}"
    out = {"note": (f"rocprofv3 --pmc passes (one counter set each, --kernel-trace only) of `{cmd}`; FETCH_SIZE / WRITE_SIZE are KiB. "
                    "hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024. Calibration (profiles/r03_pmc_calibration.txt, tools/pmc_cal.hip): "
                    "every read request of gfx950 is 128 B and FETCH_SIZE tallies it at 64 B — measured 0.5000 for 1-, 4-, 8- and "
                    "16-byte-per-lane coalesced loads, the env-minor 8-byte gather and scattered 12-byte triples alike (the latter: one "
                    "128-B request per triple) — so the factor 2 holds for every load these kernels issue; WRITE_SIZE is exact for stores "
                    "that fill 64-byte lines and tallies a partial-line store at 32 B per request. "
                    f"Mean over the last iteration's frames >= {a.first_frame}. " + a.note).strip(),
           "config": cfg, "layout": FUSED_LAYOUT, "source_sha16": frame_kernel_source_hash(),
           "source_sha16_of": "tarl-simulator_amd/csrc/fused.hip + fused_common.h (bench.py ignores a record taken on other code)",
           "first_frame": a.first_frame, "kernels": {}}
    for k, per_iter in ROLLOUT_KERNELS.items():
        if k not in allc["FETCH_SIZE"] or k not in allc["WRITE_SIZE"]:
            print(f"warning: {k} not in the traces ({sorted(allc['FETCH_SIZE'])[:12]} ...)", file=sys.stderr)
            continue
        rec = {"launches_total": len(allc["FETCH_SIZE"][k])}
        for c, per in sorted(allc.items()):
            if k in per:
                v = window(per[k], per_iter(T), T, a.first_frame)
                rec[c + ("_KiB" if c in ("FETCH_SIZE", "WRITE_SIZE") else "")] = sum(v) / len(v)
                rec["launches_averaged"] = len(v)
        fm, wm = rec["FETCH_SIZE_KiB"], rec["WRITE_SIZE_KiB"]
        rec["formula"] = "2*FETCH_SIZE + WRITE_SIZE"
        rec["hbm_bytes_per_launch"] = 2 * fm * 1024 + wm * 1024
        out["kernels"][k] = rec
    txt = json.dumps(out, indent=1)
    if a.out:
        open(a.out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
