#!/usr/bin/env python3
"""Developer tool: reduce the rocprofv3 --pmc passes of tools/pmc_cal.hip to "counter per useful byte" per access shape.

  python3 tools/pmc_cal_reduce.py gpurun_out/pmc_cal.out gpurun_out/cal_fetch gpurun_out/cal_write [gpurun_out/cal_rdreq ...]

First argument: the program's stdout (its CAL lines = the known byte counts); the rest: rocprofv3 output directories
(one per counter pass). FETCH_SIZE / WRITE_SIZE are KiB; request counters are counts. Prints one table; the factors
bench.py applies per kernel (tools/pmc_bench.py --fetch-factor / the record's "correction") are read off it."""
import csv
import glob
import os
import re
import sys


def norm(kname: str):
    m = re.search(r"(cal_[a-z0-9_]+)(<(.*?)>)?\(", kname)
    if not m:
        return None
    base, targ = m.group(1), m.group(3)
    if targ is None:
        return base
    if targ.strip().isdigit():
        return f"{base}<{targ.strip()}>"
    if "HIP_vector_type" in targ:
        n = re.search(r",\s*(\d)", targ).group(1)
        return f"{base}<uint{n}>"
    return f"{base}<{targ.strip().replace(' ', '_')}>"


def main():
    cal = {}
    for line in open(sys.argv[1]):
        if line.startswith("CAL "):
            _, name, rd, wr, note = line.rstrip("\n").split(" ", 4)
            cal[name] = (int(rd), int(wr), note)
    counters = {}     # counter -> kernel -> [values]
    for d in sys.argv[2:]:
        for p in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(p)):
                k = norm(r["Kernel_Name"])
                if k is None:
                    continue
                counters.setdefault(r["Counter_Name"], {}).setdefault(k, []).append(float(r["Counter_Value"]))
    names = sorted(counters)
    # kernel durations (us) from the kernel traces of the same passes
    dur = {}
    for d in sys.argv[2:]:
        for p in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(p)):
                k = norm(r["Kernel_Name"])
                if k:
                    dur.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# tools/pmc_cal.hip under rocprofv3 --pmc (one pass per counter set), mean of 3 launches per kernel; MI355X (gfx950)")
    print("# FETCH_SIZE / WRITE_SIZE in bytes (counter KiB * 1024); request counters as counts; ratio = counter bytes / useful bytes")
    for k, (rd, wr, note) in cal.items():
        dk = sorted(dur.get(k, [0.0]))
        med = dk[len(dk) // 2]
        rate = f", median {med:.1f} us = {(rd + wr) / med / 1e6:.2f} TB/s of useful bytes" if med else ""
        print(f"\n{k}: useful read {rd} B, useful write {wr} B — {note}{rate}")
        for c in names:
            v = counters[c].get(k)
            if not v:
                continue
            mean = sum(v) / len(v)
            if c in ("FETCH_SIZE", "WRITE_SIZE"):
                byts = mean * 1024.0
                use = rd if c == "FETCH_SIZE" else wr
                ratio = f"{byts / use:.4f} of the useful bytes  => correction x{use / byts:.3f}" if use else "(no useful bytes on this side)"
                print(f"  {c:28s} {byts:16.0f} B   {ratio}")
            else:
                use = rd if "RD" in c else wr
                per = f"{use / mean:.2f} useful B per request" if use and mean else ""
                print(f"  {c:28s} {mean:16.0f}     {per}")


if __name__ == "__main__":
    main()
