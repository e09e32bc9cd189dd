#!/usr/bin/env python3
"""Developer tool: one-line digest of a bench.py JSON line (stdin): env-steps/s, ms per iteration, the three frame kernels'
live launch times (late frames / all frames), the congested regime."""
import json
import sys

for line in sys.stdin:
    if not line.startswith('{"metric"'):
        continue
    d = json.loads(line)
    r, dr, ic = d["roofline"], d["roofline_direction"], d["roofline_insert"]
    s = (f"{d['value'] / 1e6:7.2f} M env-steps/s  {d['ms_per_step']:7.2f} ms/iter | rows {r['avg_launch_us']:6.1f} ({r['avg_launch_us_all_frames']:6.1f}) "
         f"dir {dr['avg_launch_us']:6.1f} ({dr['avg_launch_us_all_frames']:6.1f}) ins {ic['avg_launch_us']:5.1f} ({ic['avg_launch_us_all_frames']:5.1f}) us")
    c = d.get("congested_regime")
    if c:
        s += f" | congested {c['value'] / 1e6:6.2f} M  {c['ms_per_step']:7.2f} ms/iter"
        if "roofline" in c:
            s += (f" rows {c['roofline']['avg_launch_us_all_frames']:6.1f} dir {c['roofline_direction']['avg_launch_us_all_frames']:6.1f} "
                  f"ins {c['roofline_insert']['avg_launch_us_all_frames']:6.1f}")
    p = d.get("state_dependent_policy")
    if p:
        s += f" | policy fp32 {p['fp32']['value'] / 1e6:5.2f} M bf16 {p['bf16']['value'] / 1e6:5.2f} M"
        if "fp32_mfma" in p:
            s += f" fp32-mfma {p['fp32_mfma']['value'] / 1e6:5.2f} M"
        for k in ("fp32", "bf16"):
            r = p[k].get("roofline")
            if r:
                s += f" mlp[{k}] {r['avg_launch_us']:.0f} us"
    print(s)
