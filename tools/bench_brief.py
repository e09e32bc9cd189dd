#!/usr/bin/env python3
"""Developer tool: one-line digest of a bench.py JSON line (stdin): env-steps/s, ms per iteration, the three frame kernels'
live launch times (late frames / all frames), the congested regime, config 5, the update path, the policy lines."""
import json
import sys


def us(r):
    return r["avg_launch_us"], r.get("avg_launch_us_all_frames", r.get("us_all", 0.0))


for line in sys.stdin:
    if not line.startswith('{"metric"'):
        continue
    d = json.loads(line)
    r, dr, ic = us(d["roofline"]), us(d["roofline_direction"]), us(d["roofline_insert"])
    s = (f"{d['value'] / 1e6:7.2f} M env-steps/s  {d['ms_per_step']:7.2f} ms/iter | rows {r[0]:6.1f} ({r[1]:6.1f}) "
         f"dir {dr[0]:6.1f} ({dr[1]:6.1f}) ins {ic[0]:5.1f} ({ic[1]:5.1f}) us")
    c = d.get("congested_regime")
    if c:
        s += f" | congested {c['value'] / 1e6:6.2f} M  {c['ms_per_step']:7.2f} ms/iter"
        if "roofline" in c:
            s += f" rows {us(c['roofline'])[1]:6.1f} dir {us(c['roofline_direction'])[1]:6.1f} ins {us(c['roofline_insert'])[1]:6.1f}"
    c5 = d.get("config5")
    if c5:
        s += f" | c5 {c5['value'] / 1e6:5.2f} M"
        if "roofline" in c5:
            s += f" rows {us(c5['roofline'])[0]:6.1f} dir {us(c5['roofline_direction'])[0]:6.1f} ins {us(c5['roofline_insert'])[0]:6.1f} ps/pair {c5.get('ps_per_pair', 0):.1f}"
    u = d.get("update_path")
    if u:
        s += f" | update {u['value'] / 1e6:5.2f} M ({u['update_frac']:.2f} of the iteration, {u['ms_per_minibatch_step']:.2f} ms/minibatch)"
    p = d.get("state_dependent_policy")
    if p:
        s += " | policy " + " ".join(f"{k} {p[k]['value'] / 1e6:5.2f} M" for k in ("fp32", "fp32_x3", "bf16") if k in p)
        for k in ("fp32_x3", "bf16"):
            rr = p.get(k, {}).get("roofline")
            if rr:
                s += f" mlp[{k}] {rr['avg_launch_us']:.0f} us"
    print(s)
