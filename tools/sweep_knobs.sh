#!/bin/bash
# Developer tool: bench the fused path under different knobs (each run is its own process: knobs are read once).
# usage: tools/sweep_knobs.sh "TARL_NCHUNK=1 TARL_NCHUNK_DIR=2" "TARL_ROLLOUT_MERGE=1" ...     (BENCH_ARGS adds bench flags)
cd "$(dirname "$0")/.."
for kv in "$@"; do
  env $kv python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --congested-window 0 --policy-envs 0 $BENCH_ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$kv', 'value %.3fM' % (d['value']/1e6), 'ms/iter %.2f' % d['ms_per_step'], 'rows %.1f dir %.1f insert %.1f us' % (d['roofline']['avg_launch_us'], d['roofline_direction']['avg_launch_us'], d['roofline_insert']['avg_launch_us']))"
done
