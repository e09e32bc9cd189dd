#!/bin/bash
# Developer tool: clocks and socket power while the bench's rollout runs (read-only rocm-smi queries).
cd "$(dirname "$0")/.."
python bench.py --steps 200 --warmup 1 --cpu-seconds 0 --congested-window 0 --policy-envs 0 --no-kernel-timing > /tmp/pp_bench.log 2>&1 &
BP=$!
sleep 9
for i in $(seq 12); do
  rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ' ' | sed 's/=\+//g'
  echo
  sleep 0.5
done
wait $BP
python tools/bench_brief.py < /tmp/pp_bench.log
echo "idle:"
sleep 3
rocm-smi --showpower --showclocks 2>/dev/null | grep -i "sclk\|mclk\|power" | tr '\n' ' '
echo
