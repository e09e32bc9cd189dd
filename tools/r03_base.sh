#!/bin/bash
# Developer tool (round 3): baseline evidence at the start of the round — kernel-trace statistics of the default line and of
# the congested regime (bench.py --departure-window 600), into gpurun_out/r03base/.
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r03base}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
Q="--cpu-seconds 0 --congested-window 0 --policy-envs 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o run -- python3 bench.py $Q --steps 3 --no-kernel-timing > $O/prof_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_congested -o run -- python3 bench.py $Q --steps 2 --no-kernel-timing --departure-window 600 > $O/prof_congested.log 2>&1
rm -f $O/*/run_kernel_trace.csv
python tools/profile_summary.py stats $O/prof_default/run_kernel_stats.csv "default" | head -24
python tools/profile_summary.py stats $O/prof_congested/run_kernel_stats.csv "congested (--departure-window 600)" | head -24
