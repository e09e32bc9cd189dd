#!/bin/bash
# Developer tool: copy the summaries tools/r04_final.sh left under gpurun_out/<tag> into profiles/ as r04_*.
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r04final}; P=profiles
cp $O/bench_default.json.log $P/r04_bench_b16384.json.log
cp $O/pmc/pmc_traffic.json $P/r04_pmc_traffic.json
cp $O/pmc/pmc_traffic_congested.json $P/r04_pmc_traffic_congested.json
S="python tools/profile_summary.py stats"
for n in default congested policy; do cp $O/prof_$n/run_kernel_stats.csv $P/r04_${n}_kernel_stats.csv; done
$S $O/prof_default/run_kernel_stats.csv "rocprofv3 --kernel-trace --stats of bench.py --steps 3 --no-kernel-timing --cpu-seconds 0 --congested-window 0 --policy-envs 0 (default: config 4, B = 16384), round 4 final" > $P/r04_default_kernel_stats.txt
$S $O/prof_congested/run_kernel_stats.csv "the same with --departure-window 600 --steps 2 (the congested regime: every agent departs within 600 s)" > $P/r04_congested_kernel_stats.txt
$S $O/prof_policy/run_kernel_stats.csv "rocprofv3 --kernel-trace --stats of bench.py --cpu-seconds 0 --congested-window 0 --steps 1 --policy-steps 2 --no-kernel-timing (state-dependent policy lines at B = 2048), round 4 final" > $P/r04_policy_kernel_stats.txt
{
  echo "# bench.py --steps 3 --no-kernel-timing --cpu-seconds 0 --congested-window 0 --policy-envs 0 at other sizes (1x MI355X, round 4 final, default B = 16384)"
  for n in c3_b1 c3_b256 c3_b2048 c4_b1 c4_b256 c4_b1024 c4_b4096 c4_b8192 c4_b32768 c5_b256 c5_b1024; do
    python tools/bench_brief.py < $O/bench_$n.json.log | sed "s/^/$n /"
  done
  python tools/bench_brief.py < $O/bench_policy_b8192.json.log | sed "s/^/policy_envs_8192 /"
} > $P/r04_size_sweep.txt
cp gpurun_out/r04_pmc_calibration.txt $P/r04_pmc_calibration.txt 2>/dev/null
ls $P | grep r04
