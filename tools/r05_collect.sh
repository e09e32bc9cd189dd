#!/bin/bash
# Developer tool: copy the summaries tools/r05_final.sh left under gpurun_out/<tag> into profiles/ as r05_*; refuses
# missing or empty inputs.
set -u
cd "$(dirname "$0")/.."
O=gpurun_out/${1:-r05final}; P=profiles
need() { [ -s "$1" ] || { echo "[r05_collect] missing or empty: $1"; exit 1; }; }
need $O/bench_default.json.log; need $O/bench_default_details.json
cp $O/bench_default.json.log $P/r05_bench_b32768.json.log
cp $O/bench_default_details.json $P/r05_bench_b32768_details.json
for f in pmc_traffic pmc_traffic_congested pmc_traffic_c5; do need $O/pmc/$f.json; cp $O/pmc/$f.json $P/r05_$f.json; done
S="python tools/profile_summary.py stats"
for n in default congested c5 policy update; do need $O/prof_$n/run_kernel_stats.csv; cp $O/prof_$n/run_kernel_stats.csv $P/r05_${n}_kernel_stats.csv; done
Q="--cpu-seconds 0 --congested-window 0 --policy-envs 0 --config5-envs 0 --update-epochs 0 --no-kernel-timing"
$S $O/prof_default/run_kernel_stats.csv "rocprofv3 --kernel-trace --stats of bench.py $Q --steps 3 (default: config 4, B = 32768), round 5 final" > $P/r05_default_kernel_stats.txt
$S $O/prof_congested/run_kernel_stats.csv "the same with --departure-window 600 --steps 2 (the congested regime: every agent departs within 600 s)" > $P/r05_congested_kernel_stats.txt
$S $O/prof_c5/run_kernel_stats.csv "the same with --edges 100000 --agents 262144 --envs 2048 --steps 2 (BASELINE config 5)" > $P/r05_c5_kernel_stats.txt
$S $O/prof_policy/run_kernel_stats.csv "rocprofv3 --kernel-trace --stats of bench.py --steps 1 --policy-steps 2 --no-kernel-timing (state-dependent policy lines at B = 4096), round 5 final" > $P/r05_policy_kernel_stats.txt
$S $O/prof_update/run_kernel_stats.csv "rocprofv3 --kernel-trace --stats of bench.py --steps 1 --no-kernel-timing with the update_path object (8 epochs x 4096 frames), round 5 final" > $P/r05_update_kernel_stats.txt
if [ -s $O/bench_c3_b1.json.log ]; then      # (a run without the "sweep" part keeps the committed sweep)
{
  echo "# bench.py $Q --steps 3 at other sizes (1x MI355X, round 5 final, default B = 32768)"
  for n in c3_b1 c3_b256 c3_b2048 c4_b1 c4_b256 c4_b1024 c4_b4096 c4_b8192 c4_b16384 c5_b256 c5_b1024 c5_b4096; do
    need $O/bench_$n.json.log
    python tools/bench_brief.py < $O/bench_$n.json.log | sed "s/^/$n /"
  done
} > $P/r05_size_sweep.txt
fi
ls $P | grep r05
