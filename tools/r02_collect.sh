#!/bin/bash
# Developer tool: copy the summaries tools/r02_profiles.sh left under gpurun_out/r02final into profiles/ as r02_<version>_*.
# usage: tools/r02_collect.sh v10
cd "$(dirname "$0")/.."
V=r02_$1; O=gpurun_out/r02final; P=profiles
cp $O/bench_default.json.log $P/${V}_bench_b16384.json.log
cp $O/pmc_traffic.json $P/${V}_pmc_traffic.json
cp $O/pmc_traffic.json $P/r02_pmc_traffic.json
for n in default c4_b4096 policy c3_b1 c3_b256 c4_b256; do cp $O/prof_$n/run_kernel_stats.csv $P/${V}_${n}_kernel_stats.csv; done
S="python tools/profile_summary.py stats"
$S $O/prof_default/run_kernel_stats.csv "rocprofv3 --kernel-trace --stats of bench.py --steps 3 --no-kernel-timing --cpu-seconds 0 --congested-window 0 --policy-envs 0 (default: B = 16384), round 2 final" > $P/${V}_default_kernel_stats.txt
$S $O/prof_c4_b4096/run_kernel_stats.csv "the same, --envs 4096 (the size DESIGN's per-launch tables are quoted at)" > $P/${V}_c4_b4096_kernel_stats.txt
$S $O/prof_policy/run_kernel_stats.csv "rocprofv3 --kernel-trace --stats of bench.py --cpu-seconds 0 --congested-window 0 --steps 1 --policy-steps 2 --no-kernel-timing (state-dependent policy lines at B = 2048), round 2 final" > $P/${V}_policy_kernel_stats.txt
$S $O/prof_c3_b1/run_kernel_stats.csv "the same, --edges 1024 --agents 1024 --envs 1 (config 3, k_rollout_env)" > $P/${V}_c3_b1_kernel_stats.txt
$S $O/prof_c3_b256/run_kernel_stats.csv "the same, --edges 1024 --agents 1024 --envs 256 (config 3, k_rollout_env)" > $P/${V}_c3_b256_kernel_stats.txt
$S $O/prof_c4_b256/run_kernel_stats.csv "the same, --envs 256 (config 4, k_rollout_env)" > $P/${V}_c4_b256_kernel_stats.txt
{
  echo "# bench.py --steps 3 --no-kernel-timing --cpu-seconds 0 --congested-window 0 --policy-envs 0 at other sizes (1x MI355X, round 2 final, default B = 16384)"
  echo "# config | env-steps/s | ms per iteration | rollout kernels"
  for n in c3_b1 c3_b256 c3_b2048 c4_b1 c4_b256 c4_b512 c4_b1024 c4_b2048 c4_b4096 c4_b8192 c4_b32768 c5_b256 c5_b1024; do
    python - <<PY
import json
d=json.loads(open("$O/bench_$n.json.log").read().strip().splitlines()[-1])
print("$n", "%.4gM env-steps/s" % (d["value"]/1e6), "%.2f ms/iter" % d["ms_per_step"], d["config"]["rollout_kernels"])
PY
  done
} > $P/${V}_size_sweep.txt
ls $P | grep $V
