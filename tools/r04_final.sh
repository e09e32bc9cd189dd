#!/bin/bash
# Developer tool (round 4): the measurement runs behind profiles/r04_* and DESIGN.md (run on the GPU box through gpurun).
# usage: tools/r04_final.sh [tag]    -> gpurun_out/<tag>/ ; tools/r04_collect.sh <tag> copies the summaries into profiles/
cd "$(dirname "$0")/.."
TAG=${1:-r04final}
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
Q="--cpu-seconds 0 --congested-window 0 --policy-envs 0"
# 1. kernel-trace statistics: default line, congested regime, state-dependent policy
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o run -- python3 bench.py $Q --steps 3 --no-kernel-timing > $O/prof_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_congested -o run -- python3 bench.py $Q --steps 2 --no-kernel-timing --departure-window 600 > $O/prof_congested.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_policy -o run -- python3 bench.py --cpu-seconds 0 --congested-window 0 --steps 1 --policy-steps 2 --no-kernel-timing > $O/prof_policy.log 2>&1
rm -f $O/*/run_kernel_trace.csv
echo "[r04_final] kernel stats done"
# 2. HBM traffic (PMC passes, default + congested) with the request counters beside FETCH_SIZE / WRITE_SIZE
bash tools/r04_pmc.sh $TAG/pmc "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAVE_CYCLES" > $O/pmc.log 2>&1
cp $O/pmc/pmc_traffic.json profiles/r04_pmc_traffic.json
cp $O/pmc/pmc_traffic_congested.json profiles/r04_pmc_traffic_congested.json
echo "[r04_final] pmc done"
# 3. other sizes (env-steps/s of the whole PPO iteration)
for cfg in "c3_b1 --edges 1024 --agents 1024 --envs 1" "c3_b256 --edges 1024 --agents 1024 --envs 256" \
           "c3_b2048 --edges 1024 --agents 1024 --envs 2048" "c4_b1 --envs 1" "c4_b256 --envs 256" "c4_b1024 --envs 1024" \
           "c4_b4096 --envs 4096" "c4_b8192 --envs 8192" "c4_b32768 --envs 32768" \
           "c5_b256 --edges 100000 --agents 262144 --envs 256" "c5_b1024 --edges 100000 --agents 262144 --envs 1024"; do
  set -- $cfg; name=$1; shift
  python bench.py $Q --steps 3 --no-kernel-timing "$@" > $O/bench_$name.json.log 2>/dev/null
  python tools/bench_brief.py < $O/bench_$name.json.log | sed "s/^/$name /"
done
# 4. the state-dependent policy at two batch sizes
python bench.py --cpu-seconds 0 --congested-window 0 --steps 1 --policy-envs 8192 > $O/bench_policy_b8192.json.log 2>/dev/null
python tools/bench_brief.py < $O/bench_policy_b8192.json.log | sed "s/^/policy_b8192 /"
# 5. the default bench line last, against the traffic records just measured (in this run's copy of the tree)
python bench.py > $O/bench_default.json.log 2> $O/bench_default.err
python tools/bench_brief.py < $O/bench_default.json.log | sed "s/^/default /"
ls $O
