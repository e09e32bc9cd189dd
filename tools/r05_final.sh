#!/bin/bash
# Developer tool (round 5): the measurement runs behind profiles/r05_* and DESIGN.md (run on the GPU box through gpurun).
# usage: tools/r05_final.sh [tag] [parts]    parts = "stats pmc sweep bench" (default: all) -> gpurun_out/<tag>/
#        tools/r05_collect.sh <tag> copies the summaries into profiles/ (and fails on a missing or empty input)
# Every run's stderr is kept next to its output; a failed run is reported and makes the script exit non-zero.
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r05final}
PARTS=${2:-"stats pmc sweep bench"}
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rc=0
run() {   # run <name> <command...>: stdout -> $O/<name>.json.log, stderr -> $O/<name>.err
  local name=$1; shift
  if ! "$@" > $O/$name.json.log 2> $O/$name.err; then echo "[r05_final] $name FAILED (rc $?): $(tail -2 $O/$name.err)"; rc=1; fi
  [ -s $O/$name.json.log ] || { echo "[r05_final] $name produced no line"; rc=1; }
}
OFF="--cpu-seconds 0 --congested-window 0 --policy-envs 0 --config5-envs 0 --update-epochs 0 --details ''"
if [[ $PARTS == *stats* ]]; then
  # 1. kernel-trace statistics: default line, congested regime, config 5, state-dependent policy, update path
  for cfg in "default --steps 3" "congested --steps 2 --departure-window 600" \
             "c5 --steps 2 --edges 100000 --agents 262144 --envs 2048"; do
    set -- $cfg; name=$1; shift
    if ! eval rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -o run -- python3 bench.py $OFF --no-kernel-timing "$@" > $O/prof_$name.log 2>&1; then
      echo "[r05_final] kernel trace $name FAILED"; tail -3 $O/prof_$name.log; rc=1; fi
  done
  if ! rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_policy -o run -- python3 bench.py --cpu-seconds 0 --congested-window 0 --config5-envs 0 --update-epochs 0 --details '' --steps 1 --policy-steps 2 --no-kernel-timing > $O/prof_policy.log 2>&1; then
    echo "[r05_final] kernel trace policy FAILED"; rc=1; fi
  if ! rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_update -o run -- python3 bench.py --cpu-seconds 0 --congested-window 0 --config5-envs 0 --policy-envs 0 --details '' --steps 1 --no-kernel-timing > $O/prof_update.log 2>&1; then
    echo "[r05_final] kernel trace update FAILED"; rc=1; fi
  rm -f $O/*/run_kernel_trace.csv
  echo "[r05_final] kernel stats done"
fi
if [[ $PARTS == *pmc* ]]; then
  # 2. HBM traffic (PMC passes: default, congested, config 5) with the request counters and the instruction counters
  bash tools/r05_pmc.sh $TAG/pmc "default congested c5" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES" > $O/pmc.log 2>&1 || { echo "[r05_final] pmc FAILED"; tail -5 $O/pmc.log; rc=1; }
  echo "[r05_final] pmc done"
fi
if [[ $PARTS == *sweep* ]]; then
  # 3. other sizes (env-steps/s of the whole PPO iteration)
  for cfg in "c3_b1 --edges 1024 --agents 1024 --envs 1" "c3_b256 --edges 1024 --agents 1024 --envs 256" \
             "c3_b2048 --edges 1024 --agents 1024 --envs 2048" "c4_b1 --envs 1" "c4_b256 --envs 256" "c4_b1024 --envs 1024" \
             "c4_b4096 --envs 4096" "c4_b8192 --envs 8192" "c4_b16384 --envs 16384" \
             "c5_b256 --edges 100000 --agents 262144 --envs 256" "c5_b1024 --edges 100000 --agents 262144 --envs 1024" \
             "c5_b4096 --edges 100000 --agents 262144 --envs 4096"; do
    set -- $cfg; name=$1; shift
    eval run bench_$name python bench.py $OFF --steps 3 --no-kernel-timing "$@"
    python tools/bench_brief.py < $O/bench_$name.json.log | sed "s/^/$name /"
  done
fi
if [[ $PARTS == *bench* ]]; then
  # 4. the default bench line last, against the traffic records just measured (copied into this run's profiles/)
  for f in pmc_traffic pmc_traffic_congested pmc_traffic_c5; do [ -s $O/pmc/$f.json ] && cp $O/pmc/$f.json profiles/r05_$f.json; done
  run bench_default python bench.py --details $O/bench_default_details.json
  python tools/bench_brief.py < $O/bench_default.json.log | sed "s/^/default /"
fi
ls $O
exit $rc
