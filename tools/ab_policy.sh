#!/bin/bash
# Developer tool: same-box A/B of the state-dependent-policy lines under different environment knobs.
# usage: tools/ab_policy.sh [-r REPS] "VAR=a" "VAR=b" ...
cd "$(dirname "$0")/.."
REPS=2
if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
Q="--steps 1 --warmup 1 --cpu-seconds 0 --congested-window 0 --config5-envs 0 --update-epochs 0 --policy-steps 2 --details '' $BENCH_ARGS"
for i in $(seq $REPS); do
  for kv in "$@"; do
    printf "%-28s " "$kv"
    eval env $kv python bench.py $Q 2>/dev/null | python tools/bench_brief.py | sed 's/.*| policy/policy/'
  done
done
