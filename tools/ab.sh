#!/bin/bash
# Developer tool: same-box A/B of library variants (tmp_ab/libtarl_hip_<tag>.so built by `make -C tarl-simulator_amd/csrc variant`).
# usage: tools/ab.sh [-r REPS] tag1 tag2 ...     ("base" = the in-tree library);  BENCH_ARGS adds bench flags
cd "$(dirname "$0")/.."
REPS=2
if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
Q="--steps 4 --warmup 1 --cpu-seconds 0 --policy-envs 0 --config5-envs 0 --update-epochs 0 --congested-steps 2 --details '' $BENCH_ARGS"
for i in $(seq $REPS); do
  for tag in "$@"; do
    if [ $tag = base ]; then L=""; else L="$PWD/tmp_ab/libtarl_hip_$tag.so"; fi
    printf "%-12s " $tag
    TARL_HIP_LIB=$L python bench.py $Q 2>/dev/null | python tools/bench_brief.py
  done
done
