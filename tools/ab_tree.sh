#!/bin/bash
# Developer tool: same-box A/B of the working tree against another checkout of the repository (for changes that alter the C ABI,
# where tools/ab.sh's library swap cannot be used): `git archive <rev> | tar -x -C tmp_ab/prev_tree`, build its library in place,
# then on the GPU box:   tools/ab_tree.sh [-r REPS] [tree]     (default tree: tmp_ab/prev_tree; BENCH_ARGS adds bench flags)
cd "$(dirname "$0")/.."
REPS=2
if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
TREE=${1:-tmp_ab/prev_tree}
Q="--steps 4 --warmup 1 --cpu-seconds 0 --policy-envs 0 --config5-envs 0 --update-epochs 0 --congested-steps 2 --details '' $BENCH_ARGS"
for i in $(seq $REPS); do
  printf "%-8s " here; eval python bench.py $Q 2>/dev/null | python tools/bench_brief.py
  printf "%-8s " there; (cd $TREE && eval python bench.py $Q 2>/dev/null | python tools/bench_brief.py)
done
