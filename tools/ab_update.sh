#!/bin/bash
# Developer tool: same-box A/B of the update path (bench.py's update_path object: 8 epochs x 4 096-frame minibatches) of the working
# tree against another checkout (default tmp_ab/prev_tree): env-steps/s, ms per minibatch step, the critic pass over all frames.
# usage (GPU box): tools/ab_update.sh [-r REPS] [tree]
cd "$(dirname "$0")/.."
REPS=2
if [ "$1" = "-r" ]; then REPS=$2; shift 2; fi
TREE=${1:-tmp_ab/prev_tree}
Q="--steps 2 --warmup 1 --cpu-seconds 0 --policy-envs 0 --config5-envs 0 --congested-window 0 --update-steps 2"
show() { python3 -c "
import json,sys
d=json.load(open(sys.argv[1])); u=d['update_path']
print(f\"{d['value']/1e6:6.2f} M headline {d['ms_per_step']:7.2f} ms/iter | update_path {u['value']/1e6:6.2f} M  {u['ms_per_minibatch_step']:.2f} ms/minibatch  critic pass {u['stage_us']['critic_all_frames']:.0f} us\")" $1; }
for i in $(seq $REPS); do
  printf "%-8s " here; python bench.py $Q --details /tmp/ab_here.json > /dev/null 2>&1 && show /tmp/ab_here.json
  printf "%-8s " there; (cd $TREE && python bench.py $Q --details /tmp/ab_there.json > /dev/null 2>&1) && show /tmp/ab_there.json
done
