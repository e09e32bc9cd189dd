#!/bin/bash
# Developer tool (round 5): HBM-traffic PMC passes of bench.py's own rollout — default line, congested regime, config 5.
# usage: tools/r05_pmc.sh <tag> [modes] [extra counter sets...]   modes = "default congested c5" (quoted list)
#   -> gpurun_out/<tag>/{pmc_traffic.json,pmc_traffic_congested.json,pmc_traffic_c5.json}
# Every pass's output goes to <dir>.log; a failed pass or an empty reduction makes the script fail (non-zero exit).
set -u
cd "$(dirname "$0")/.."
TAG=${1:-r05pmc}; shift
MODES=${1:-"default congested c5"}; shift || true
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
Q="--steps 1 --warmup 1 --cpu-seconds 0 --congested-window 0 --policy-envs 0 --config5-envs 0 --update-epochs 0 --no-kernel-timing"
SETS=("FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "$@")
rc=0
for mode in $MODES; do
  X=""; R=""; F=pmc_traffic.json
  [ $mode = congested ] && { X="--departure-window 600"; R="--departure-window 600"; F=pmc_traffic_congested.json; }
  [ $mode = c5 ] && { X="--edges 100000 --agents 262144 --envs 2048"; R="$X"; F=pmc_traffic_c5.json; }
  dirs=""
  i=0
  for set in "${SETS[@]}"; do
    d=$O/${mode}_$i; i=$((i+1)); rm -rf $d
    if ! rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o run -- python3 bench.py $Q $X > $d.log 2>&1; then
      echo "[r05_pmc] pass '$set' ($mode) FAILED: tail of $d.log"; tail -5 $d.log; rc=1
    fi
    rm -f $d/run_kernel_trace.csv
    dirs="$dirs $d"
    echo "[r05_pmc] $mode pass $i done"
  done
  if ! python3 tools/pmc_bench.py $dirs $R --out $O/$F > /dev/null; then echo "[r05_pmc] reduction of $mode FAILED"; rc=1; fi
  [ -s $O/$F ] || { echo "[r05_pmc] $O/$F missing or empty"; rc=1; }
done
python3 - <<PY
import json, os
for f in ("pmc_traffic.json", "pmc_traffic_congested.json", "pmc_traffic_c5.json"):
    p = "$O/" + f
    if not os.path.exists(p):
        continue
    d = json.load(open(p))
    print(f)
    for k, r in d["kernels"].items():
        print(" ", k, {c: round(v) for c, v in r.items() if isinstance(v, (int, float))})
PY
exit $rc
