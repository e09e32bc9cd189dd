#!/bin/bash
# Developer tool (round 4): HBM-traffic PMC passes of bench.py's own rollout — default line and congested regime.
# usage: tools/r04_pmc.sh <tag> [extra counter sets...]   -> gpurun_out/<tag>/{pmc_traffic.json,pmc_traffic_congested.json}
cd "$(dirname "$0")/.."
TAG=${1:-r04pmc}; shift
O=gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
Q="--steps 1 --warmup 1 --cpu-seconds 0 --congested-window 0 --policy-envs 0 --no-kernel-timing"
SETS=("FETCH_SIZE" "WRITE_SIZE" "$@")
for mode in default congested; do
  X=""; [ $mode = congested ] && X="--departure-window 600"
  dirs=""
  i=0
  for set in "${SETS[@]}"; do
    d=$O/${mode}_$i; i=$((i+1)); rm -rf $d
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o run -- python3 bench.py $Q $X > $d.log 2>&1 || echo "pass '$set' ($mode) failed"
    rm -f $d/run_kernel_trace.csv
    dirs="$dirs $d"
  done
  if [ $mode = default ]; then
    python3 tools/pmc_bench.py $dirs --out $O/pmc_traffic.json > /dev/null
  else
    python3 tools/pmc_bench.py $dirs --departure-window 600 --out $O/pmc_traffic_congested.json > /dev/null
  fi
done
python3 - <<PY
import json
for f in ("pmc_traffic.json", "pmc_traffic_congested.json"):
    d = json.load(open("$O/" + f))
    print(f)
    for k, r in d["kernels"].items():
        print(" ", k, {c: round(v) for c, v in r.items() if isinstance(v, (int, float))})
PY
