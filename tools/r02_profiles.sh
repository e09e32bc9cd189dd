#!/bin/bash
# Developer tool: the measurement runs behind profiles/r02_* and DESIGN.md (run on the GPU box through gpurun).
cd "$(dirname "$0")/.."
O=gpurun_out/r02final
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
Q="--cpu-seconds 0 --congested-window 0 --policy-envs 0"
for cfg in "c3_b1 --edges 1024 --agents 1024 --envs 1" "c3_b256 --edges 1024 --agents 1024 --envs 256" \
           "c3_b2048 --edges 1024 --agents 1024 --envs 2048" "c4_b1 --envs 1" "c4_b256 --envs 256" "c4_b512 --envs 512" \
           "c4_b1024 --envs 1024" "c4_b2048 --envs 2048" "c4_b4096 --envs 4096" "c4_b8192 --envs 8192" "c4_b32768 --envs 32768" \
           "c5_b256 --edges 100000 --agents 262144 --envs 256" "c5_b1024 --edges 100000 --agents 262144 --envs 1024"; do
  set -- $cfg; name=$1; shift
  python bench.py $Q --steps 3 --no-kernel-timing "$@" > $O/bench_$name.json.log 2>/dev/null
  python - <<PY
import json
d=json.loads(open("$O/bench_$name.json.log").read().strip().splitlines()[-1])
print("$name", "%.4gM env-steps/s" % (d["value"]/1e6), "%.2f ms/iter" % d["ms_per_step"], d["config"]["rollout_kernels"])
PY
done
# kernel-trace statistics
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o run -- python3 bench.py $Q --steps 3 --no-kernel-timing > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4_b4096 -o run -- python3 bench.py $Q --steps 3 --no-kernel-timing --envs 4096 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_b256 -o run -- python3 bench.py $Q --steps 3 --no-kernel-timing --edges 1024 --agents 1024 --envs 256 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c3_b1 -o run -- python3 bench.py $Q --steps 3 --no-kernel-timing --edges 1024 --agents 1024 --envs 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4_b256 -o run -- python3 bench.py $Q --steps 3 --no-kernel-timing --envs 256 > /dev/null 2>&1
# the state-dependent policy line on its own (per-frame observation -> edge MLP -> GraphDistribution draw -> frame)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_policy -o run -- python3 bench.py --cpu-seconds 0 --congested-window 0 --steps 1 --policy-steps 2 --no-kernel-timing > /dev/null 2>&1
# HBM traffic of the rollout kernels on the bench's own rollout
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -o run -- python3 bench.py $Q --steps 1 --warmup 1 --no-kernel-timing > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -o run -- python3 bench.py $Q --steps 1 --warmup 1 --no-kernel-timing > /dev/null 2>&1
python3 tools/pmc_bench.py $O/pmc_f $O/pmc_w --out $O/pmc_traffic.json > /dev/null
# the default bench line last, against the traffic record just measured (in this run's copy of the tree)
cp $O/pmc_traffic.json profiles/r02_pmc_traffic.json
python bench.py > $O/bench_default.json.log 2> $O/bench_default.err
rm -f $O/*/run_kernel_trace.csv
ls $O
