#!/bin/bash
# Developer tool (round 3): PMC calibration passes of tools/pmc_cal.hip on the GPU box -> gpurun_out/cal_*
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 tools/pmc_cal.hip -o gpurun_out/pmc_cal
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1 || true
gpurun_out/pmc_cal > gpurun_out/pmc_cal.out
for set in "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum" "TCC_BUBBLE_sum TCC_EA0_RD_UNCACHED_32B_sum"; do
  tag=$(echo $set | tr ' ' '+')
  rm -rf gpurun_out/cal_$tag
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/cal_$tag -o run -- gpurun_out/pmc_cal > gpurun_out/cal_$tag.log 2>&1 || echo "pass $tag failed (see gpurun_out/cal_$tag.log)"
done
python3 tools/pmc_cal_reduce.py gpurun_out/pmc_cal.out gpurun_out/cal_*/ > gpurun_out/r03_pmc_calibration.txt
tail -n 80 gpurun_out/r03_pmc_calibration.txt
