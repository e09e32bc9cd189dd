#!/bin/bash
# Developer tool: SQ wait / activity counters of the per-edge MLP forward kernels on tools/time_edge_mlp.py's launches
# (one counter set per pass, --pmc with --kernel-trace only).   usage (GPU box): tools/pmc_mlp.sh <tag>   -> gpurun_out/<tag>/
cd "$(dirname "$0")/.."
TAG=${1:-pmc_mlp}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
SETS=("SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES"
      "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES"
      "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_VMEM"
      "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_SALU")
i=0
for s in "${SETS[@]}"; do
  i=$((i + 1))
  rocprofv3 --pmc $s --kernel-trace --output-format csv -d $O/p$i -o run -- python3 tools/time_edge_mlp.py --reps 3 > $O/p$i.log 2>&1 || { echo "pass $i FAILED"; tail -5 $O/p$i.log; exit 1; }
done
python3 tools/pmc_reduce.py $O/p* --last 3 --kernels k_edge_mlp_fwd_bf16,k_edge_mlp_fwd_x3,k_edge_mlp_fwd_f32 | tee $O/counters.txt
