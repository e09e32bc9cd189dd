#!/bin/bash
# Developer tool: SQ wait / activity counters of the critic's pass over the rollout (k_critic_fwd_slab_u8x3) inside bench.py's
# update-path object (one counter set per pass, --pmc with --kernel-trace only).   usage (GPU box): tools/pmc_critic.sh <tag>
cd "$(dirname "$0")/.."
TAG=${1:-pmc_critic}; O=gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
SETS=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
      "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
      "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU"
      "SQ_INSTS_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_COEXEC_CYCLES")
i=0
for s in "${SETS[@]}"; do
  i=$((i + 1))
  rocprofv3 --pmc $s --kernel-trace --output-format csv -d $O/p$i -o run -- python3 bench.py --steps 1 --warmup 0 --cpu-seconds 0 --policy-envs 0 --config5-envs 0 --congested-window 0 --update-steps 1 --update-epochs 2 --no-kernel-timing --details '' > $O/p$i.log 2>&1 || { echo "pass $i FAILED"; tail -5 $O/p$i.log; exit 1; }
  rm -f $O/p$i/run_kernel_trace.csv
done
python3 tools/pmc_reduce.py $O/p* --last 2 --kernels k_critic_fwd_slab_u8x3 | tee $O/counters.txt
