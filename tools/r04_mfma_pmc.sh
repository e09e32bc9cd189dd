#!/bin/bash
# Developer tool (round 4): matrix-pipe counters of the MFMA kernels (per-edge MLP x3 / fp32 / bf16, critic pass) — two
# rocprofv3 --pmc passes (--kernel-trace only) of bench.py's policy lines. usage: tools/r04_mfma_pmc.sh [tag] -> gpurun_out/<tag>/
cd "$(dirname "$0")/.."
TAG=${1:-r04mfma}; O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
Q="--steps 1 --warmup 1 --cpu-seconds 0 --congested-window 0 --policy-steps 1 --no-kernel-timing"
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES"; do
  d=$O/pass_$i; i=$((i+1)); rm -rf $d
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o run -- python3 bench.py $Q > $d.log 2>&1 || echo "pass '$set' failed"
  rm -f $d/run_kernel_trace.csv
done
python3 tools/pmc_reduce.py $O/pass_0 $O/pass_1 --last 100 --kernels k_edge_mlp_fwd_x3,k_edge_mlp_fwd_f32,k_edge_mlp_fwd_bf16,k_critic_fwd_slab_u8x3,k_edge_mlp_bwd_edges > $O/mfma_pmc.txt 2>&1
cat $O/mfma_pmc.txt
