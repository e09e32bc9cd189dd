#!/bin/bash
# Developer tool: bench the fused path under different chunk knobs (each run is its own process: knobs are read once).
cd "$(dirname "$0")/.."
for kv in "TARL_NCHUNK_CHOICE=8" "TARL_NCHUNK_CHOICE=4" "TARL_NCHUNK_CHOICE=16" "TARL_NCHUNK_CHOICE=32" "TARL_NCHUNK_CHOICE=64" "TARL_NCHUNK=1" "TARL_NCHUNK=4" "TARL_NCHUNK_DIR=2"; do
  v=$(env $kv python bench.py --steps 2 --warmup 1 --cpu-seconds 0 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)
  echo "$kv $v"
done
