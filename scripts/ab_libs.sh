#!/bin/bash
# Runs ON THE GPU BOX: per-op times of the B=64 canonical forward with two builds of the library, alternated (A/B across processes on one device).
#   bash scripts/ab_libs.sh <base.so> <rounds> [ops] [mode ...]      output: gpurun_out/ab_libs.txt
BASE=$1; ROUNDS=${2:-2}; OPS=${3:--}; shift 3 || true
MODES=${@:-split f16}
O=gpurun_out/ab_libs.txt
mkdir -p gpurun_out
: > $O
for r in $(seq 1 $ROUNDS); do
  for m in $MODES; do
    echo "== round $r base" >> $O;  TS2D_AB_LIB=$BASE timeout -k 10 200 python3 scripts/gpu_ops_only.py $m $OPS >> $O 2>&1 || exit 1
    echo "== round $r new" >> $O;   timeout -k 10 200 python3 scripts/gpu_ops_only.py $m $OPS >> $O 2>&1 || exit 1
  done
done
