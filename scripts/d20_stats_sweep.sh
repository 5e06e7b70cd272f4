#!/bin/bash
# k_stats_gemm<20> (28 % of the headline step): the tile shape of round 3 (C5 x CGW3 = the 15 column tiles of 231 monomials,
# 4 sample slices, 12 wavefronts) against its neighbours, re-measured on round 5's code.  GPU box; rebuilds the unit in place
# (the last line restores the product's shape).
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for K in 32 64; do
export KB_ARGS="--K $K"
bash scripts/tune_unit.sh stats 20 estep:k_stats \
  "-DPMC_GEMM_C=5 -DPMC_GEMM_CGW=3 -DPMC_GEMM_SL=4 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=5 -DPMC_GEMM_CGW=3 -DPMC_GEMM_SL=2 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=5 -DPMC_GEMM_CGW=3 -DPMC_GEMM_SL=2 -DPMC_GEMM_NS=2 -DPMC_GEMM_WGS=1" \
  "-DPMC_GEMM_C=5 -DPMC_GEMM_CGW=3 -DPMC_GEMM_SL=4 -DPMC_GEMM_NS=1" \
  "-DPMC_GEMM_C=3 -DPMC_GEMM_CGW=5 -DPMC_GEMM_SL=2 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=4 -DPMC_GEMM_CGW=4 -DPMC_GEMM_SL=4 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=8 -DPMC_GEMM_CGW=2 -DPMC_GEMM_SL=4 -DPMC_GEMM_NS=2" \
  "-DPMC_GEMM_C=5 -DPMC_GEMM_CGW=3 -DPMC_GEMM_SL=4 -DPMC_GEMM_NS=2"
done
