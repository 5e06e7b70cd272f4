#!/bin/bash
# Where k_mgemm's time goes (GPU box): the product against timing-only variants of pmc_mgemm_d40_p0 (wrong numbers) that
# leave a piece out -- built here first, e.g.
#   PMC_VARIANT=mgA PMC_VARIANT_UNITS=pmc_mgemm_d40_p0 PMC_EXTRA_FLAGS="-DPMC_MG_AB_NOEPI -DPMC_MG_AB_NOPRO -DPMC_MG_AB_NOBAR" python -m pypmc_amd.build
# Switches: PMC_MG_AB_NOEPI (no per-pass epilogue), _NOPRO (no sample loads in the prologue), _NOBAR (no chunk barriers /
# staging), _NOMUL (no v_mul_f64 for the monomials), _NODREAD / _NOTHREAD (no LDS reads of the sample / coefficient image),
# PMC_MG_MUL_INTERLEAVED / PMC_MG_MUL_AFTER (where the products of the next step sit).  Results: profiles/r04_mgemm_ab.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for round in 1 2; do
for v in "" "$@"; do
  if [ -n "$v" ]; then export PMC_HIP_LIBRARY=$R/pypmc_amd/lib/libpmc_hip_$v.so; else unset PMC_HIP_LIBRARY; fi
  python scripts/mgemm_time.py 2>&1 | grep -v amdgpu.ids
done
done
