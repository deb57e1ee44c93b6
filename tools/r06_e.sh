#!/bin/bash
# round 6, session E: XCD-aware workgroup -> states mapping; 8-byte (general) instead of pair stores
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
GRI=$R/pyjac_amd/data/gri30_shaped.inp; USC=$R/pyjac_amd/data/usc2_shaped.inp
PJ_VAR_RATES=0 timeout 900 python tools/rblk_variants.py time $GRI 1000000 base3 xcd gen base3 xcd gen 2>&1 | grep -v amdgpu > $O/r06_gri_variants_e.txt
PJ_VAR_RATES=0 timeout 900 python tools/rblk_variants.py time $USC 200000 stg xcd stg xcd 2>&1 | grep -v amdgpu >> $O/r06_gri_variants_e.txt
cat $O/r06_gri_variants_e.txt
