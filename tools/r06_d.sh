#!/bin/bash
# round 6, session D: plain instead of nontemporal Jacobian stores; the kernels with two-instruction exponentials (results wrong:
# how sensitive is the step to its fp64 instruction count?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
GRI=$R/pyjac_amd/data/gri30_shaped.inp; USC=$R/pyjac_amd/data/usc2_shaped.inp
PJ_VAR_RATES=0 timeout 900 python tools/rblk_variants.py time $GRI 1000000 base3 stg nont fakeexp expt base3 stg nont fakeexp expt 2>&1 | grep -v amdgpu > $O/r06_gri_variants_d.txt
PJ_VAR_RATES=0 timeout 900 python tools/rblk_variants.py time $USC 200000 stg nont stg nont 2>&1 | grep -v amdgpu >> $O/r06_gri_variants_d.txt
cat $O/r06_gri_variants_d.txt
