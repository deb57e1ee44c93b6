#!/bin/bash
# round 6, session G: lean exp / log in the falloff / PLOG body (k_pre, k_rate), lane groups rebalanced with the measured costs
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
GRI=$R/pyjac_amd/data/gri30_shaped.inp; USC=$R/pyjac_amd/data/usc2_shaped.inp
timeout 900 python tools/rblk_variants.py time $GRI 1000000 fat lean leanbal fat lean leanbal 2>&1 | grep -v amdgpu > $O/r06_gri_variants_g.txt
timeout 900 python tools/rblk_variants.py time $USC 200000 fat lean fat lean 2>&1 | grep -v amdgpu >> $O/r06_gri_variants_g.txt
cat $O/r06_gri_variants_g.txt
