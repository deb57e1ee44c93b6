#!/bin/bash
export TMPDIR=/tmp PJ_VAR_RATES=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
M=pyjac_amd/data/gri30_shaped.inp
timeout 900 python tools/rblk_variants.py time $M 1000000 rblk jv4 jv5 rblk jv4 jv5 2>&1 | grep -v amdgpu.ids > $O/r05_gri_variants_h.txt
cat $O/r05_gri_variants_h.txt
timeout 600 python tools/jv_time.py $M 1000000 rblk jv4 jv5 rblk jv4 jv5 2>&1 | grep -v amdgpu.ids > $O/r05_gri_jv_h.txt
cat $O/r05_gri_jv_h.txt
