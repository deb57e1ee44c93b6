#!/bin/bash
export TMPDIR=/tmp PJ_VAR_RATES=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
M=pyjac_amd/data/usc2_shaped.inp
timeout 900 python tools/rblk_variants.py time $M 200000 rblk $@ rblk $@ 2>&1 | grep -v amdgpu.ids > $O/r05_usc_variants_g.txt
cat $O/r05_usc_variants_g.txt
