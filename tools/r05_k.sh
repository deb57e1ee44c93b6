#!/bin/bash
export TMPDIR=/tmp PJ_VAR_RATES=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python tools/jv_time.py pyjac_amd/data/gri30_shaped.inp 1000000 rblk jv6 jv7 rblk jv6 jv7 2>&1 | grep -v amdgpu.ids > $O/r05_gri_jv_k.txt
cat $O/r05_gri_jv_k.txt
timeout 900 python tools/rblk_variants.py time pyjac_amd/data/usc2_shaped.inp 200000 rblk wide3p $@ rblk wide3p $@ 2>&1 | grep -v amdgpu.ids > $O/r05_usc_variants_k.txt
cat $O/r05_usc_variants_k.txt
