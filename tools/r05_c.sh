#!/bin/bash
export TMPDIR=/tmp PJ_VAR_RATES=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python tools/rblk_variants.py time pyjac_amd/data/usc2_shaped.inp 200000 rblk ecldla wide2 $1 > $O/r05_usc_variants_c.txt 2>&1
cat $O/r05_usc_variants_c.txt
