#!/bin/bash
export TMPDIR=/tmp PJ_VAR_RATES=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python tools/rblk_variants.py time pyjac_amd/data/usc2_shaped.inp 200000 rblk ecl ecldla wide2b wide1 > $O/r05_usc_variants_d.txt 2>&1
timeout 900 python tools/rblk_variants.py time pyjac_amd/data/gri30_shaped.inp 1000000 rblk ecl3 ecld3 rblk ecl3 ecld3 > $O/r05_gri_variants_d.txt 2>&1
cat $O/r05_usc_variants_d.txt $O/r05_gri_variants_d.txt
