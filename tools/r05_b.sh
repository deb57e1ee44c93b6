#!/bin/bash
# round 5: PJQ_DEFER (rows of block b stored during the visits of block b + 1) on top of PJQ_ECL
export TMPDIR=/tmp PJ_VAR_RATES=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python tools/rblk_variants.py time pyjac_amd/data/usc2_shaped.inp 200000 rblk ecl ecld ecldla ecld8 > $O/r05_usc_variants_b.txt 2>&1
timeout 900 python tools/rblk_variants.py time pyjac_amd/data/gri30_shaped.inp 1000000 rblk ecl ecld > $O/r05_gri_variants_b.txt 2>&1
cat $O/r05_usc_variants_b.txt $O/r05_gri_variants_b.txt
