#!/bin/bash
# round 6, session N: the last row kernel's energy-row column sums requested in front of its last block's stores (PJQ_EARLY_E)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
USC=$R/pyjac_amd/data/usc2_shaped.inp
PJ_VAR_RATES=0 timeout 600 python tools/rblk_variants.py time $USC 200000 rblk early rblk early rblk early 2>&1 | grep -v amdgpu > $O/r06_usc_variants_n.txt
timeout 300 python tools/rblk_timing.py $USC 200000 pyjac_amd/spec/var/usc2_shaped_earlytiming.so 2>&1 | grep -v amdgpu | grep "kernel\|all\|ms per" >> $O/r06_usc_variants_n.txt
cat $O/r06_usc_variants_n.txt
