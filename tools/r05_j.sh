#!/bin/bash
export TMPDIR=/tmp PJ_VAR_RATES=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
M=pyjac_amd/data/usc2_shaped.inp
timeout 900 python tools/rblk_variants.py time $M 200000 rblk eclab rblk eclab rblk eclab 2>&1 | grep -v amdgpu.ids > $O/r05_usc_variants_j.txt
cat $O/r05_usc_variants_j.txt
timeout 300 python tools/rblk_timing.py $M 200000 pyjac_amd/spec/var/usc2_shaped_eclabt.so 2>&1 | grep -v "amdgpu.ids" > $O/r05_phase_eclabt.txt
grep -v group $O/r05_phase_eclabt.txt | tail -3; grep group $O/r05_phase_eclabt.txt | tail -4
