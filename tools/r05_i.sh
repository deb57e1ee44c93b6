#!/bin/bash
export TMPDIR=/tmp PJ_VAR_RATES=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 300 python tools/rblk_timing.py pyjac_amd/data/usc2_shaped.inp 200000 pyjac_amd/spec/var/usc2_shaped_$1.so 2>&1 | grep -v "amdgpu.ids" > $O/r05_phase_$1.txt
grep -v group $O/r05_phase_$1.txt; grep group $O/r05_phase_$1.txt | tail -4
