#!/bin/bash
export TMPDIR=/tmp PJ_VAR_RATES=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
M=pyjac_amd/data/usc2_shaped.inp
for t in eclt wide6t; do
timeout 300 python tools/rblk_timing.py $M 200000 pyjac_amd/spec/var/usc2_shaped_$t.so 2>&1 | grep -v "amdgpu.ids" > $O/r05_phase_$t.txt
tail -7 $O/r05_phase_$t.txt
done
