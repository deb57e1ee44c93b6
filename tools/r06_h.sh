#!/bin/bash
# round 6, session H: scheduler strategy (max-ilp), lane groups of a workgroup shifted against each other (PJQ_GSTAGGER),
# phase cycles of the timing build at 64 / 256 / all workgroups (is the output phase's stall local to the CU or chip-wide?)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
GRI=$R/pyjac_amd/data/gri30_shaped.inp
V=$R/pyjac_amd/spec/var
PJ_VAR_RATES=0 timeout 900 python tools/rblk_variants.py time $GRI 1000000 rblk $(cd $V; ls gri30_shaped_*.so | sed 's/gri30_shaped_//; s/\.so//' | grep -v timing) rblk 2>&1 | grep -v amdgpu > $O/r06_gri_variants_h.txt
cat $O/r06_gri_variants_h.txt
if [ -f $V/gri30_shaped_timing.so ]; then
for n in 4096 16384 65536 1000000; do
  echo "== n = $n" >> $O/r06_gri_phase_by_n.txt
  timeout 300 python tools/rblk_timing.py $GRI $n $V/gri30_shaped_timing.so 2>&1 | grep -v amdgpu >> $O/r06_gri_phase_by_n.txt
done
cat $O/r06_gri_phase_by_n.txt
fi
