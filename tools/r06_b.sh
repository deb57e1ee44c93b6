#!/bin/bash
# round 6, session B: GRI-shaped variants (same box): shipped-geometry base, visit constants through the scalar cache (PJQ_VCT),
# phase maps with and without Jacobian stores
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
GRI=$R/pyjac_amd/data/gri30_shaped.inp
PJ_VAR_RATES=0 timeout 900 python tools/rblk_variants.py time $GRI 1000000 base vct nostore base vct > $O/r06_gri_variants_b.txt 2>&1
grep -v amdgpu $O/r06_gri_variants_b.txt
for t in timing timing_nostore; do
  timeout 600 python tools/rblk_timing.py $GRI 1000000 $R/pyjac_amd/spec/var/gri30_shaped_$t.so > $O/r06_rblk_gri_phase_cycles_$t.txt 2>&1
  grep -v amdgpu $O/r06_rblk_gri_phase_cycles_$t.txt
done
