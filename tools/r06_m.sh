#!/bin/bash
# round 6, session M: geometry / accumulator-budget variants of the row-block kernels for SMALL mechanisms (17 - 27 species)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/r06_small_variants_m.txt
for m in sweep_n017 sweep_r3 sweep_r1; do
  echo "== $m" >> $O/r06_small_variants_m.txt
  tags=$(cd pyjac_amd/spec/var; ls ${m}_*.so | sed "s/${m}_//; s/\.so//" | tr '\n' ' ')
  PJ_VAR_RATES=0 timeout 600 python tools/rblk_variants.py time tests/golden/sweep/$m.inp 1000000 $tags base 2>&1 | grep -v amdgpu >> $O/r06_small_variants_m.txt
done
cat $O/r06_small_variants_m.txt
