#!/bin/bash
# round 6, session P: conc / spec_rates / dydt through k_jvd's dydt build (default) against k_rate's lean kernels (PJ_RBLK_RATE_FAST=0)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
: > $O/r06_rate_fast_p.txt
for w in gri:1000000 usc:200000; do
  m=${w%%:*}; n=${w##*:}
  if [ $m = gri ]; then MECH=$R/pyjac_amd/data/gri30_shaped.inp; else MECH=$R/pyjac_amd/data/usc2_shaped.inp; fi
  for f in 1 0 1 0; do
    echo "== $m PJ_RBLK_RATE_FAST=$f" >> $O/r06_rate_fast_p.txt
    PJ_RBLK_RATE_FAST=$f timeout 300 python tools/rblk_variants.py time $MECH $n fast 2>&1 | grep -v amdgpu >> $O/r06_rate_fast_p.txt
  done
done
cat $O/r06_rate_fast_p.txt
