#!/bin/bash
# round 5: w = J v as a directional derivative (k_jvd): variant libraries (tools/rblk_variants.py build <mech> "<tag>:" with
# PJ_RBLK_PAIR_MODES= and PJ_RBLK_JVD_GEOMETRY / PJ_RBLK_JVD_KC_GLOBAL / PJ_RBLK_JVD_DEFINES -> pyjac_amd/spec/var/) against
# each other and the shipped library ("rblk"), same box: r05_jvd.sh "<GRI tags>" "<USC tags>"  (profiles/r05_jvd_variants.txt)
mkdir -p gpurun_out
{
python tools/jv_time.py pyjac_amd/data/gri30_shaped.inp 1000000 rblk $1
python tools/jv_time.py pyjac_amd/data/usc2_shaped.inp 200000 rblk $2
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r05_jvd.txt
