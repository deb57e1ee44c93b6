#!/bin/bash
# timing of the no-compile path k_tab with parts switched off (PJ_TAB_DBG: 1 no reaction arithmetic, 2 no
# accumulation, 4 no output phase), and of the other mechanisms / k_eval
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-tab}; mkdir -p $O
for d in 0 1 2 4 3 7; do echo "PJ_TAB_DBG=$d"; PJ_TAB_DBG=$d python tools/rblk_variants.py time pyjac_amd/data/gri30_shaped.inp 1000000 tab 2>&1 | grep -v amdgpu.ids; done | tee $O/gri_dbg.txt
for L in 64 128; do echo "PJ_TAB_L=$L"; PJ_TAB_L=$L python tools/rblk_variants.py time pyjac_amd/data/gri30_shaped.inp 1000000 tab 2>&1 | grep -v amdgpu.ids; done | tee $O/gri_L.txt
python tools/rblk_variants.py time pyjac_amd/data/usc2_shaped.inp 200000 tab keval 2>&1 | grep -v amdgpu.ids | tee $O/usc.txt
python tools/rblk_variants.py time pyjac_amd/data/h2o2_n2.inp 1000000 tab keval 2>&1 | grep -v amdgpu.ids | tee $O/h2.txt
