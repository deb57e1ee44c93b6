#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-tab}; mkdir -p $O
for d in 0 8 16 24 32 4; do echo "PJ_TAB_DBG=$d"; PJ_TAB_DBG=$d python tools/rblk_variants.py time pyjac_amd/data/gri30_shaped.inp 1000000 tab 2>&1 | grep -v amdgpu.ids; done | tee $O/gri_dbg.txt
