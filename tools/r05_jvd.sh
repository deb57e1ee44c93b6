#!/bin/bash
# round 5: w = J v as a directional derivative (k_jvd) against the row kernels' fused product (shipped library)
mkdir -p gpurun_out
{
python tools/jv_time.py pyjac_amd/data/gri30_shaped.inp 1000000 rblk jvd2
python tools/jv_time.py pyjac_amd/data/usc2_shaped.inp 200000 rblk jvdg jvdg2 jvdg24
} 2>&1 | grep -v "Warning\|amdgpu.ids" | tee gpurun_out/r05_jvd.txt
