#!/bin/bash
# round 5: run-time launch settings of the pj_rblk libraries (streams / chunk / tail split), shipped libraries, same box
mkdir -p gpurun_out
run() { echo "== $1 | $2"; env $2 PJ_VAR_RATES=0 python tools/rblk_variants.py time $1 rblk 2>&1 | grep "^rblk" | cut -c1-80; }
{
for m in "pyjac_amd/data/usc2_shaped.inp 200000" "pyjac_amd/data/gri30_shaped.inp 1000000"; do
  run "$m" "PJ_X=0"
  run "$m" "PJ_RBLK_SPLIT=0"
  for c in 16384 32768 49152 65536 131072; do
    run "$m" "PJ_RBLK_STREAMS=2 PJ_RBLK_CHUNK=$c"
  done
  run "$m" "PJ_RBLK_STREAMS=3 PJ_RBLK_CHUNK=32768"
  run "$m" "PJ_RBLK_STREAMS=4 PJ_RBLK_CHUNK=32768"
  run "$m" "PJ_X=1"
done
} 2>&1 | tee gpurun_out/r05_split_sweep.txt
