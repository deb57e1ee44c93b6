#!/bin/bash
# round 5: AoS Jacobians (SoA staging block + transpose): states per staging block (PJ_RBLK_AOS_CHUNK; default 65536)
mkdir -p gpurun_out
{
for c in 0 4096 8192 16384 32768 131072; do
  echo "PJ_RBLK_AOS_CHUNK=$c"
  PJ_RBLK_AOS_CHUNK=$c python bench.py --layout aos --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('   gri aos  ms %.3f frac %.3f' % (j['ms_per_step'], j['roofline']['frac']))"
  PJ_RBLK_AOS_CHUNK=$c python bench.py --workload usc --layout aos --steps 20 --warmup 3 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('   usc aos  ms %.3f frac %.3f' % (j['ms_per_step'], j['roofline']['frac']))"
done
} 2>&1 | tee gpurun_out/r05_aos_chunk.txt
