#!/bin/bash
# does the last partially filled round of workgroups cost a whole round?  (pj_rblk, GRI-/USC-shaped)
# PJ_RBLK_SPLIT=0: one part on the caller's stream; default: two unequal parts on two streams when the
# last round is partially filled
cd $GRAFT_REPO_ROOT
run() { python bench.py --workload $1 --states $2 --steps 30 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 n=$2 env=[$3]', 'ms', round(d['roofline']['kernel_ms'],4), 'ns/state', round(d['roofline']['kernel_ms']*1e6/$2,3))"; }
for n in 131072 200000 300000 500000 983040 1000000 1048576; do PJ_RBLK_SPLIT=0 run gri $n "split off"; run gri $n "default"; done
for n in 65536 100000 196608 200000 229376; do PJ_RBLK_SPLIT=0 run usc $n "split off"; run usc $n "default"; done
