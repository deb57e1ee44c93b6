#!/bin/bash
# round 5: per-kernel durations, SQ counters and HBM traffic (separate --pmc passes) of the pj_rblk libraries
# usage: r05_prof.sh <gri|usc>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
W=${1:-gri}
if [ $W = gri ]; then MECH=$R/pyjac_amd/data/gri30_shaped.inp; N=1000000; NP=262144; BPS=22904; LBL="GRI-shaped 53sp, pj_rblk (one row kernel), 262144 states"; KL="k_rblk:1,k_pre:1";
else MECH=$R/pyjac_amd/data/usc2_shaped.inp; N=200000; NP=65536; BPS=99464; LBL="USC-shaped 111sp, pj_rblk, 65536 states"; KL="k_rblk:${2:-6},k_pre:1"; fi
cd /tmp
# the bench command itself under --kernel-trace --stats (kernel average durations); PJ_RBLK_SPLIT=0: kernels back to back
PJ_RBLK_SPLIT=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/r05_kt_$W --output-format csv -- python $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/r05_kt_$W.log 2>&1
cp $(ls $O/r05_kt_$W/*/*kernel_stats.csv | head -1) $O/r05_rblk_${W}_kernel_stats.csv
tail -1 $O/r05_kt_$W.log > $O/r05_rblk_${W}_bench_line_under_rocprof.json
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/r05_pmc_$name --output-format csv -- python $R/tools/one_step.py $MECH $NP 2 rblk > $O/r05_pmc_$name.log 2>&1; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
pass c FETCH_SIZE
pass d WRITE_SIZE
pass e SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES
cd $R
python tools/pmc_summarize.py $O/r05_pmc_a $O/r05_pmc_b > $O/r05_rblk_${W}_sq_counters.json 2>&1
python tools/pmc_summarize.py $O/r05_pmc_e > $O/r05_rblk_${W}_lds_counters.json 2>&1
python tools/traffic_pmc.py $O/r05_pmc_c $O/r05_pmc_d 2 $NP $BPS "$LBL" > $O/traffic_$W.json 2>&1
python tools/valu_roof.py $O/r05_rblk_${W}_sq_counters.json $NP $KL "profiles/r05_rblk_${W}_sq_counters.json (rocprofv3 --pmc SQ_* over tools/one_step.py, $NP states)" > $O/valu_$W.json 2>&1
grep '"Name"\|k_rblk\|k_pre' $O/r05_rblk_${W}_kernel_stats.csv | cut -c1-200
cut -c1-700 $O/r05_rblk_${W}_bench_line_under_rocprof.json
head -12 $O/traffic_$W.json; tail -4 $O/traffic_$W.json
head -8 $O/valu_$W.json
rm -rf $O/r05_pmc_a $O/r05_pmc_b $O/r05_pmc_c $O/r05_pmc_d $O/r05_pmc_e $O/r05_kt_$W
