#!/bin/bash
# round 6: per-kernel durations, SQ counters and HBM traffic (separate --pmc passes) of the shipped pj_rblk libraries, with the
# library's file name (= digest of kernel sources + build options) stored next to the counters
# usage: r06_prof.sh <gri|usc|h2>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
W=${1:-gri}
KIND=rblk
if [ $W = gri ]; then MECH=$R/pyjac_amd/data/gri30_shaped.inp; NP=262144; BPS=22904; LBL="GRI-shaped 53sp, pj_rblk (one row kernel), 262144 states"; KL="k_rblk:1,k_pre:1";
elif [ $W = usc ]; then MECH=$R/pyjac_amd/data/usc2_shaped.inp; NP=65536; BPS=99464; LBL="USC-shaped 111sp, pj_rblk, 65536 states"; KL="k_rblk:${2:-6},k_pre:1";
else MECH=$R/pyjac_amd/data/h2o2_n2.inp; NP=1048576; BPS=888; LBL="H2/O2+N2 10sp, pj_lane, 1048576 states"; KL="k_lane:1"; KIND=lane; fi
cd /tmp
PJ_RBLK_SPLIT=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/r06_kt_$W --output-format csv -- python $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/r06_kt_$W.log 2>&1
cp $(ls $O/r06_kt_$W/*/*kernel_stats.csv | head -1) $O/r06_${KIND}_${W}_kernel_stats.csv
tail -1 $O/r06_kt_$W.log > $O/r06_${KIND}_${W}_bench_line_under_rocprof.json
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/r06_pmc_$name --output-format csv -- python $R/tools/one_step.py $MECH $NP 2 $KIND > $O/r06_pmc_$name.log 2>&1; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
pass c FETCH_SIZE
pass d WRITE_SIZE
LIB=$(grep -o "library [^ ]*" $O/r06_pmc_a.log | tail -1 | cut -d' ' -f2)
cd $R
python tools/pmc_summarize.py $O/r06_pmc_a $O/r06_pmc_b > $O/r06_${KIND}_${W}_sq_counters.json 2>&1
python tools/traffic_pmc.py $O/r06_pmc_c $O/r06_pmc_d 2 $NP $BPS "$LBL" "$LIB" > $O/traffic_$W.json 2>&1
python tools/valu_roof.py $O/r06_${KIND}_${W}_sq_counters.json $NP $KL "profiles/r06_${KIND}_${W}_sq_counters.json (rocprofv3 --pmc SQ_* over tools/one_step.py, $NP states)" "$LIB" > $O/valu_$W.json 2>&1
grep '"Name"\|k_rblk\|k_pre\|k_lane' $O/r06_${KIND}_${W}_kernel_stats.csv | cut -c1-200
cut -c1-500 $O/r06_${KIND}_${W}_bench_line_under_rocprof.json
head -4 $O/traffic_$W.json; grep ratio $O/traffic_$W.json
head -8 $O/valu_$W.json
rm -rf $O/r06_pmc_a $O/r06_pmc_b $O/r06_pmc_c $O/r06_pmc_d $O/r06_kt_$W
