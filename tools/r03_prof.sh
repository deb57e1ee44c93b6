#!/bin/bash
# per-kernel durations, SQ counters and HBM traffic (separate --pmc passes) of the pj_rblk libraries
# usage: r03_prof_rblk.sh <gri|usc>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
W=${1:-gri}
if [ $W = gri ]; then MECH=$R/pyjac_amd/data/gri30_shaped.inp; N=1000000; NP=262144; BPS=22904; LBL="GRI-shaped 53sp, pj_rblk, 262144 states";
else MECH=$R/pyjac_amd/data/usc2_shaped.inp; N=200000; NP=65536; BPS=99464; LBL="USC-shaped 111sp, pj_rblk, 65536 states"; fi
cd /tmp
# the bench command itself under --kernel-trace --stats (kernel average durations)
timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03_kt_$W --output-format csv -- python $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/r03_kt_$W.log 2>&1
cp $(ls $O/r03_kt_$W/*/*kernel_stats.csv | head -1) $O/r03_rblk_${W}_kernel_stats.csv
# the kernels of the two parts of a batch overlap (two streams): step time = union of their intervals
python $R/tools/trace_span.py $(ls $O/r03_kt_$W/*/*kernel_trace.csv | head -1) 2 "$W-shaped, pj_rblk, bench.py --workload $W --steps 20 (full batch)" > $O/r03_rblk_${W}_step_span.json 2>&1
# and the same command with the batch as one part on the caller's stream: kernels back to back
PJ_RBLK_SPLIT=0 timeout 600 rocprofv3 --kernel-trace --stats -d $O/r03_kt1_$W --output-format csv -- python $R/bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-also > $O/r03_kt1_$W.log 2>&1
cp $(ls $O/r03_kt1_$W/*/*kernel_stats.csv | head -1) $O/r03_rblk_${W}_onepart_kernel_stats.csv
tail -1 $O/r03_kt1_$W.log | cut -c1-300
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/r03_pmc_$name --output-format csv -- python $R/tools/one_step.py $MECH $NP 2 rblk > $O/r03_pmc_$name.log 2>&1; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
pass c FETCH_SIZE
pass d WRITE_SIZE
# the rate pass (k_rate: every array / dydt only): per-kernel durations and SQ counters
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r03_kr_$W --output-format csv -- python $R/tools/one_step.py $MECH $N 10 rblk rates > $O/r03_kr_$W.log 2>&1
cp $(ls $O/r03_kr_$W/*/*kernel_stats.csv | head -1) $O/r03_rates_${W}_all_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r03_kd_$W --output-format csv -- python $R/tools/one_step.py $MECH $N 10 rblk dydt > $O/r03_kd_$W.log 2>&1
cp $(ls $O/r03_kd_$W/*/*kernel_stats.csv | head -1) $O/r03_rates_${W}_dydt_kernel_stats.csv
passr() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/r03_pmc_$name --output-format csv -- python $R/tools/one_step.py $MECH $NP 2 rblk rates > $O/r03_pmc_$name.log 2>&1; }
passr ra SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
passr rb SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
cd $R
python tools/pmc_summarize.py $O/r03_pmc_ra $O/r03_pmc_rb > $O/r03_rates_${W}_sq_counters.json 2>&1
python tools/pmc_summarize.py $O/r03_pmc_a $O/r03_pmc_b > $O/r03_rblk_${W}_sq_counters.json 2>&1
python tools/traffic_pmc.py $O/r03_pmc_c $O/r03_pmc_d 2 $NP $BPS "$LBL" > $O/traffic_$W.json 2>&1
grep '"Name"\|k_rblk\|k_pre' $O/r03_rblk_${W}_kernel_stats.csv | cut -c1-200
tail -3 $O/r03_kt_$W.log | cut -c1-600
cat $O/traffic_$W.json | head -30
cat $O/r03_rblk_${W}_step_span.json
grep 'k_rate' $O/r03_rates_${W}_all_kernel_stats.csv $O/r03_rates_${W}_dydt_kernel_stats.csv | cut -c1-200
rm -rf $O/r03_pmc_a $O/r03_pmc_b $O/r03_pmc_c $O/r03_pmc_d $O/r03_pmc_ra $O/r03_pmc_rb $O/r03_kt_$W $O/r03_kt1_$W $O/r03_kr_$W $O/r03_kd_$W
