#!/bin/bash
# per-kernel durations and SQ counters of the pj_rblk library of the GRI-shaped mechanism
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
GRI=$R/pyjac_amd/data/gri30_shaped.inp
KIND=${1:-rblk}; TAG=${2:-rblk}
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/r02_kt_$TAG --output-format csv -- python $R/tools/one_step.py $GRI 1000000 5 $KIND > $O/r02_kt_$TAG.log 2>&1
cp $(ls $O/r02_kt_$TAG/*/*kernel_stats.csv | head -1) $O/r02_${TAG}_gri_kernel_stats.csv
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/r02_pmc_$name --output-format csv -- python $R/tools/one_step.py $GRI 262144 2 $KIND > $O/r02_pmc_$name.log 2>&1; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
pass c FETCH_SIZE
pass d WRITE_SIZE
cd $R
python tools/pmc_summarize.py $O/r02_pmc_a $O/r02_pmc_b $O/r02_pmc_c $O/r02_pmc_d > $O/r02_${TAG}_pmc_summary.json 2>&1
python tools/traffic_pmc.py $O/r02_pmc_c $O/r02_pmc_d 2 262144 22904 "GRI-shaped 53sp, pj_$TAG, 262144 states" > $O/r02_traffic_gri_$TAG.json 2>&1
rm -rf $O/r02_pmc_a $O/r02_pmc_b $O/r02_pmc_c $O/r02_pmc_d $O/r02_kt_$TAG
