#!/bin/bash
# round 5: SQ counters of k_jvd (variant library <tag> of pyjac_amd/spec/var): r05_jvd_prof.sh <gri|usc> <tag>
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
W=${1:-usc}; TAG=${2:-jvd}
if [ $W = gri ]; then MECH=$R/pyjac_amd/data/gri30_shaped.inp; N=1000000; else MECH=$R/pyjac_amd/data/usc2_shaped.inp; N=200000; fi
cd /tmp
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/jvd_pmc_$name --output-format csv -- python $R/tools/jv_time.py $MECH $N $TAG > $O/jvd_pmc_$name.log 2>&1; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS
pass e SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES
timeout 300 rocprofv3 --kernel-trace --stats -d $O/jvd_kt --output-format csv -- python $R/tools/jv_time.py $MECH $N $TAG > $O/jvd_kt.log 2>&1
cd $R
python tools/pmc_summarize.py $O/jvd_pmc_a $O/jvd_pmc_b > $O/r05_jvd_${W}_${TAG}_sq_counters.json 2>&1
python tools/pmc_summarize.py $O/jvd_pmc_e > $O/r05_jvd_${W}_${TAG}_lds_counters.json 2>&1
cp $(ls $O/jvd_kt/*/*kernel_stats.csv | head -1) $O/r05_jvd_${W}_${TAG}_kernel_stats.csv
grep '"Name"\|k_jvd' $O/r05_jvd_${W}_${TAG}_kernel_stats.csv | cut -c1-200
cat $O/r05_jvd_${W}_${TAG}_sq_counters.json | head -60
cat $O/r05_jvd_${W}_${TAG}_lds_counters.json | head -40
rm -rf $O/jvd_pmc_a $O/jvd_pmc_b $O/jvd_pmc_e $O/jvd_kt
