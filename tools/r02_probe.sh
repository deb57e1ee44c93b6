#!/bin/bash
# round-2 first probe: baseline tests + GRI bench + instruction-fetch / stall counters of pj_rows
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/r02_pytest0.log 2>&1; echo "pytest rc=$?" >> $O/r02_pytest0.log
timeout 300 python bench.py --workload gri --steps 5 --warmup 1 --no-also --no-cpu-baseline > $O/r02_bench_gri0.log 2>&1
GRI=$R/pyjac_amd/data/gri30_shaped.inp
cd /tmp
rocprofv3 -L > $O/r02_counters.txt 2>&1
pass() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/r02_pmc_$name --output-format csv -- python $R/tools/one_step.py $GRI 262144 2 rows > $O/r02_pmc_$name.log 2>&1; }
pass a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass b SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
pass c SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE
pass d SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
cd $R
python tools/pmc_summarize.py $O/r02_pmc_a $O/r02_pmc_b $O/r02_pmc_c $O/r02_pmc_d > $O/r02_pmc_summary.json 2>&1
# keep only summaries (the raw csv trees are big)
rm -rf $O/r02_pmc_a $O/r02_pmc_b $O/r02_pmc_c $O/r02_pmc_d
