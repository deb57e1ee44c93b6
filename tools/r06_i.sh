#!/bin/bash
# round 6, session I: which wait class grows when the Jacobian stores flow?  SQ / SQC counters of the shipped GRI-shaped
# one-kernel build next to its -DPJQ_NO_STORE twin (separate --pmc passes over tools/one_step.py, 262144 states)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
GRI=$R/pyjac_amd/data/gri30_shaped.inp; NP=262144
V=$R/pyjac_amd/spec/var
cd /tmp
pass() { tag=$1; name=$2; shift; shift; PJ_ONE_STEP_LIB=$LIB timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d $O/r06i_${tag}_$name --output-format csv -- python $R/tools/one_step.py $GRI $NP 2 rblk > $O/r06i_${tag}_$name.log 2>&1; }
for tag in store nostore; do
  if [ $tag = store ]; then LIB=$(ls $R/pyjac_amd/spec/libpj_rblk_54c47061081581f0_*.so | head -1); else LIB=$V/gri30_shaped_nostore.so; fi
  pass $tag a SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_BUSY_CYCLES
  pass $tag b SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
  pass $tag c SQC_TC_INST_REQ SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQC_ICACHE_INPUT_VALID_READYB SQC_ICACHE_REQ SQC_ICACHE_MISSES SQ_IFETCH SQ_IFETCH_LEVEL
  pass $tag d SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_SMEM
  python $R/tools/pmc_summarize.py $O/r06i_${tag}_a $O/r06i_${tag}_b $O/r06i_${tag}_c $O/r06i_${tag}_d > $O/r06_rblk_gri_waitclass_$tag.json 2>&1
  rm -rf $O/r06i_${tag}_a $O/r06i_${tag}_b $O/r06i_${tag}_c $O/r06i_${tag}_d
done
python - <<'PY'
import json,os
O=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/'
a=json.load(open(O+'r06_rblk_gri_waitclass_store.json'))['k_rblk']; b=json.load(open(O+'r06_rblk_gri_waitclass_nostore.json'))['k_rblk']
for k in sorted(a):
    print('%-34s store %14.0f   nostore %14.0f   ratio %.3f' % (k, a[k]['mean'], b.get(k,{'mean':0})['mean'], a[k]['mean']/max(b.get(k,{'mean':0})['mean'],1)))
PY
