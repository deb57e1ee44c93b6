#!/bin/bash
# round 5, first GPU session: USC-shaped row kernels -- phase map of the shipped geometry (-DPJQ_TIMING), the same kernels
# without any energy-row sum (-DPJQ_NO_E: results wrong, the bound of that lever) and with what a row block cannot see of its
# column summed by the pre-pass (PJQ_ECL), at two budgets, with and without the one-visit look-ahead
export TMPDIR=/tmp PJ_VAR_RATES=0
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
M=pyjac_amd/data/usc2_shaped.inp
timeout 900 python tools/rblk_variants.py time $M 200000 rblk noe noela ecl ecl40la ecl56la ecl56 > $O/r05_usc_variants.txt 2>&1
timeout 300 python tools/rblk_timing.py $M 200000 pyjac_amd/spec/var/usc2_shaped_tim.so > $O/r05_rblk_usc_phase_cycles.txt 2>&1
cat $O/r05_usc_variants.txt; cat $O/r05_rblk_usc_phase_cycles.txt
