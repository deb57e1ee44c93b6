#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_lu.py -x -q -m gpu 2>&1 | tail -15 > $O/r05_lu_tests.txt
timeout 600 python tools/lu_probe.py 65:200000 80:200000 96:200000 111:200000 128:100000 140:100000 > $O/r05_lu_probe.txt 2>&1
cat $O/r05_lu_tests.txt $O/r05_lu_probe.txt
