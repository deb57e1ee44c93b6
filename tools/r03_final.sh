#!/bin/bash
# final round-3 measurements of the shipped build: GPU suite, bench lines, LU probe
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03f; mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -s > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"; cut -c1-400 $O/bench_default.json
python bench.py --workload h2 --no-also > $O/bench_h2.json 2> $O/bench_h2.err; cut -c1-300 $O/bench_h2.json
python bench.py --workload usc --no-also > $O/bench_usc.json 2> $O/bench_usc.err; cut -c1-300 $O/bench_usc.json
python tools/lu_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/lu_probe.txt
