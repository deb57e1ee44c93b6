export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in tests/golden/sweep/sweep_r5.inp tests/golden/synth_mid24.inp; do echo "== $m"; PJ_VAR_RATES=0 timeout 600 python tools/rblk_variants.py time $m 1000000 base h2 h2b80 base h2 h2b80 2>&1 | grep -v amdgpu; done > gpurun_out/r06_small_variants_m2.txt; cat gpurun_out/r06_small_variants_m2.txt
