# round 4, GPU call H: the whole -m gpu suite on the final kernels, smoke, then the measurement round r04c
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04h; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -4 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
ROUND_TAG=r04c bash tools/profile_round.sh > $O/round.log 2>&1; tail -3 $O/round.log
for A in "--envs 16384" "--workload wfc_dr --envs 16384"; do python bench.py $A --no-cpu-baseline --no-other-configs > gpurun_out/r04c/bench_$(echo $A | tr -d ' -').json 2>/dev/null; done; ls gpurun_out/r04c | head -40
