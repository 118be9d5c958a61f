# round 4, GPU call B: the fused acting step (tests, bench rollout row), multi-rank readiness tests on this box
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04b; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_acting.py tests/test_gpu_multi.py tests/test_gpu_train.py -q -s -k "train_alike or faster_than or multi or single_rank or improves" > $O/pytest_acting.log 2>&1; echo "pytest rc=$?"; tail -25 $O/pytest_acting.log | cut -c1-300


