# round 5: soak of the correctly rounded build (three layouts x {level13 + DR, level4}, 3000 steps; quad by auto at 16384 / 32768 envs, 2000 steps) and the final measurement round
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05i; mkdir -p $O
python tools/gpu_soak.py 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt
python tools/gpu_soak_big.py 2>&1 | grep -v amdgpu.ids | tee -a $O/soak.txt
