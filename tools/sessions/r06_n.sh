cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06n; mkdir -p $O
python tools/gpu_policy_family_stats.py 2>&1 | grep -v amdgpu.ids | tee $O/policy_family.txt
