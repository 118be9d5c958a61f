cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06h; mkdir -p $O
python tools/gpu_model_switch_relevance.py 1024 100 2>&1 | grep -v amdgpu.ids | tee $O/model_switch_relevance.txt
