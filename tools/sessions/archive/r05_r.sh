cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05r; mkdir -p $O
python tools/gpu_model_bisect.py stairs hex 2>&1 | grep "^==" | tee $O/bisect.txt
