# round 5: large-sample parity of the correctly rounded product, three layouts (1024 envs x 100 control steps on level4; 512 x 80 on level13 + DR for hex / oct)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05l; mkdir -p $O
timeout 2400 python tools/gpu_big_parity.py 2>&1 | grep -v amdgpu.ids | tee $O/parity_big.txt | grep -E "OK|Error|assert" | cut -c1-400
