cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06d; mkdir -p $O
for L in hex oct quad; do python tools/gpu_explain_case.py alt_build/cases/test_level4_parity_hex_step33.npz $L 2>&1 | grep -v amdgpu.ids; done | tee $O/case.txt
