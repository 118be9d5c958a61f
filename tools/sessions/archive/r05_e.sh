# round 5: per-substep Newton-trip histogram on the -DPGTT_EFFORT build of the hex terrain kernel
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05e; mkdir -p $O
PGTT_LIB=$PWD/alt_build/libpgtt_effort.so python tools/gpu_effort.py level4 150 2>&1 | grep -v amdgpu.ids | tee $O/effort_level4.txt
PGTT_LIB=$PWD/alt_build/libpgtt_effort.so python tools/gpu_effort.py flat 150 2>&1 | grep -v amdgpu.ids | tee $O/effort_flat.txt
