# round 6, VERDICT item 5: ceiling of de-synchronised bracketing from the -DPGTT_EFFORT build (alt_build/libpgtt_effort.so)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06g; mkdir -p $O
PGTT_LIB=$GRAFT_REPO_ROOT/alt_build/libpgtt_effort.so python tools/gpu_effort.py level4 150 2>&1 | grep -v amdgpu.ids | tee $O/effort.txt
