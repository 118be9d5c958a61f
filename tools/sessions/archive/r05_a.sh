# round 5, first GPU call: A/B of the guard changes against HEAD (both correctly rounded), the whole -m gpu suite on the new product
# (correctly rounded `/` and sqrt), smoke, the driver's command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05a; mkdir -p $O
PGTT_AB_OCT=1 python tools/gpu_ab_bitwise.py alt_build/libpgtt_ref.so phase_guided_terrain_traversal_amd/libpgtt.so 40 2>&1 | grep -v amdgpu.ids | tee $O/ab_bitwise.txt | tail -9
timeout 2400 python -m pytest tests -q -m gpu -x --durations=15 > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -25 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_bench.json 2>$O/driver_cmd_bench.err ) 2>&1 | grep real; tail -3 $O/driver_cmd_bench.err; python -c "
import json; d=json.load(open('$O/driver_cmd_bench.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['profile_stale'], d['warmup'], d['ranks_dt']); print(d['cpu_baseline']['value'], d['cpu_baseline']['per_core']); [print(r['workload'], r['envs'], r.get('level',''), round(r.get('value',0)/1e6,2), r.get('kernels_ms'), r.get('skipped')) for r in d['other_configs']]"
