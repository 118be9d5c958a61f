# round 6, closing check: the whole -m gpu suite, smoke(), the driver's command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06m; mkdir -p $O
python -m pytest tests -q -m gpu 2>&1 | tail -6 | tee $O/gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $O/smoke.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_bench.json 2>$O/driver_cmd_bench.err; echo "driver rc=$?"; python -c "
import json; d=json.load(open('$O/driver_cmd_bench.json')); r=d['roofline']; print(d['value'], d['ms_per_step'], d['warmup'], r['frac'], r['frac_unpacked_ceiling'], r['traffic'], r['valu_busy'], r['valu_packed_share'], r['profile_stale'], d['cpu_baseline']['value'])"
