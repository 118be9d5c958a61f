# round 5, final check: the whole -m gpu suite, smoke, the driver's command
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r05v; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu --durations=8 > $O/pytest_gpu.log 2>&1; echo "pytest -m gpu rc=$?"; tail -14 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_bench.json 2>$O/driver_cmd_bench.err ) 2>&1 | grep real; python -c "
import json; d=json.load(open('$O/driver_cmd_bench.json')); print(d['value'], d['ms_per_step'], d['kernels_ms'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['profile_stale'], d['warmup']); [print(r['workload'], r['envs'], r.get('level',''), round(r.get('value',0)/1e6,2), r.get('skipped')) for r in d['other_configs']]"
