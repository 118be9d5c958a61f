cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_acting.py -q -x > $O/pytest_acting.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_acting.log
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null
grep -E "policy_act|rollout_record" $O/kt/*kernel_stats.csv | cut -c1-200
python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value']); [print(r) for r in d['other_configs'] if r['workload']=='rollout']"
rm -rf $O/kt
