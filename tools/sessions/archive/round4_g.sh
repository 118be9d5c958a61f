cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "16384" 2>&1 | grep -E "n=16384|env-steps in W|passed|failed|Error" | cut -c1-420
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value']); [print(r['workload'], r['envs'], round(r.get('value',0)/1e6,2), r.get('kernels_ms')) for r in d['other_configs']]"
for A in "--envs 16384" "--envs 32768" "--workload wfc_dr --envs 16384" "--workload flat --envs 16384"; do python bench.py $A --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; print('%-40s %.3f M  physics %.2f us  observe %.2f us' % ('$A', d['value']/1e6, 1e3*k['physics_kernel'], 1e3*k['observe_kernel']))"; done
