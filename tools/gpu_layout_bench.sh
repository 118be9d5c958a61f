#!/bin/bash
# physics / observe kernel times and env-steps/s of bench.py per lane layout:  tools/gpu_layout_bench.sh ENVS WORKLOAD LAYOUT...
n=$1; w=$2; shift 2
for lay in "$@"; do
  python bench.py --layout $lay --no-other-configs --envs $n --workload $w --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lay', '$w', $n, 'envs: %.2f M env-steps/s' % (d['value']/1e6), '%.4f ms/step' % d['ms_per_step'], d.get('kernels_ms'))"
done
