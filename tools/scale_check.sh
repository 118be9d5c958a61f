#!/bin/bash
# First-contact check of the multi-GPU path on a node with several MI355X (nothing in this repository has ever run RCCL with more than one rank:
# the build container has no GPU, the test box one).  Run it BEFORE the scaling sweep:
#
#     bash tools/scale_check.sh [max_gpus=8] [steps=300]
#
# 1. the node: GPUs visible, xGMI topology
# 2. the three RCCL tests of tests/test_gpu_multi.py (they skip themselves with fewer than two GPUs)
# 3. bench.py at N = 1, 2, 4, ... max_gpus, launched exactly like the driver does (one process per GPU, torch.distributed.run on 127.0.0.1):
#    every line must say n_gpus == N and count N x 4096 x steps env-steps through the all-reduce; prints the weak-scaling table
#    (aggregate env-steps/s, efficiency against N x the one-GPU figure) and each run's slowest / fastest rank (`ranks_dt`: a slow GCD shows here)
# 4. two iterations of train.py under torchrun on all GPUs (gradient / normaliser all-reduce over xGMI, rank-0 logging)
# Exit code 0 = every assertion held.  Output under gpurun_out/scale_check/ (or $OUT).
set -u
cd "$(dirname "$0")/.."
MAXG=${1:-8}; STEPS=${2:-300}; OUT=${OUT:-gpurun_out/scale_check}; mkdir -p "$OUT"
export TMPDIR=${TMPDIR:-/tmp} HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
fail=0
echo "== 1. node"; (rocm-smi --showtopo 2>/dev/null || echo "rocm-smi --showtopo unavailable") | tee "$OUT/topo.txt" | tail -40
HAVE=$(python -c "import torch; print(torch.cuda.device_count() if torch.cuda.is_available() else 0)")
echo "GPUs visible to torch: $HAVE"
if [ "$HAVE" -lt 1 ]; then echo "no GPU: nothing to check"; exit 2; fi
[ "$HAVE" -lt "$MAXG" ] && { echo "only $HAVE GPU(s): checking up to $HAVE"; MAXG=$HAVE; }

echo "== 2. RCCL tests"
python -m pytest tests/test_gpu_multi.py -q -m gpu 2>&1 | tail -3 | tee "$OUT/pytest_multi.txt"
grep -q "failed\|error" "$OUT/pytest_multi.txt" && fail=1

echo "== 3. bench.py, weak scaling"
NS=""; n=1; while [ "$n" -le "$MAXG" ]; do NS="$NS $n"; n=$((2 * n)); done
for N in $NS; do
  PORT=$((29700 + N))
  if [ "$N" -eq 1 ]; then CMD="python bench.py"; else CMD="python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT bench.py"; fi
  $CMD --gpus "$N" --steps "$STEPS" --warmup 30 --no-cpu-baseline --no-other-configs > "$OUT/bench_n$N.json" 2> "$OUT/bench_n$N.err"
  rc=$?; [ $rc -ne 0 ] && { echo "bench.py --gpus $N: exit code $rc"; tail -5 "$OUT/bench_n$N.err"; fail=1; }
done
python - "$OUT" "$STEPS" $NS <<'EOF' || fail=1
import json, sys
out, steps, ns = sys.argv[1], int(sys.argv[2]), [int(x) for x in sys.argv[3:]]
ok, base = True, None
print(f"{'N':>3} {'env-steps/s':>14} {'efficiency':>10} {'ms/step':>9} {'physics us':>10} {'slowest rank s':>15} {'fastest rank s':>15} {'spread':>7}")
for n in ns:
    try:
        d = json.loads([l for l in open(f"{out}/bench_n{n}.json") if l.startswith("{")][-1])
    except Exception as e:
        print(f"{n:>3} no JSON line ({e})"); ok = False; continue
    want = float(4096 * n * steps)
    good = d["n_gpus"] == n and d["env_steps_allreduced"] == want and "error" not in d and len(d["ranks_dt"]) == n
    base = base or d["value"]
    r = d["ranks_dt"]
    print(f"{n:>3} {d['value']:>14.0f} {d['value'] / (n * base):>10.3f} {d['ms_per_step']:>9.4f} {1e3 * d['kernels_ms']['physics_kernel']:>10.1f} {max(r):>15.5f} {min(r):>15.5f} {max(r) / min(r) - 1:>7.1%}"
          + ("" if good else f"   <-- n_gpus {d['n_gpus']}, env-steps {d['env_steps_allreduced']} (want {want}), {d.get('error', '')}"))
    ok &= good
sys.exit(0 if ok else 1)
EOF

echo "== 4. train.py, two iterations on $MAXG GPU(s)"
ENVS=$((4096 * MAXG))
if [ "$MAXG" -eq 1 ]; then TR="python train.py"; else TR="python -m torch.distributed.run --nnodes=1 --nproc-per-node $MAXG --master-addr 127.0.0.1 --master-port 29790 train.py"; fi
$TR --task_name stairs --terrain_file level4 --num_envs "$ENVS" --batch_size $((256 * MAXG)) --num_timesteps $((ENVS * 20 * 32 * 2)) --num_evals 2 --index 905 > "$OUT/train.txt" 2>&1
rc=$?; tail -4 "$OUT/train.txt"; [ $rc -ne 0 ] && { echo "train.py: exit code $rc"; fail=1; }
grep -q "time to train" "$OUT/train.txt" || { echo "train.py did not finish"; fail=1; }
[ $fail -eq 0 ] && echo "SCALE CHECK OK" || echo "SCALE CHECK FAILED"
exit $fail
