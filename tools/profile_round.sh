# One measurement round on the GPU box (run through gpurun):   ROUND_TAG=r02b bash tools/profile_round.sh
# bench lines (the driver's exact command first, its stdout KEPT), rocprofv3 kernel trace of the default bench, PMC passes (separate runs,
# counters only) -> gpurun_out/$ROUND_TAG/ ; tools/collect_profiles.py copies the summaries into profiles/
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/${ROUND_TAG:-r02}; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_bench.json 2>$O/driver_cmd_bench.err; echo "driver command rc=$?"; cat $O/driver_cmd_bench.json
python bench.py --no-other-configs > $O/bench_level4.json 2>$O/bench_level4.err
python bench.py --workload flat --no-cpu-baseline --no-other-configs > $O/bench_flat.json 2>/dev/null
python bench.py --workload wfc_dr --envs 8192 --no-cpu-baseline --no-other-configs > $O/bench_wfc_dr_8192.json 2>/dev/null
python bench.py --layout quad --no-cpu-baseline --no-other-configs > $O/bench_level4_quad.json 2>/dev/null
python bench.py --layout quad --workload wfc_dr --envs 8192 --no-cpu-baseline --no-other-configs > $O/bench_wfc_dr_8192_quad.json 2>/dev/null
python bench.py --envs 8192 --no-cpu-baseline --no-other-configs > $O/bench_level4_8192.json 2>/dev/null
python bench.py --envs 32768 --no-cpu-baseline --no-other-configs > $O/bench_level4_32768.json 2>/dev/null
python bench.py --workload flat --envs 16384 --no-cpu-baseline --no-other-configs > $O/bench_flat_16384.json 2>/dev/null
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python bench.py --no-cpu-baseline --no-other-configs > $O/kt_bench.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c -d $O/pmc_$c -o p --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > /dev/null 2>&1; done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $O/pmc_sq -o p --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_mfma -o p --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_sq $O/pmc_mfma > $O/pmc_summary.txt
# the same counters for the other single-GPU configs (BASELINE configs[1], configs[3]) and for level4 with the variants in draw order
for W in "flat:--workload flat" "wfc_dr_8192:--workload wfc_dr --envs 8192" "level4_unsorted:--unsorted-variants"; do
  T=${W%%:*}; A=${W#*:}
  for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c -d $O/pmc_${T}_$c -o p --output-format csv -- python bench.py $A --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > /dev/null 2>&1; done
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc_${T}_sq -o p --output-format csv -- python bench.py $A --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
  python tools/pmc_summary.py $O/pmc_${T}_FETCH_SIZE $O/pmc_${T}_WRITE_SIZE $O/pmc_${T}_sq > $O/pmc_summary_$T.txt
done
cp $O/kt/*kernel_stats.csv $O/kernel_stats.csv
# round 4: the caller of the hot path on the record - kernel trace of the driver's command WITH its other_configs rows (the 'rollout' row: policy_act_kernel,
# rollout_record_kernel next to the two env kernels) and of a short train.py run (acting step + PPO update)
rocprofv3 --kernel-trace --stats -d $O/kt_rollout -o kt --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/kt_rollout_bench.json 2>/dev/null
cp $O/kt_rollout/*kernel_stats.csv $O/train_rollout_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/kt_train -o kt --output-format csv -- python train.py --task_name stairs --terrain_file level4 --num_envs 4096 --num_timesteps 6553600 --num_evals 5 --index 904 > $O/train_run.txt 2>&1
cp $O/kt_train/*kernel_stats.csv $O/train_ppo_kernel_stats.csv; tail -5 $O/train_run.txt
for f in $O/bench_*.json $O/driver_cmd_bench.json $O/kt_bench.json; do python -c "import sys,json; d=json.load(open('$f')); print('$f', d['value'], d['ms_per_step'], d['kernels_ms'], d.get('wall_over_kernels'))"; done
grep "physics_kernel<0" $O/pmc_summary.txt | grep -E "FETCH|WRITE|WAVE_CYCLES|WAIT_ANY|ACTIVE_INST_VALU|INSTS_|SQ_WAVES"
head -3 $O/kernel_stats.csv | cut -c1-150
# the merged gpurun_out/ may hold 64 MiB: the raw traces stay on the box, the summaries travel
rm -rf $O/kt $O/kt_rollout $O/kt_train $O/pmc_*/ 2>/dev/null; du -sh $O
