cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04e; mkdir -p $O
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $O/pmc -o p --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc | grep -E "policy_act|rollout_record" 
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $O/pmc2 -o p --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1
python tools/pmc_summary.py $O/pmc2 | grep -E "policy_act"
for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --pmc $c -d $O/pmc_$c -o p --output-format csv -- python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > /dev/null 2>&1; python tools/pmc_summary.py $O/pmc_$c | grep -E "policy_act"; done
rm -rf $O/pmc*
