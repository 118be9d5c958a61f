# where do the cycles of physics_kernel that are not VALU issue go?  memory instructions in flight, scalar / branch counts, instruction
# cache (counter passes only, five runs of the short bench; run through gpurun:  bash tools/pmc_waits.sh) -> profiles/archive/r02e_physics_waits.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/waits; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS}"
rocprofv3 --pmc SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS -d $O/a -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_ANY -d $O/b -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_VSKIPPED SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $O/c -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ -d $O/d -o p --output-format csv -- $B > /dev/null 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY -d $O/e -o p --output-format csv -- $B > /dev/null 2>&1
python tools/pmc_summary.py $O/a $O/b $O/c $O/d $O/e | grep "physics_kernel<0"
