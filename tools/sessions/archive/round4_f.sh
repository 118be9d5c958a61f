# round 4, GPU call F: many-box pass timing, soak of the three layouts, a stairs training run with the fused acting step + its evaluation, then the measurement round
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r04f; mkdir -p $O
python tools/gpu_manybox_time.py 2>&1 | grep -v amdgpu.ids > $O/manybox_time.txt; cat $O/manybox_time.txt
timeout 900 python tools/gpu_soak.py 2>&1 | grep -v amdgpu.ids > $O/soak.txt; tail -8 $O/soak.txt
( time python train.py --method pgtt --task_name stairs --terrain_file level1 --num_envs 4096 --num_timesteps 120000000 --num_evals 13 --index 905 ) > $O/train_level1.txt 2>&1; grep -E "^steps|time to train|real" $O/train_level1.txt | tail -16
python evaluate.py --method pgtt --task_name stairs --terrain_file level1 --checkpoint_folder checks_stairs/checkpoint_905 2>&1 | grep -v amdgpu.ids | tail -4 > $O/eval_level1.txt; cat $O/eval_level1.txt
python evaluate.py --method pgtt --task_name stairs --terrain_file level4 --checkpoint_folder checks_stairs/checkpoint_905 2>&1 | grep -v amdgpu.ids | tail -2 >> $O/eval_level1.txt; tail -2 $O/eval_level1.txt
ROUND_TAG=r04b bash tools/profile_round.sh > $O/round.log 2>&1; tail -4 $O/round.log
