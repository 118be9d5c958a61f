#!/bin/bash
# Mirror of the reference's training/training.sh:31-58: a three-stage terrain curriculum, every stage resuming from the checkpoint of
# the one before (train.py --checkpoint_folder).  METHOD=pgtt|baseline, STEPS per stage, three checkpoint indexes, three terrain files
# (the reference: level4 -> level7 / level10 -> level10 / level13).  With GPUS > 1 every stage is a data-parallel torchrun job.
METHOD=${METHOD:-pgtt}
STEPS=${STEPS:-300000000}
INDEXES=(${INDEXES:-153 154 155})
LEVELS=(${LEVELS:-level4 level10 level13})
GPUS=${GPUS:-1}
cd "$(dirname "$0")"
if [ "$GPUS" -gt 1 ]; then RUN="python -m torch.distributed.run --nnodes=1 --nproc-per-node $GPUS --master-addr 127.0.0.1 --master-port ${MASTER_PORT:-29555} train.py"; else RUN="python train.py"; fi
for i in 0 1 2; do rm -rf "checks_stairs/checkpoint_${INDEXES[$i]}"; done
$RUN --method "$METHOD" --index "${INDEXES[0]}" --terrain_file "${LEVELS[0]}" --num_timesteps "$STEPS" || exit 1
$RUN --method "$METHOD" --index "${INDEXES[1]}" --checkpoint_folder "checks_stairs/checkpoint_${INDEXES[0]}" --terrain_file "${LEVELS[1]}" --num_timesteps "$STEPS" || exit 1
$RUN --method "$METHOD" --index "${INDEXES[2]}" --checkpoint_folder "checks_stairs/checkpoint_${INDEXES[1]}" --terrain_file "${LEVELS[2]}" --num_timesteps "$STEPS" || exit 1
