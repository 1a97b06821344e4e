#!/bin/bash
# bash start.sh $MODEL $DATASET [$GPU]  - same dispatch as the reference's src/start.sh:3-11, run from src/.
# HIP_VISIBLE_DEVICES replaces CUDA_VISIBLE_DEVICES; SRGNN gets the script the reference forgot to ship.
GPU=${3:-0}
export HIP_VISIBLE_DEVICES=$GPU
if [[ $1 == NISER ]]; then
    python -u scripts/main_niser.py --dataset-dir ../datasets/$2
elif [[ $1 == SRGNN ]]; then
    python -u scripts/main_srgnn.py --dataset-dir ../datasets/$2
elif [[ $1 == LESSR ]]; then
    python -u scripts/main_lessr.py --dataset-dir ../datasets/$2 --num-layers 1
elif [[ $1 == MSGIFSR ]]; then
    python -u scripts/main_msgifsr.py --dataset-dir ../datasets/$2 --num-layers 1 --order 1
fi
