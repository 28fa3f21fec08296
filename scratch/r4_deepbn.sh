#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "finalizes_the_input or deep_kernels or odd_batches" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_vae.py -x -q -m gpu -k "parity and 8l" 2>&1 | tail -3
bash scratch/ab.sh 2 --arch speccnn8l1_bn --steps 30 --warmup 5
bash scratch/ab.sh 1 --arch speccnn8l1_bn --dim-z 512 --dtype bf16 --steps 30 --warmup 5
