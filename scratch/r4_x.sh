#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "bf16 or bwd_fuse or linear" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_vae.py -x -q -m gpu -k "bf16 or dz512 or b256 or eager_outputs or survive" 2>&1 | tail -3
bash scratch/ab.sh 3 --arch speccnn8l1_bn --dim-z 512 --dtype bf16 --steps 30 --warmup 5
