#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "linear_gemm" 2>&1 | tail -5
python scratch/fc_bench.py 2>&1 | grep -v amdgpu.ids
