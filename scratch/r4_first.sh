#!/bin/bash
# round 4, first GPU call: new host-side features + micro-benchmark + copy-node hunt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./scratch/ubench/ws_share32 > gpurun_out/ws_share32.txt 2>&1
python scratch/graph_dot.py > gpurun_out/graph_dot.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_distributed.py tests/test_gpu_vae.py -x -q -m gpu -k "distributed or two_ranks or bucket_graph or two_graph or eager_outputs or consecutive_forwards or synth_params_loss or eager_steps or graph_mode or graph_replay" > gpurun_out/tests_first.txt 2>&1
F="--no-extra --no-cpu-baseline --no-roofline --steps 200 --warmup 20"
for mode in "" "--force-dist --dist-mode bucket-graphs" "--force-dist --dist-mode two-graph" "--force-dist --dist-mode eager" "" "--force-dist --dist-mode bucket-graphs"; do
  python bench.py $F $mode 2>gpurun_out/bench_err.txt | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('MODE [$mode]', d['ms_per_step'], d['config'].get('launch'), d['config'].get('collective_launches_per_step'))"
done > gpurun_out/modes.txt 2>&1
tail -5 gpurun_out/tests_first.txt; cat gpurun_out/modes.txt; head -40 gpurun_out/ws_share32.txt; tail -30 gpurun_out/graph_dot.txt
