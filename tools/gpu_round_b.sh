#!/bin/bash
set -x
mkdir -p gpurun_out/rb
timeout 900 python -m pytest tests/test_gpu_evaluator.py -m gpu -x -q > gpurun_out/rb/pytest_eval.txt 2>&1
tail -5 gpurun_out/rb/pytest_eval.txt
timeout 600 python tools/lola_latency.py > gpurun_out/rb/lola.txt 2>&1
tail -8 gpurun_out/rb/lola.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rb/bench.json 2> gpurun_out/rb/bench.err
cat gpurun_out/rb/bench.json | cut -c1-200
timeout 900 python tools/cifar_latency.py > gpurun_out/rb/cifar.txt 2>&1
tail -6 gpurun_out/rb/cifar.txt
