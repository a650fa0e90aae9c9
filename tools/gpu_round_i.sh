#!/bin/bash
mkdir -p gpurun_out/ri
python tools/ks_probe.py > gpurun_out/ri/ks.txt 2>&1; cat gpurun_out/ri/ks.txt
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py -m gpu -x -q 2>&1 | tail -3
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/ri/bench.json 2> gpurun_out/ri/bench.err
cut -c1-220 gpurun_out/ri/bench.json
timeout 900 python tools/cifar_latency.py > gpurun_out/ri/cifar.txt 2>&1
tail -3 gpurun_out/ri/cifar.txt | cut -c1-250
