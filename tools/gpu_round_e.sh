#!/bin/bash
mkdir -p gpurun_out/re
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_lola_cifar.py -m gpu -x -q > gpurun_out/re/pytest.txt 2>&1
tail -8 gpurun_out/re/pytest.txt
timeout 900 python tools/cifar_latency.py > gpurun_out/re/cifar.txt 2>&1
tail -6 gpurun_out/re/cifar.txt
