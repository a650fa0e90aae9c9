#!/bin/bash
O=gpurun_out/r05u; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_evaluator.py tests/test_gpu_multi_context.py tests/test_lola_cifar.py tests/test_lola.py -m gpu -q -x > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
