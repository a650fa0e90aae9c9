#!/bin/bash
python -m pytest tests/test_gpu_evaluator.py tests/test_gpu_multi_context.py -q -x -m gpu -k "permutation_pass or hardware_queue or copy_many" 2>&1 | grep -E "passed|failed|FAILED|Error" | tail -5
