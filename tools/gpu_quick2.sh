#!/bin/bash
python -m pytest tests/test_deferred.py -q -x -m gpu -k "random_programs" 2>&1 | grep -E "passed|failed|FAILED|Error|assert|seed" | tail -8
