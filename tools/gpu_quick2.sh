#!/bin/bash
rocm-smi -c -P 2>&1 | head -20
python tools/clock_probe.py 2>&1 | tail -8
