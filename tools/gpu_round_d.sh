#!/bin/bash
mkdir -p gpurun_out/rd
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/rd/pytest.txt 2>&1
tail -8 gpurun_out/rd/pytest.txt
timeout 600 python tools/lola_latency.py > gpurun_out/rd/lola.txt 2>&1
tail -6 gpurun_out/rd/lola.txt
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/rd/bench.json 2> gpurun_out/rd/bench.err
cut -c1-220 gpurun_out/rd/bench.json
