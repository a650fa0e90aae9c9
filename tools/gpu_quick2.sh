#!/bin/bash
SECONDS=0; python bench.py --steps 20 --warmup 3 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench.py --steps 20 --warmup 3: $SECONDS s"
SECONDS=0; python bench.py --workload lola --steps 20 --warmup 2 > gpurun_out/bench_lola_default.json 2> gpurun_out/bench_lola_default.err; echo "bench.py --workload lola --steps 20: $SECONDS s"
