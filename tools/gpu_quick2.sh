#!/bin/bash
python bench.py --workload cifar --steps 2 --warmup 2 > gpurun_out/bench_cifar_q.json 2>/dev/null; cut -c1-330 gpurun_out/bench_cifar_q.json
