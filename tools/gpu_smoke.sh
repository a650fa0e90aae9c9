#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py --gpus 1 --steps 3 --warmup 1 --no-unchanged-caller --no-cpu-baseline 2>/dev/null | cut -c1-200
BENCH_SELF_LAUNCH=1 python bench.py --gpus 1 --steps 3 --warmup 1 --no-unchanged-caller --no-cpu-baseline 2>/dev/null | cut -c1-200
