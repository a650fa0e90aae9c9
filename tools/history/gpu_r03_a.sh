#!/bin/bash
# Round-3 visit A: new tests first (broadcast keys device-copy + forced RCCL, forced process group, self-launcher, encode/decode batch, reused-handle
# bias fold, queue drain), then the whole GPU suite, the default bench line (all-slot verification, CPU baseline) and smoke()
OUT=gpurun_out/r03a
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_multi_context.py tests/test_deferred.py tests/test_gpu_client.py -m gpu -x -q > $OUT/pytest_new.txt 2>&1
tail -15 $OUT/pytest_new.txt
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['verified_slots'], d['roofline']['frac'], d['key_switch']['ms_per_launch'], d['unchanged_caller'], d['cpu_baseline'])" || tail -20 $OUT/bench.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
