#!/bin/bash
python -m pytest tests/test_basic_operations.py tests/test_lola.py tests/test_layers.py tests/test_call_trace.py tests/test_deferred.py -q -x -m gpu 2>&1 | grep -E "passed|failed|FAILED" | tail -3
python bench.py --workload lola --steps 20 --warmup 2 > gpurun_out/bench_lola_q.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/bench_lola_q.json')); u=d['unchanged_caller']; print('bench lola', d['value'], d['ms_per_step'], d['verified_against_integer_model'], {k:u.get(k) for k in u if k not in ('all_rows','pattern')})
for r in u['all_rows']: print('  %-60s %-62s %6.2f ms %s' % (r['pattern'][:60], r['host'][:62], r['ms_per_image'], r.get('launches_per_prime','')))"
