#!/bin/bash
# visit AA: automorphism inside the two-launch key switch
O=gpurun_out/r03aa; mkdir -p $O
python -m pytest tests -q -x -m gpu > $O/pytest.txt 2>&1; grep -E "passed|failed|FAILED" $O/pytest.txt | tail -5
python bench.py --workload lola > $O/bench_lola.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03aa/bench_lola.json')); u=d['unchanged_caller']; print('bench lola', d['value'], d['ms_per_step'], d['verified_against_integer_model'], u['ms_per_image'], u['batched_from_the_same_host_ms'], u['frac_of_batched'], u['every_call_launched_on_its_own_ms'], u['launches_per_prime'])"
python tools/lola_latency.py LoLa --graph 2>/dev/null | tail -3
python bench.py --workload cifar --no-unchanged-caller > $O/bench_cifar.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r03aa/bench_cifar.json')); print('bench cifar', d['value'], d['ms_per_step'], d['verified_against_integer_model'])"
