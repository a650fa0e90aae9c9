#!/bin/bash
# visit V: whole GPU suite + the three bench lines on the tree with pick_stream / direct staged operands / size-3 GEMMs
O=gpurun_out/r03v; mkdir -p $O
python -m pytest tests -q -x -m gpu 2>&1 | tail -4
python bench.py > $O/bench.json 2> $O/bench.err
python bench.py --workload lola > $O/bench_lola.json 2>> $O/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r03v/bench.json'))
print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['key_switch'].get('ms'), d['unchanged_caller']['frac_of_batched'], d['unchanged_caller']['skipped_taps']['frac_of_batched'], d['unchanged_caller'].get('at_visible_cpu_count'), d['relinearize_late']['ms_per_step'], d['cpu_baseline']['value'])
l=json.load(open('gpurun_out/r03v/bench_lola.json'))
u=l['unchanged_caller']
print(l['value'], l['ms_per_step'], l['verified_against_integer_model'], u['ms_per_image'], u['batched_from_the_same_host_ms'], u['frac_of_batched'], u['every_call_launched_on_its_own_ms'], u['launches_per_prime'])
PY
