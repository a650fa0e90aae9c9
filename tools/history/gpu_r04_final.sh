#!/bin/bash
# Round-4 closing visit: GPU suite + smoke + the default bench line + the stand-alone LoLa / CIFAR lines on the final tree (copied over profiles/r04_bench*.json)
O=gpurun_out/r04final; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; grep -E "passed|failed|FAILED" $O/pytest.txt | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
python bench.py > $O/bench_default_flags.json 2> $O/bench.err
python -c "import json; d=json.loads(open('$O/bench_default_flags.json').read().strip().splitlines()[-1]); print('default flags:', d['value'], d['steps'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], (d.get('lola') or {}).get('ms_per_image'), (d.get('cifar') or {}).get('s_per_image'))"
python bench.py --steps 20 --warmup 3 > $O/bench.json 2>> $O/bench.err
python -c "import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['key_switch']['ms_per_launch'], d['square']['ms_per_chain'], d['unchanged_caller']['frac_of_batched'], d['lola']['ms_per_image'], d['lola']['unchanged_caller_ms'], d['cifar']['s_per_image'], d['relinearize_late']['ms_per_step'], d['cpu_baseline']['value'])"
python bench.py --workload lola --steps 20 --warmup 3 > $O/bench_lola.json 2>> $O/bench.err
python bench.py --workload cifar --steps 1 --warmup 1 > $O/bench_cifar.json 2>> $O/bench.err
python -c "
import json
for w in ('lola','cifar'):
    d=json.loads(open('$O/bench_%s.json' % w).read().strip().splitlines()[-1]); print(w, d['ms_per_step'], d['verified_against_integer_model'])"
