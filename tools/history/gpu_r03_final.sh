#!/bin/bash
# Round-3 closing visit: GPU suite + the three bench lines + the LoLa unchanged-caller table on the final tree (outputs copied over profiles/r03_bench*.json, r03_lola_unchanged_caller.txt)
O=gpurun_out/r03final; mkdir -p $O
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest.txt 2>&1; grep -E "passed|failed|FAILED" $O/pytest.txt | tail -4
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-120
python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err
python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['key_switch']['ms_per_launch'], d['unchanged_caller']['frac_of_batched'], d['unchanged_caller']['at_visible_cpu_count'], d['unchanged_caller']['skipped_taps']['frac_of_batched'], d['relinearize_late']['ms_per_step'], d['cpu_baseline']['value'])"
python bench.py --workload lola --steps 20 --warmup 2 > $O/bench_lola.json 2>> $O/bench.err
python -c "import json; d=json.load(open('$O/bench_lola.json')); u=d['unchanged_caller']; print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], {k:u.get(k) for k in u if k not in ('all_rows','pattern')})"
python tools/lola_unchanged_caller.py LoLa --reps 20 > $O/lola_unchanged_caller.txt 2>/dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r03final/lola_unchanged_caller.txt"):
    r=json.loads(l); print("  %-60s %-62s %6.2f ms %s %s" % (r["pattern"][:60], r["host"][:62], r["ms_per_image"], r.get("launches_per_prime",""), r["logits_exact"]))
PY
