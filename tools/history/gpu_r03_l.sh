#!/bin/bash
# Round-3 visit L: pinned upload ring (small tables), rotations no longer heavy for the flush trigger: tests, LoLa unchanged caller, CryptoNets sweep, bench lines
OUT=gpurun_out/r03l
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; grep -n "passed\|failed\|rror" $OUT/pytest.txt | head -10
timeout 900 python tools/lola_unchanged_caller.py LoLa --reps 20 > $OUT/lola_unchanged_caller.txt 2>&1
python - <<PY
import json
for ln in open("$OUT/lola_unchanged_caller.txt"):
    try: d = json.loads(ln)
    except Exception: print(ln.strip()[:200]); continue
    print(d["pattern"][:60].ljust(60), "|", d["host"][:52].ljust(52), d["ms_per_image"], d.get("calls_per_prime"), d.get("launches_per_prime"), d["logits_exact"])
PY
timeout 900 python tools/replay_reference_calls.py --trained --threads 1,4,16,64,256 --literal-threads 1,4,16,64,256 --steps 5 > $OUT/unchanged_caller_replay.txt 2>&1
python - <<PY
import json
for ln in open("$OUT/unchanged_caller_replay.txt"):
    try: d = json.loads(ln)
    except Exception: print(ln.strip()[:200]); continue
    print("%-40s thr %3d  %6.2f ms  %.3f  launches %s %s" % (d["caller"][:40], d["threads"], d["ms_per_batch"], d.get("frac_of_batched", 1.0), d.get("launches_per_batch"), d["words_identical"]))
PY
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); u=d['unchanged_caller']; print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['roofline']['traffic_source'][:30], u['frac_of_batched'], u['threads'], u['skipped_taps']['frac_of_batched'], d['cpu_baseline']['value'])" || tail -20 $OUT/bench.err
timeout 900 python bench.py --workload lola --steps 20 --warmup 2 > $OUT/bench_lola.json 2> $OUT/bench_lola.err; python -c "import json; d=json.load(open('$OUT/bench_lola.json')); u=d.get('unchanged_caller',{}); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], {k:u.get(k) for k in u if k not in ('all_rows','pattern')})"
