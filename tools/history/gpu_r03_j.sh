#!/bin/bash
# Round-3 visit J (the record visit): whole GPU suite; default bench line; kernel trace of the same command; NTT HBM traffic (PMC passes);
# unchanged-caller tables (CryptoNets: thread sweep, padded taps skipped / literal; LoLa: recorded call trace replayed from C++); LoLa and
# CIFAR bench lines; NTT grid.  Outputs under gpurun_out/r03j - the ones to be judged are copied into profiles/r03_*.
OUT=gpurun_out/r03j
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1; grep -n "passed\|failed\|rror" $OUT/pytest.txt | head -10
timeout 900 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
python -c "import json; d=json.load(open('$OUT/bench.json')); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['key_switch']['ms_per_launch'], d['unchanged_caller']['frac_of_batched'], d['unchanged_caller']['skipped_taps']['frac_of_batched'], d['cpu_baseline']['value'])" || tail -20 $OUT/bench.err
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --serialize --stagger 0 > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \; ; find $OUT/prof -name "*kernel_trace.csv" -delete
head -14 $OUT/trace_summary.txt | cut -c1-130
timeout 600 bash tools/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1; tail -12 $OUT/pmc_traffic.txt | cut -c1-200
cp gpurun_out/pmc_traffic/ntt_hbm_traffic.json $OUT/ 2>/dev/null; cp gpurun_out/pmc_traffic/*_counter_collection.csv $OUT/ 2>/dev/null
timeout 900 python tools/replay_reference_calls.py --trained --threads 1,4,16,64,256 --literal-threads 1,4,16,64,256 --immediate --steps 5 > $OUT/unchanged_caller_replay.txt 2>&1
python - <<PY
import json
for ln in open("$OUT/unchanged_caller_replay.txt"):
    try: d = json.loads(ln)
    except Exception: print(ln.strip()[:200]); continue
    print("%-40s thr %3d  %6.2f ms  %.3f  launches %s %s" % (d["caller"][:40], d["threads"], d["ms_per_batch"], d.get("frac_of_batched", 1.0), d.get("launches_per_batch"), d["words_identical"]))
PY
timeout 900 python tools/lola_unchanged_caller.py LoLa --reps 20 > $OUT/lola_unchanged_caller.txt 2>&1; cut -c1-300 $OUT/lola_unchanged_caller.txt | tail -12
timeout 900 python bench.py --workload lola --steps 20 --warmup 2 > $OUT/bench_lola.json 2> $OUT/bench_lola.err; python -c "import json; d=json.load(open('$OUT/bench_lola.json')); u=d.get('unchanged_caller',{}); print(d['value'], d['ms_per_step'], d['verified_against_integer_model'], {k:u.get(k) for k in u if k!='all_rows'})"
timeout 1200 python bench.py --workload cifar --steps 2 --warmup 1 > $OUT/bench_cifar.json 2> $OUT/bench_cifar.err; cut -c1-330 $OUT/bench_cifar.json
timeout 600 python tools/ntt_grid.py > $OUT/ntt_grid.txt 2>&1; tail -14 $OUT/ntt_grid.txt | cut -c1-160
