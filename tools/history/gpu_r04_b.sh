#!/bin/bash
# round 4, visit b: whole GPU suite on the tree with both key-switch conventions / self-test / shared keys; the full default bench line (square pricing,
# LoLa + CIFAR children, cpu baseline) with its wall time; kernel trace of the batch
OUT=gpurun_out/r04b
mkdir -p $OUT
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
T1=$(date +%s); echo "pytest wall $((T1-T0)) s"
python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
T2=$(date +%s); echo "bench wall $((T2-T1)) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04b/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified_against_integer_model"], d["roofline"]["frac"])
print("square", json.dumps(d.get("square"))[:900])
print("lola", json.dumps(d.get("lola"))[:900])
print("cifar", json.dumps(d.get("cifar"))[:700])
u=d.get("unchanged_caller") or {}
print("unchanged", u.get("frac_of_batched"), u.get("frac_of_batched_mean_over_mean"), u.get("windows_ms"), u.get("error"))
print("cpu", json.dumps(d.get("cpu_baseline"))[:300])
PY
tail -3 $OUT/bench.err
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name "*kernel_trace.csv" -delete
head -30 $OUT/trace_summary.txt
