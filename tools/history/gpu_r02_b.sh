#!/bin/bash
# Round-2 visit B: queue lock / handle table / BEHZ tests, caller-thread sweep, LoLa / CIFAR bench workloads (image and prime sharding,
# forced-distributed at world 1), NTT grid, fused squaring with LDS parking (A/B), kernel + marker trace with roctx ranges.
OUT=gpurun_out/r02b
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-200 $OUT/bench.json; tail -2 $OUT/bench.err
CN_SQ_LDS=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-unchanged-caller > $OUT/bench_sqlds.json 2> $OUT/bench_sqlds.err
echo "sq_lds=1:"; cut -c1-200 $OUT/bench_sqlds.json
timeout 600 python tools/replay_reference_calls.py --threads 1,8,32,64,128,256 --steps 5 --trained > $OUT/replay.txt 2>&1
tail -8 $OUT/replay.txt | cut -c1-220
timeout 600 python bench.py --workload lola --steps 20 --warmup 3 > $OUT/bench_lola.json 2> $OUT/bench_lola.err
cut -c1-330 $OUT/bench_lola.json; tail -2 $OUT/bench_lola.err
BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload lola --shard primes --steps 10 --warmup 2 > $OUT/bench_lola_primes.json 2> $OUT/bench_lola_primes.err
cut -c1-330 $OUT/bench_lola_primes.json; tail -2 $OUT/bench_lola_primes.err
timeout 900 python bench.py --workload cifar --steps 2 --warmup 1 > $OUT/bench_cifar.json 2> $OUT/bench_cifar.err
cut -c1-330 $OUT/bench_cifar.json; tail -2 $OUT/bench_cifar.err
timeout 600 python tools/ntt_grid.py > $OUT/ntt_grid.txt 2>&1
head -14 $OUT/ntt_grid.txt
export TMPDIR=/tmp
R=$PWD
(cd /tmp && CN_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --serialize > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name "*marker*stats*.csv" -exec cp {} $OUT/marker_stats.csv \;
MT=$(find $OUT/prof -name "*marker_api_trace.csv" | head -1)
[ -n "$MT" ] && python - "$MT" "$KT" > $OUT/layer_attribution.txt 2>&1 <<'PY'
import csv, sys, collections
marks = [r for r in csv.DictReader(open(sys.argv[1]))]
kern = [r for r in csv.DictReader(open(sys.argv[2]))]
print("marker columns:", list(marks[0].keys()) if marks else None)
print("kernel columns:", list(kern[0].keys()) if kern else None)
print("ranges:", len(marks), "kernels:", len(kern))
PY
find $OUT/prof -name "*kernel_trace.csv" -delete; find $OUT/prof -name "*marker_api_trace.csv" -size +2M -delete
head -16 $OUT/trace_summary.txt; head -5 $OUT/layer_attribution.txt
