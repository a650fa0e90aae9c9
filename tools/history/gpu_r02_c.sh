#!/bin/bash
# Round-2 visit C: matrix-core GEMM (parity + A/B), spin lock sweep, LoLa bench with encryption outside the window, per-layer device time
# (roctx ranges that synchronise), HBM traffic of the fused squaring (both parking variants) and of the GEMM kernels.
OUT=gpurun_out/r02c
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
for mf in 1 0; do
  CN_GEMM_MFMA=$mf timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-unchanged-caller > $OUT/bench_mfma$mf.json 2> $OUT/bench_mfma$mf.err
  echo "gemm_mfma=$mf:"; cut -c1-200 $OUT/bench_mfma$mf.json; tail -1 $OUT/bench_mfma$mf.err
done
timeout 600 python bench.py --steps 10 --warmup 2 > $OUT/bench.json 2> $OUT/bench.err
cut -c1-200 $OUT/bench.json
timeout 600 python tools/replay_reference_calls.py --threads 1,8,16,32,64,256 --steps 5 --trained > $OUT/replay.txt 2>&1
tail -7 $OUT/replay.txt | cut -c1-220
timeout 600 python bench.py --workload lola --steps 20 --warmup 3 > $OUT/bench_lola.json 2> $OUT/bench_lola.err
cut -c1-250 $OUT/bench_lola.json
BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload lola --shard primes --steps 10 --warmup 2 > $OUT/bench_lola_primes.json 2> $OUT/bench_lola_primes.err
cut -c1-250 $OUT/bench_lola_primes.json
export TMPDIR=/tmp
R=$PWD
for mf in 1 0; do
(cd /tmp && CN_GEMM_MFMA=$mf CN_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats -f csv -d $R/$OUT/prof$mf -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --serialize > $R/$OUT/prof_bench$mf.json 2> $R/$OUT/prof$mf.err)
KT=$(find $OUT/prof$mf -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $KT > $OUT/trace_summary$mf.txt 2>&1
find $OUT/prof$mf -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats$mf.csv \;
find $OUT/prof$mf -name "*marker*stats*.csv" -exec cp {} $OUT/marker_stats$mf.csv \;
find $OUT/prof$mf -name "*_trace.csv" -delete
echo "== trace gemm_mfma=$mf"; head -12 $OUT/trace_summary$mf.txt; cat $OUT/marker_stats$mf.csv | cut -c1-120
done
# HBM traffic (separate passes per counter): fused squaring with the operand parked in the outputs' place / in LDS; GEMM kernels
for sq in 0 1; do for c in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && CN_SQ_LDS=$sq timeout 120 rocprofv3 --kernel-trace --pmc $c -f csv -d $R/$OUT/pmc_sq${sq}_$c -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-unchanged-caller --serialize > /dev/null 2> $R/$OUT/pmc_sq${sq}_$c.err)
done; done
python - <<'PY' > gpurun_out/r02c/pmc_summary.txt 2>&1
import csv, glob, collections
for sq in (0, 1):
    res = collections.defaultdict(dict)
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob("gpurun_out/r02c/pmc_sq%d_%s/**/*counter_collection.csv" % (sq, c), recursive=True):
            rows = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c and any(k in r["Kernel_Name"] for k in ("k_square_fused", "k_behz", "k_addsub", "k_scalar_gemm", "k_keyswitch_rr")):
                    rows[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])].append(float(r["Counter_Value"]))
            for k, v in rows.items():
                res[k][c] = sum(v) / len(v)
    print("== sq_lds = %d  (KiB per launch; FETCH_SIZE under-reports by 2x on gfx950, see profiles/r01_ntt_hbm_traffic.json)" % sq)
    for k, v in sorted(res.items()):
        print("  ", k, {c: round(x) for c, x in v.items()})
PY
cat $OUT/pmc_summary.txt | cut -c1-200
find $OUT -name "*_trace.csv" -delete
