#!/bin/bash
# Round-2 visit D: matrix-core GEMM with the cooperative B operand, convolution tiled onto it (BENCH_CONV_TILE), fused squaring parked in LDS by default
OUT=gpurun_out/r02d
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -5 $OUT/pytest.txt
for tile in 1 2 4 7; do
  BENCH_CONV_TILE=$tile timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-unchanged-caller > $OUT/bench_tile$tile.json 2> $OUT/bench_tile$tile.err
  echo "conv tile $tile:"; cut -c1-200 $OUT/bench_tile$tile.json; tail -1 $OUT/bench_tile$tile.err | cut -c1-200
done
export TMPDIR=/tmp
R=$PWD
for tile in 1 4; do
(cd /tmp && BENCH_CONV_TILE=$tile CN_ROCTX=1 rocprofv3 --kernel-trace --marker-trace --stats -f csv -d $R/$OUT/prof$tile -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --serialize > $R/$OUT/prof_bench$tile.json 2> $R/$OUT/prof$tile.err)
KT=$(find $OUT/prof$tile -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $KT > $OUT/trace_summary$tile.txt 2>&1
find $OUT/prof$tile -name "*marker*stats*.csv" -exec cp {} $OUT/marker_stats$tile.csv \;
find $OUT/prof$tile -name "*_trace.csv" -delete
echo "== trace conv tile $tile"; head -14 $OUT/trace_summary$tile.txt; cat $OUT/marker_stats$tile.csv | cut -c1-110
done
