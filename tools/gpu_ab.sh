#!/bin/bash
# A/B of kernel builds in one GPU visit: cryptonets_amd/lib/libcnhip_<tag>.so variants (built with -D switches) against the default
# library.  Parity first (the default build must pass the evaluator + workload suites), then timings per build.
OUT=gpurun_out/ab
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py -m gpu -x -q > $OUT/pytest_new.txt 2>&1
tail -3 $OUT/pytest_new.txt
for tag in "" _base $AB_EXTRA; do
  lib=$PWD/cryptonets_amd/lib/libcnhip$tag.so
  [ -f $lib ] || continue
  echo "== build '$tag'"
  if [ -z "$AB_NO_NTT" ]; then CNHIP_LIB=$lib NTT_PROBE_ONLY=1 timeout 300 python tools/ntt_probe.py 2>&1 | tail -2; fi
  CNHIP_LIB=$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench$tag.json 2> $OUT/bench$tag.err
  cut -c1-200 $OUT/bench$tag.json
  if [ -n "$AB_TRACE" ]; then
    export TMPDIR=/tmp; R=$PWD
    (cd /tmp && CNHIP_LIB=$lib rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof$tag -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --serialize > /dev/null 2> $R/$OUT/prof$tag.err)
    KT=$(find $OUT/prof$tag -name "*kernel_trace.csv" | head -1)
    python tools/summarize_trace.py $KT > $OUT/trace$tag.txt 2>&1; find $OUT/prof$tag -name "*kernel_trace.csv" -delete
    grep -E "gemm|keyswitch_rr" $OUT/trace$tag.txt | cut -c1-120
  fi
done
if [ -n "$AB_LOLA" ]; then for tag in "" _base; do
  CNHIP_LIB=$PWD/cryptonets_amd/lib/libcnhip$tag.so timeout 300 python tools/lola_latency.py 2>&1 | tail -2 | cut -c1-200
done; fi
