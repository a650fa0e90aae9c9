#!/bin/bash
# A/B of kernel builds in one GPU visit: cryptonets_amd/lib/libcnhip_<tag>.so variants (built with -D switches, see tools/README.md)
# against the default library.  Parity first (the default build must pass the evaluator suite), then timings per build.
OUT=gpurun_out/ab
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py -m gpu -x -q > $OUT/pytest_new.txt 2>&1
tail -3 $OUT/pytest_new.txt
for tag in "" _base _v1 _v2 $AB_EXTRA; do
  lib=$PWD/cryptonets_amd/lib/libcnhip$tag.so
  [ -f $lib ] || continue
  echo "== build '$tag'"
  CNHIP_LIB=$lib NTT_PROBE_ONLY=1 timeout 300 python tools/ntt_probe.py > $OUT/ntt$tag.txt 2>&1
  cat $OUT/ntt$tag.txt | tail -2
  CNHIP_LIB=$lib timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench$tag.json 2> $OUT/bench$tag.err
  cut -c1-200 $OUT/bench$tag.json
done
for tag in "" _base; do
  lib=$PWD/cryptonets_amd/lib/libcnhip$tag.so
  CNHIP_LIB=$lib timeout 300 python tools/lola_latency.py > $OUT/lola$tag.txt 2>&1
  tail -2 $OUT/lola$tag.txt | cut -c1-200
done
