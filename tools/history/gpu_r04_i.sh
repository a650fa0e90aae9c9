#!/bin/bash
# round 4, visit i: fused encryption kernel - words against the three-launch chain, the unchanged caller with and without it, HBM traffic passes + NTT grid of the final tree
OUT=gpurun_out/r04i
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_client.py tests/test_deferred.py tests/test_cryptonets_mnist.py tests/test_gpu_serialization.py -m gpu -x -q > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for v in 0 1 0 1; do
  CN_ENC_FUSED=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-single-image --no-relinearize-late > $OUT/bench_enc$v.json 2>> $OUT/bench.err
  python -c "import json; d=json.loads(open('$OUT/bench_enc$v.json').read().strip().splitlines()[-1]); u=d['unchanged_caller']; print('enc_fused=$v', d['ms_per_step'], d['verified_against_integer_model'], 'unchanged', u['ms_per_step'], u['frac_of_batched'], u['frac_of_batched_mean_over_mean'], u['verified_against_integer_model'], u['windows_ms']['unchanged'])"
done
bash tools/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1; grep -E "traffic_over_algorithmic|correction" $OUT/pmc_traffic.txt
cp gpurun_out/pmc_traffic/ntt_hbm_traffic.json gpurun_out/pmc_traffic/fetch_size_counter_collection.csv gpurun_out/pmc_traffic/write_size_counter_collection.csv $OUT/ 2>/dev/null
python tools/ntt_grid.py > $OUT/ntt_grid.txt 2>/dev/null; head -13 $OUT/ntt_grid.txt
