#!/bin/bash
# round 4, visit h: the forward transform from the unit built under max-ilp (cn_l_nttf_f64l.hip): words, A/B on the NTT grid and on the bench line, HBM traffic passes
OUT=gpurun_out/r04h
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_gpu_client.py tests/test_lola.py -m gpu -x -q -k "ntt or encrypt or keygen or lola or linear or plain" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
for v in 0 1 0 1; do
  echo "CN_NTT_FWD_ILP=$v"
  CN_NTT_FWD_ILP=$v python tools/ntt_grid.py 2>/dev/null | grep -E "^C[23] +8192 +[25] +(1690|8192)" 
done | tee $OUT/ntt_grid_ab.txt
for v in 0 1; do
  CN_NTT_FWD_ILP=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late > $OUT/bench_ilp$v.json 2>> $OUT/bench.err
  python -c "import json; d=json.loads(open('$OUT/bench_ilp$v.json').read().strip().splitlines()[-1]); print('ntt_fwd_ilp=$v', d['ms_per_step'], d['verified_against_integer_model'], d['roofline']['frac'], d['roofline']['ms_per_launch'])"
done
bash tools/pmc_traffic.sh > $OUT/pmc_traffic.txt 2>&1; tail -12 $OUT/pmc_traffic.txt
cp gpurun_out/pmc_traffic/ntt_hbm_traffic.json $OUT/ 2>/dev/null; cp gpurun_out/pmc_traffic/fetch_size_counter_collection.csv gpurun_out/pmc_traffic/write_size_counter_collection.csv $OUT/ 2>/dev/null
python tools/ntt_grid.py > $OUT/ntt_grid.txt 2>/dev/null; head -14 $OUT/ntt_grid.txt
