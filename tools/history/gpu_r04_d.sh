#!/bin/bash
# round 4, visit d: k_square_pipe (pipelined resident squaring) - words, then A/B against k_square_fused on the batch; the merged zero-vector / free calls
OUT=gpurun_out/r04d
mkdir -p $OUT
timeout 1200 python -m pytest "tests/test_gpu_evaluator.py" tests/test_deferred.py tests/test_gpu_client.py tests/test_cryptonets_mnist.py -m gpu -x -q -k "squar or multiply or pipelined or deferred or literal or encrypt_zero or end_to_end or random_programs" > $OUT/pytest.txt 2>&1
tail -4 $OUT/pytest.txt
for v in 0 1 0 1; do
  CN_SQ_PIPE=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late > $OUT/bench_pipe$v.json 2>> $OUT/bench.err
  python - <<PY
import json
d=json.loads(open("$OUT/bench_pipe$v.json").read().strip().splitlines()[-1])
s=d["square"]
print("sq_pipe=$v", d["ms_per_step"], d["verified_against_integer_model"], "square chain ms", s.get("ms_per_chain"), "frac_fp64", s.get("frac_fp64_in_situ"), "ks", d["key_switch"]["ms_per_launch"], s.get("error"))
PY
done
