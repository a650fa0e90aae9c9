#!/bin/bash
# round 4, visit o: does chunking Multiply + Relinearize (tensor kept in the 256 MB infinity cache between the squaring and the floor kernels) pay?  CN_SCRATCH_GB sweeps the chunk size
OUT=gpurun_out/r04o
mkdir -p $OUT
for gb in 24 5.6 2.8 1.4; do
  CN_SCRATCH_GB=$gb python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late 2>> $OUT/bench.err | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('CN_SCRATCH_GB=$gb', d['ms_per_step'], d['verified_against_integer_model'], 'square chain', d['square']['ms_per_chain'], 'ks', d['key_switch']['ms_per_launch'])"
done | tee $OUT/chunk_sweep.txt
