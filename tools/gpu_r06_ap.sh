#!/bin/bash
# Round 6, visit AP: Multiply + Relinearize of a batch as two halves pipelined over two streams of the context (cn_set_option "sq_halves", CN_SQ_HALVES): words, then A/B on the bench line
R=$(pwd); O=$R/gpurun_out/r06ap; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_evaluator.py -q -k "two_pipelined_halves or pipelined_squaring" 2>&1 | tail -5 | tee $O/test.txt
for rep in 1 2; do for hv in 0 1; do
  CN_SQ_HALVES=$hv python bench.py --no-cpu-baseline --no-single-image --no-relinearize-late 2>$O/err_$hv.txt | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); u = d['unchanged_caller']
print('halves $hv rep $rep: batched', d['ms_per_step'], d['verified_against_integer_model'], 'unstaggered', u.get('unstaggered_batched_ms'), 'literal', u['ms_per_step'], u['frac_of_batched'], u['verified_against_integer_model'], 'skipped', u['skipped_taps']['ms_per_step'], u['skipped_taps']['words_identical_to_batched'], 'square chain', d['square']['ms_per_chain'], 'ks', d['key_switch']['ms_per_launch'])" | tee -a $O/ab.txt
done; done
