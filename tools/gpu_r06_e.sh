#!/bin/bash
# Round 6, visit E: (1) squaring chain with the q-side transform kernel on a second stream (CN_SQ_OVERLAP=1) against the serial chain: parity, then the bench line
# alternating; kernel trace of one overlapped run.  (2) the literal unchanged caller with the three-terms-in-flight fold kernel
O=gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
R=$PWD
CN_SQ_OVERLAP=1 timeout 900 python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_deferred.py -m gpu -x -q -k "squar or multiply or mul_relin or cryptonets or deferred" > $O/pytest_overlap.txt 2>&1; tail -3 $O/pytest_overlap.txt
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late"
for rep in 1 2 3; do for ov in 0 1; do
  CN_SQ_OVERLAP=$ov $B > $O/bench_ov${ov}_$rep.json 2> $O/bench_ov${ov}_$rep.err
  python -c "
import json; d=json.loads(open('$O/bench_ov${ov}_$rep.json').read().strip().splitlines()[-1])
print('overlap $ov rep $rep:', d['value'], d['ms_per_step'], 'chain', d['square']['ms_per_chain'], 'ks', d['key_switch']['ms_per_launch'], d['verified_against_integer_model'])"
done; done
(cd /tmp && CN_SQ_OVERLAP=1 timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof -- $B --steps 3 --warmup 1 --serialize > /dev/null 2> $R/$O/prof.err)
KT=$(find $O/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $O/trace_overlap.txt 2>&1; find $O/prof -name "*kernel_trace.csv" -delete
head -12 $O/trace_overlap.txt | cut -c1-130
python tools/replay_reference_calls.py --trained --threads 16 --literal-threads 16,256 --steps 5 > $O/replay.txt 2> $O/replay.err
python -c "
import json
for ln in open('$O/replay.txt'):
    d = json.loads(ln); print(d['caller'][:48], d['threads'], d['ms_per_batch'], d.get('frac_of_batched'), d.get('words_identical'))"
