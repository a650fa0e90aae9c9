#!/bin/bash
# Round-3 visit B: (1) can HBM-bound kernels of one prime channel hide under the other's key switch (tools/overlap_probe.py)?  (2) bench with the
# channels staggered by half a batch vs lock step, 3 repeats each; (3) XCD-aware placement of the key switch's workgroups (CN_KS_XCD=1):
# parity test, HIP-event time, bench; (4) kernel trace of the staggered run
OUT=gpurun_out/r03b
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_evaluator.py -m gpu -q -k "xcd_placement or variants_agree" 2>&1 | tail -3
timeout 600 python tools/overlap_probe.py > $OUT/overlap_probe.txt 2>&1; cat $OUT/overlap_probe.txt | tail -8
B="--no-cpu-baseline --no-unchanged-caller --steps 10 --warmup 2"
for rep in 1 2 3; do
  for st in 0 1; do
    timeout 600 python bench.py $B --stagger $st > $OUT/bench_st${st}_$rep.json 2>> $OUT/bench.err
    python -c "import json; d=json.load(open('$OUT/bench_st${st}_$rep.json')); print('stagger $st rep $rep:', d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['key_switch']['ms_per_launch'], d['logit_words_sha256'][:12])"
  done
done
for st in 0 1; do
  CN_KS_XCD=1 timeout 600 python bench.py $B --stagger $st > $OUT/bench_xcd_st$st.json 2>> $OUT/bench.err
  python -c "import json; d=json.load(open('$OUT/bench_xcd_st$st.json')); print('ks_xcd stagger $st:', d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['key_switch']['ms_per_launch'])"
done
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --stagger 1 > /dev/null 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > $OUT/trace_stagger.txt 2>&1; cp $KT $OUT/kernel_trace_stagger.csv; find $OUT/prof -name "*kernel_trace.csv" -delete
head -16 $OUT/trace_stagger.txt | cut -c1-140
