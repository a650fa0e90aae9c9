#!/bin/bash
# visit AB: slice-major workgroup order of the scalar GEMM (conv): words, kernel time, batch time, fetch counters
O=gpurun_out/r03ab; mkdir -p $O
python -m pytest tests/test_gpu_evaluator.py tests/test_cryptonets_mnist.py tests/test_deferred.py -q -x -m gpu -k "gemm or cryptonets or unchanged or deferred" 2>&1 | grep -E "passed|failed|FAILED" | tail -3
for rep in 1 2; do for o in 0 1; do
  CN_GEMM_ORDER=$o python bench.py --steps 20 --warmup 3 --no-unchanged-caller --no-cpu-baseline > $O/b_o${o}_$rep.json 2>/dev/null
  python -c "
import json; d=json.load(open('$O/b_o${o}_$rep.json')); print('order $o rep $rep', d['value'], d['ms_per_step'], d['verified_against_integer_model'], d['relinearize_late']['ms_per_step'])"
done; done
export TMPDIR=/tmp; R=$PWD
for o in 0 1; do
(cd /tmp && CN_GEMM_ORDER=$o rocprofv3 --kernel-trace --stats -f csv -d $R/$O/prof$o -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-relinearize-late --serialize --stagger 0 > /dev/null 2>&1)
KT=$(find $O/prof$o -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT 2>/dev/null | grep -i "gemm" | cut -c1-120
find $O/prof$o -name "*kernel_trace.csv" -delete
(cd /tmp && CN_GEMM_ORDER=$o rocprofv3 --pmc FETCH_SIZE -f csv -d $R/$O/pmc$o -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-unchanged-caller --no-relinearize-late --serialize --stagger 0 > /dev/null 2>&1)
CC=$(find $O/pmc$o -name "*counter_collection.csv" | head -1)
python - $CC <<'PY'
import csv, sys, collections
rows=list(csv.DictReader(open(sys.argv[1])))
acc=collections.defaultdict(lambda:[0,0.0])
for r in rows:
    if 'gemm' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE':
        key=(r['Kernel_Name'].split('(')[0][:60], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X',''))
        acc[key][0]+=1; acc[key][1]+=float(r['Counter_Value'])
for k,v in acc.items(): print("FETCH_SIZE x2", k, "launches", v[0], "per launch %.1f MiB" % (2*v[1]/v[0]/1024))
PY
done
