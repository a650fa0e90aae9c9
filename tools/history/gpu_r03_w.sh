#!/bin/bash
# visit W: CryptoNets batch with and without the stream probe, alternating, same box
O=gpurun_out/r03w; mkdir -p $O
for rep in 1 2 3; do
for p in 0 1; do
  CN_STREAM_PROBE=$p python bench.py --steps 20 --warmup 3 --no-unchanged-caller > $O/b_p${p}_$rep.json 2>/dev/null
  python - $p $rep <<'PY'
import json,sys
d=json.load(open('gpurun_out/r03w/b_p%s_%s.json' % (sys.argv[1], sys.argv[2])))
print("probe", sys.argv[1], "rep", sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['frac'], d['relinearize_late']['ms_per_step'], {k: v for k, v in d['key_switch'].items() if 'ms' in k and not isinstance(v, dict)})
PY
done; done
