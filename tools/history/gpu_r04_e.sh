#!/bin/bash
# round 4, visit e: whole GPU suite, the full default bench line (LoLa + CIFAR children), kernel trace of the batch, SQ counter passes of the shipped build
# (VERDICT r03 next #6), the two single-image workloads with ONE client's keys broadcast (world 1, process group forced)
OUT=gpurun_out/r04e
mkdir -p $OUT
T0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q > $OUT/pytest.txt 2>&1
grep -E "passed|failed" $OUT/pytest.txt | tail -3
T1=$(date +%s); echo "pytest wall $((T1-T0)) s"
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | cut -c1-160
python bench.py --steps 20 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
T2=$(date +%s); echo "bench wall $((T2-T1)) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04e/bench.json").read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d["verified_against_integer_model"], d["roofline"]["frac"], "ks", d["key_switch"]["ms_per_launch"], d["key_switch"]["frac_in_situ"])
print("square", {k: d["square"].get(k) for k in ("ms_per_chain","share_of_batch","frac_fp64_in_situ","frac_valu_in_situ","hbm_frac_designed","error")})
l=d.get("lola") or {}; print("lola", {k: l.get(k) for k in ("ms_per_image","verified","unchanged_caller_ms","launches_per_prime","unchanged_frac_of_batched","child_wall_s","skipped","error")})
c=d.get("cifar") or {}; print("cifar", {k: c.get(k) for k in ("s_per_image","verified","child_wall_s","skipped","error")})
u=d.get("unchanged_caller") or {}
print("unchanged", u.get("ms_per_step"), u.get("frac_of_batched"), u.get("frac_of_batched_mean_over_mean"), u.get("windows_ms"), (u.get("at_visible_cpu_count") or {}), u.get("error"))
print("late", (d.get("relinearize_late") or {}).get("ms_per_step"), "cpu", (d.get("cpu_baseline") or {}).get("value"), (d.get("cpu_baseline") or {}).get("cores"))
PY
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/$OUT/prof -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize > $R/$OUT/prof_bench.json 2> $R/$OUT/prof.err)
KT=$(find $OUT/prof -name "*kernel_trace.csv" | head -1)
python tools/summarize_trace.py $KT > $OUT/trace_summary.txt 2>&1
find $OUT/prof -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/prof -name "*kernel_trace.csv" -delete
head -16 $OUT/trace_summary.txt
# SQ counters of the shipped build: three separate --pmc passes (no other trace domain)
P=$OUT/pmc; mkdir -p $P
B="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-unchanged-caller --no-single-image --no-relinearize-late --serialize"
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -f csv -d $R/$P/p1 -- $B > $R/$P/run1.txt 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS -f csv -d $R/$P/p2 -- $B > $R/$P/run2.txt 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_ANY -f csv -d $R/$P/p3 -- $B > $R/$P/run3.txt 2>&1)
python - <<'PY'
import csv, glob, collections
for p in ("p1", "p2", "p3"):
    f = glob.glob("gpurun_out/r04e/pmc/%s/**/*counter_collection.csv" % p, recursive=True)
    if not f:
        print(p, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = (r["Kernel_Name"].split("(")[0].replace("void ", "")[:50], r["Grid_Size"])
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    names = sorted({c for v in acc.values() for c in v})
    with open("gpurun_out/r04e/pmc_%s_summary.txt" % p, "w") as o:
        o.write("%-52s %10s " % ("kernel", "grid") + " ".join("%16s" % n for n in names) + "\n")
        rows = sorted(acc.items(), key=lambda kv: -max(kv[1].values()))[:20]
        for k, v in rows:
            o.write("%-52s %10s " % k + " ".join("%16.4g" % (v[n] / max(1, cnt[(k, n)])) for n in names) + "\n")
    print(open("gpurun_out/r04e/pmc_%s_summary.txt" % p).read()[:2600])
PY
rm -rf $P
# configs 4 / 5 with one client's keys (RCCL broadcast at world 1)
for w in lola cifar; do
  BENCH_FORCE_DIST=1 MASTER_PORT=29561 python bench.py --workload $w --steps 5 --warmup 1 --shared-keys --no-unchanged-caller > $OUT/bench_${w}_shared.json 2>> $OUT/bench.err
  python -c "import json; d=json.loads(open('$OUT/bench_${w}_shared.json').read().strip().splitlines()[-1]); print('$w shared keys', d['ms_per_step'], d['verified_against_integer_model'], d['key_broadcast'])"
done
