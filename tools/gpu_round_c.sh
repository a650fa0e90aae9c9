#!/bin/bash
mkdir -p gpurun_out/rc
timeout 600 python -c "
import cProfile, pstats, sys, runpy
sys.argv=['tools/lola_latency.py']
pr=cProfile.Profile(); pr.enable()
runpy.run_path('tools/lola_latency.py', run_name='__main__')
pr.disable()
import io
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('cumulative').print_stats(70); open('gpurun_out/rc/lola_cprofile.txt','w').write(s.getvalue())
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('tottime').print_stats(40); open('gpurun_out/rc/lola_cprofile_tot.txt','w').write(s.getvalue())
" > gpurun_out/rc/lola.txt 2>&1
tail -5 gpurun_out/rc/lola.txt
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats -f csv -d $R/gpurun_out/rc/prof -- python $R/tools/lola_latency.py > $R/gpurun_out/rc/lola_prof.txt 2>&1)
KT=$(find gpurun_out/rc/prof -name "*kernel_trace.csv" | head -1); python tools/summarize_trace.py $KT > gpurun_out/rc/lola_trace_summary.txt 2>&1
find gpurun_out/rc/prof -name "*kernel_trace.csv" -delete
head -40 gpurun_out/rc/lola_trace_summary.txt
