#!/bin/bash
# one traced run of a bench workload (environment as given): tools/trace_wl.sh WORKLOAD  -> gpurun_out/trace_WORKLOAD/{timeline.txt,stats.csv}
W=$1; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/trace_$W; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --workload $W --no-cpu-baseline --no-e2e --steps 200 --warmup 60 > $O/run.log 2>&1 < /dev/null
python $R/tools/step_timeline.py $(ls $O/t/*/*kernel_trace.csv | head -1) > $O/timeline.txt 2>&1
cp $(ls $O/t/*/*kernel_stats.csv | head -1) $O/stats.csv
rm -rf $O/t
