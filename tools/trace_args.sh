#!/bin/bash
# one traced replayed train step with extra bench.py arguments: tools/trace_args.sh TAG <bench args>  -> gpurun_out/trace_TAG/timeline.txt
T=$1; shift; R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/trace_$T; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --no-cpu-baseline --no-e2e --steps 50 --warmup 60 --graphs "$@" > $O/run.log 2>&1 < /dev/null
python $R/tools/step_timeline.py $(ls $O/t/*/*kernel_trace.csv | head -1) > $O/timeline.txt
cp $(ls $O/t/*/*kernel_stats.csv | head -1) $O/stats.csv
rm -rf $O/t
