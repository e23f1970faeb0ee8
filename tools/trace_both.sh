#!/bin/bash
# traced replayed step of this tree and of _r04/ on one box -> gpurun_out/trace_both/{r05,r04}_timeline.txt
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/trace_both; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for t in r05 r04; do
  D=$([ $t = r04 ] && echo $R/_r04 || echo $R)
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $O/$t -- python3 $D/bench.py --steps 200 --warmup 60 --graphs --no-cpu-baseline --no-e2e > $O/$t.log 2>&1 < /dev/null
  python $R/tools/step_timeline.py $(ls $O/$t/*/*kernel_trace.csv | head -1) > $O/${t}_timeline.txt 2>&1
  rm -rf $O/$t
done
