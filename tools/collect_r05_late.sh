#!/bin/bash
# Round-5 late collection (after the 32-seed fusions): tools/collect_r05_late.sh A | B
#   A: the driver's default command + rocprofv3 kernel stats of the same command + the Reddit step's timeline
#   B: the small rungs: bench lines (incl. the reference's own settings), traced steps, A/B of the three switches of this round's fusions
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/prof_late; mkdir -p $O
case "$1" in
A)
  timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err < /dev/null
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $O/rocprof_default.log 2>&1 < /dev/null)
  cp $(ls $O/t/*/*kernel_stats.csv | head -1) $O/rocprofv3_kernel_stats.csv 2> /dev/null
  python tools/step_timeline.py $(ls $O/t/*/*kernel_trace.csv | head -1) > $O/step_timeline.txt 2>&1
  rm -rf $O/t
  python tools/bench_brief.py $O/bench_default.json
  ;;
B)
  for w in pubmed_rbr arxiv_rbr pubmed_settings arxiv_settings bitcoin_settings; do
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err < /dev/null
  done
  for w in pubmed_rbr arxiv_rbr; do
    bash tools/trace_wl.sh $w > /dev/null 2>&1 || true
    cp gpurun_out/trace_$w/timeline.txt $O/step_timeline_$w.txt 2> /dev/null || true
    cp gpurun_out/trace_$w/stats.csv $O/kernel_stats_$w.csv 2> /dev/null || true
  done
  python tools/bench_brief.py $O/bench_*.json
  ;;
esac
