#!/bin/bash
# same-box comparison of this tree with the round-4 tree kept under _r04/ (a git worktree at the round-4 commit; not tracked):
# tools/ab_r04.sh [rounds] [extra bench args]  — alternating runs of the driver's command and of the 200-step replayed run
R=${1:-3}; shift; O=gpurun_out/ab_r04; mkdir -p $O
for i in $(seq 1 $R); do
  for t in r04 r05; do
    D=$([ $t = r04 ] && echo _r04 || echo .)
    (cd $D && timeout -k 10 200 python bench.py --steps 200 --warmup 60 --graphs --no-cpu-baseline --no-e2e "$@" 2> /dev/null < /dev/null) > $O/${t}_long_$i.json
    (cd $D && timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e "$@" 2> /dev/null < /dev/null) > $O/${t}_drv_$i.json
  done
done
python tools/bench_brief.py $O/*.json
