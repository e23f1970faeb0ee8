#!/bin/bash
# A/B of several values of one environment variable inside the replayed train step: tools/ab_vals.sh NAME "v1 v2 ..." [rounds]
N=$1; V=$2; R=${3:-3}; O=gpurun_out/ab_$N; mkdir -p $O
for i in $(seq 1 $R); do
  for c in $V; do
    env $N=$c timeout -k 10 200 python bench.py --steps 200 --warmup 60 --graphs --no-cpu-baseline --no-e2e > $O/${c}_$i.json 2> /dev/null < /dev/null
  done
done
python tools/bench_brief.py $O/*.json
