#!/bin/bash
# A/B of one environment switch on a bench workload: tools/ab_wl.sh NAME WORKLOAD [rounds]
N=$1; W=$2; R=${3:-3}; O=gpurun_out/ab_${N}_$W; mkdir -p $O
for i in $(seq 1 $R); do
  for c in 0 1; do
    env $N=$c timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline > $O/${c}_$i.json 2> /dev/null < /dev/null
  done
done
python tools/bench_brief.py $O/*.json
