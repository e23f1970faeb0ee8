#!/bin/bash
# A/B of several ENVIRONMENT COMBINATIONS inside the replayed train step: tools/ab_combo.sh TAG rounds "A=1 B=0" "A=0" ...   (alternating runs of 200 steps)
T=$1; R=$2; shift 2; O=gpurun_out/ab_$T; mkdir -p $O
for i in $(seq 1 $R); do
  k=0
  for c in "$@"; do
    env $c timeout -k 10 200 python bench.py --steps 200 --warmup 60 --graphs --no-cpu-baseline --no-e2e > $O/c${k}_$i.json 2> /dev/null < /dev/null
    k=$((k+1))
  done
done
k=0; for c in "$@"; do echo "c$k = $c"; k=$((k+1)); done
python tools/bench_brief.py $O/*.json
