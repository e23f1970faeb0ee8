#!/bin/bash
# A/B of several values of one environment variable with extra bench.py arguments: tools/ab_vals_args.sh NAME "v1 v2 ..." ROUNDS TAG -- <bench args>
N=$1; V=$2; R=$3; T=$4; shift 5; O=gpurun_out/ab_${N}_$T; mkdir -p $O
for i in $(seq 1 $R); do
  for c in $V; do
    env $N=$c timeout -k 10 200 python bench.py --steps 200 --warmup 60 --graphs --no-cpu-baseline --no-e2e "$@" > $O/${c}_$i.json 2> /dev/null < /dev/null
  done
done
python tools/bench_brief.py $O/*.json
