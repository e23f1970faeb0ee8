#!/bin/bash
# A/B of one environment switch with extra bench.py arguments: tools/ab_env_args.sh NAME ROUNDS TAG -- <bench args>
N=$1; R=$2; T=$3; shift 4; O=gpurun_out/ab_${N}_$T; mkdir -p $O
for i in $(seq 1 $R); do
  for c in 0 1; do
    env $N=$c timeout -k 10 200 python bench.py --steps 200 --warmup 60 --graphs --no-cpu-baseline --no-e2e "$@" > $O/${c}_$i.json 2> /dev/null < /dev/null
  done
done
python tools/bench_brief.py $O/*.json
