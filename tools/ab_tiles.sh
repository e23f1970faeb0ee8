#!/bin/bash
# A/B of the 160-row (OGL_X3_CFG3) and 160-column (OGL_X3_CFG4) image-GEMM tiles inside the replayed train step, alternating runs.
O=gpurun_out/ab_tiles; mkdir -p $O
for i in 1 2 3; do
  for c in 00 10 01 11; do
    OGL_X3_CFG3=${c:0:1} OGL_X3_CFG4=${c:1:1} timeout -k 10 200 python bench.py --steps 200 --warmup 60 --graphs --no-cpu-baseline --no-e2e > $O/ab_${c}_$i.json 2> /dev/null < /dev/null
  done
done
python tools/bench_brief.py $O/ab_*.json
