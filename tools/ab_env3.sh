#!/bin/bash
# same-box A/B of one environment switch inside the replayed step AND at the driver's command: NAME, runs per arm
# usage (through gpurun, from the repo root): bash tools/ab_env3.sh OGL_SAMPLE_PIPELINE 3
N=$1; R=${2:-3}
O=gpurun_out/ab_$N; mkdir -p $O
for i in $(seq 1 $R); do
  for v in 0 1; do
    env $N=$v timeout -k 10 200 python bench.py --steps 200 --warmup 60 --graphs --no-cpu-baseline --no-e2e > $O/long_${v}_$i.json 2> /dev/null < /dev/null
    env $N=$v timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $O/drv_${v}_$i.json 2> /dev/null < /dev/null
  done
done
python - "$O" <<'PY'
import glob, json, sys
for kind in ("long", "drv"):
    for v in "01":
        ms = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob("%s/%s_%s_*.json" % (sys.argv[1], kind, v)))]
        print(kind, v, ms)
PY
