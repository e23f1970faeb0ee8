#!/bin/bash
# same-box A/B of several values of one environment variable: NAME "v1 v2 ..." runs  (long run + the driver's command)
N=$1; V=$2; R=${3:-3}
O=gpurun_out/abv_$N; mkdir -p $O
for i in $(seq 1 $R); do
  for v in $V; do
    env $N=$v timeout -k 10 200 python bench.py --steps 200 --warmup 60 --graphs --no-cpu-baseline --no-e2e > $O/long_${v}_$i.json 2> $O/long_${v}_$i.err < /dev/null
    env $N=$v timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-e2e > $O/drv_${v}_$i.json 2> /dev/null < /dev/null
  done
done
python - "$O" "$V" <<'PY'
import glob, json, sys
for kind in ("long", "drv"):
    for v in sys.argv[2].split():
        ms = [json.loads(open(f).read().strip().splitlines()[-1])["ms_per_step"] for f in sorted(glob.glob("%s/%s_%s_*.json" % (sys.argv[1], kind, v)))]
        print(kind, v, ms)
PY
