#!/bin/bash
# The exact kernel sequence of ONE train step (rocprofv3 kernel trace of a short bench run, the launches between two Adam steps),
# with each kernel's duration: gpurun_out/seq/one_step.txt.  Run from the repo root through gpurun.
# WORKLOAD=reddit_pbr_forward MARK=k_ce_fwd_bwd lists one priority-forward batch instead.
R=$PWD; O=$R/gpurun_out/seq; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/t -- python3 $R/bench.py --workload ${WORKLOAD:-reddit_rbr} --steps 12 --warmup 2 --no-cpu-baseline > $O/log.txt 2>&1
cd $R; python - <<'PY'
import csv, glob, os
f = glob.glob('gpurun_out/seq/t/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if os.environ.get('MARK', 'k_adam') in r['Kernel_Name']]
a, b = idx[5], idx[6]
with open('gpurun_out/seq/one_step.txt', 'w') as out:
    for r in rows[a + 1:b + 1]:
        out.write("%8.2f us  %s\n" % ((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, r['Kernel_Name'][:110]))
PY
rm -rf gpurun_out/seq/t
