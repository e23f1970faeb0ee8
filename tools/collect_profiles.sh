#!/bin/bash
# Collect the round's measurement artefacts on a GPU box (run from the repo root through gpurun); outputs under gpurun_out/prof_final.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_final
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 python bench.py --workload reddit_pbr_forward --no-cpu-baseline > $O/bench_pbr_forward.json 2> $O/bench_pbr.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $O/trace.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/pmc_write.log 2>&1
cd $R
python tools/summarize_trace.py $(ls $O/trace/*/*kernel_trace.csv | head -1) > $O/trace_by_grid.txt
python tools/summarize_trace.py $(ls $O/pmc_fetch/*/*kernel_trace.csv | head -1) $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) > $O/pmc_by_grid.txt
cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
rm -rf $O/trace $O/pmc_fetch $O/pmc_write
head -c 600 $O/bench.json
