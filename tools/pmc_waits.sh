#!/bin/bash
# Two PMC passes that attribute the image-GEMM waves' waits (tools/summarize_pmc_waits.py); outputs gpurun_out/pmc_waits.txt
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/pmc_waits
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py --no-cpu-baseline --no-e2e"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC --kernel-trace --output-format csv -d $O/a -- python3 $B --steps 10 --warmup 2 --no-graphs > $O/a.log 2>&1 < /dev/null
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --kernel-trace --output-format csv -d $O/b -- python3 $B --steps 10 --warmup 2 --no-graphs > $O/b.log 2>&1 < /dev/null
cd $R
python tools/summarize_pmc_waits.py $(ls $O/a/*/*kernel_trace.csv | head -1) $(ls $O/a/*/*counter_collection.csv | head -1) $(ls $O/b/*/*counter_collection.csv | head -1) > $R/gpurun_out/pmc_waits.txt 2>&1
rm -rf $O/a $O/b
tail -3 $O/a.log $O/b.log
cat $R/gpurun_out/pmc_waits.txt
