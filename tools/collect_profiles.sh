#!/bin/bash
# Collect the round's measurement artefacts on a GPU box (run from the repo root through gpurun); outputs under gpurun_out/prof_final.
# Every rocprofv3 command puts the program itself after `--` (python3 bench.py ...): no env / bash -c hop.
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof_final
mkdir -p $O
cd $R
PART=${1:-all}
part_A1() {
  timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err < /dev/null
  timeout -k 10 300 python bench.py --no-graphs --no-cpu-baseline --no-e2e > $O/bench_eager.json 2> $O/bench_eager.err < /dev/null
  for w in reddit_pbr_forward arxiv_pbr_forward arxiv_rbr pubmed_rbr pubmed_settings arxiv_settings bitcoin_settings reddit_settings reddit_settings_pbr_forward; do
    timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err < /dev/null
  done
  timeout -k 10 300 python bench.py --workload arxiv_pbr_snapshot --steps 30 --warmup 8 > $O/bench_arxiv_pbr_snapshot.json 2> $O/bench_arxiv_pbr_snapshot.err < /dev/null
  # the in-repo aggregator modes at the Reddit rung (round 4: first-class workloads, mean backward as a planned segmented gather)
  for a in meanpool mean; do
    timeout -k 10 300 python bench.py --aggregator $a --no-cpu-baseline --no-e2e > $O/bench_reddit_rbr_$a.json 2> $O/bench_reddit_rbr_$a.err < /dev/null
  done
  # the numbers the round-4 parity tests print (loss curves, x6-vs-fp32 counts, gradient errors at the Reddit rung)
  timeout -k 10 600 python -m pytest tests/test_gpu_rungs.py tests/test_gpu_fullsize.py tests/test_gpu_round4.py -q -s -k "200_step or no_worse or inrepo_modes or two_part or fused_output_layer_step" > $O/parity_numbers.log 2>&1 < /dev/null || true
}
part_A2() {
  # same-box A/B of the switches that are left, inside the replayed step
  bash tools/ab_combo.sh r06 2 "OGL_X=0" "OGL_X3_STAGGER=0" "OGL_X3_STAGGER=1" "OGL_DUAL_DW=0" "OGL_SLAB_ADAM=0" "OGL_POOL_PLAN=0" "OGL_FORK_BWD=0" > $O/ab_r06.txt 2>&1 || true
  cp -r gpurun_out/ab_r06 $O/ab_r06 2> /dev/null || true
  # where a step of the layer-0 image GEMMs goes: per-phase cycle sums from the diagnostic build (tools/build_variant.py phase -DOGL_X3_PHASE_STAMPS),
  # staggered and in lockstep; the in-kernel clock on random and on all-zero operands (the chip's power management, not the kernel)
  (for st in 1 0; do echo "== OGL_X3_STAGGER=$st"; OGL_X3_STAGGER=$st timeout -k 10 150 python tools/x3_phase_probe.py 2>&1 | grep -v amdgpu.ids; done) > $O/x3_phase_probe.txt 2>&1 || true
  (timeout -k 10 100 python tools/gemm_x3_bench.py clock; X3_ZERO=1 timeout -k 10 100 python tools/gemm_x3_bench.py clock) 2>&1 | grep -v amdgpu.ids > $O/x3_clock_probe.txt || true
  # the 32-seed rungs' traced steps
  for w in pubmed_rbr arxiv_rbr; do
    bash tools/trace_wl.sh $w > /dev/null 2>&1 || true
    cp gpurun_out/trace_$w/timeline.txt $O/step_timeline_$w.txt 2> /dev/null || true
  done
  # the narrow-row max aggregator: two rows per wave-instruction against one (arxiv-like priority forward)
  timeout -k 10 200 python tools/half_wave_probe.py > $O/half_wave_probe.txt 2>&1 || true
  (timeout -k 5 60 tools/micro/lds_atomics; timeout -k 5 60 tools/micro/grid_barrier; timeout -k 5 60 tools/micro/mfma_rate) > $O/micro.txt 2>&1 || true
  bash tools/pmc_waits.sh > /dev/null 2>&1 || true
  cp gpurun_out/pmc_waits.txt $O/pmc_waits.txt 2> /dev/null || true
  timeout -k 10 200 python tools/dw_pool0_probe.py 2>&1 | grep -v amdgpu.ids > $O/dw_pool0_probe.txt || true
}
part_B() {
  cd /tmp && export TMPDIR=/tmp
  # the traced / counted runs enqueue eagerly (--no-graphs): the same kernels at the batch's own sizes, one dispatch per launch
  B="$R/bench.py --no-cpu-baseline --no-e2e"
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $B --steps 50 --warmup 5 --no-graphs > $O/trace.log 2>&1 < /dev/null
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_graph -- python3 $B --steps 50 --warmup 60 --graphs > $O/trace_graph.log 2>&1 < /dev/null
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $B --steps 10 --warmup 2 --no-graphs > $O/pmc_fetch.log 2>&1 < /dev/null
  timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $B --steps 10 --warmup 2 --no-graphs > $O/pmc_write.log 2>&1 < /dev/null
  timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $B --steps 10 --warmup 2 --no-graphs > $O/pmc_mfma.log 2>&1 < /dev/null
  # the PBR priority-forward workloads (what dominates the PBR rungs): kernel stats + FETCH / WRITE passes each
  for w in reddit_pbr_forward arxiv_pbr_forward; do
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_$w -- python3 $B --workload $w --steps 50 --warmup 5 > $O/trace_$w.log 2>&1 < /dev/null
    timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmcf_$w -- python3 $B --workload $w --steps 20 --warmup 2 > $O/pmcf_$w.log 2>&1 < /dev/null
    timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmcw_$w -- python3 $B --workload $w --steps 20 --warmup 2 > $O/pmcw_$w.log 2>&1 < /dev/null
  done
  cd $R
  T=$(ls $O/trace/*/*kernel_trace.csv | head -1)
  python tools/summarize_trace.py $T > $O/trace_by_grid.txt
  python tools/step_timeline.py $T > $O/step_timeline_eager_traced.txt
  # the replayed step: no host in the loop (the eager step above is host-bound UNDER THE TRACER: ~130 us of device idle per step)
  python tools/step_timeline.py $(ls $O/trace_graph/*/*kernel_trace.csv | head -1) > $O/step_timeline.txt
  python tools/summarize_trace.py $(ls $O/pmc_fetch/*/*kernel_trace.csv | head -1) $(ls $O/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $O/pmc_write/*/*counter_collection.csv | head -1) > $O/pmc_by_grid.txt
  cp $(ls $O/trace/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
  cp $(ls $O/trace_graph/*/*kernel_stats.csv | head -1) $O/kernel_stats_graph.csv
  python tools/summarize_pmc_mfma.py $(ls $O/pmc_mfma/*/*kernel_trace.csv | head -1) $(ls $O/pmc_mfma/*/*counter_collection.csv | head -1) > $O/pmc_mfma_busy.txt 2>&1 || true
  for w in reddit_pbr_forward arxiv_pbr_forward; do
    cp $(ls $O/trace_$w/*/*kernel_stats.csv | head -1) $O/pbr_${w}_kernel_stats.csv
    python tools/summarize_trace.py $(ls $O/trace_$w/*/*kernel_trace.csv | head -1) > $O/pbr_${w}_trace_by_grid.txt
    python tools/summarize_trace.py $(ls $O/pmcf_$w/*/*kernel_trace.csv | head -1) $(ls $O/pmcf_$w/*/*counter_collection.csv | head -1) $(ls $O/pmcw_$w/*/*counter_collection.csv | head -1) > $O/pbr_${w}_pmc_by_grid.txt
  done
  rm -rf $O/trace $O/trace_graph $O/pmc_fetch $O/pmc_write $O/pmc_mfma $O/trace_reddit_pbr_forward $O/trace_arxiv_pbr_forward $O/pmcf_* $O/pmcw_*
}
part_C() {
  timeout -k 10 300 python bench.py --force-dist --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $O/bench_force_dist.json 2> $O/bench_force_dist.err < /dev/null
  timeout -k 10 300 python bench.py --force-dist --no-graphs --no-variants --steps 100 --warmup 10 --no-cpu-baseline --no-e2e > $O/bench_force_dist_eager.json 2> /dev/null < /dev/null
  timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --steps 50 --warmup 5 --no-cpu-baseline > $O/bench_2rank_gloo.json 2> /dev/null < /dev/null
  timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --steps 20 --warmup 2 --workload reddit_pbr_forward --partition features > $O/bench_2rank_gloo_pbr_partitioned.json 2> /dev/null < /dev/null
  timeout -k 10 300 python bench.py --gpus 2 --dist-backend gloo --steps 50 --warmup 5 --scaling strong --no-cpu-baseline > $O/bench_2rank_gloo_strong.json 2> /dev/null < /dev/null
  for i in 1 2; do
    OGL_FORK_BWD=0 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --graphs > $O/ab_fork0_graphs_$i.json 2> /dev/null < /dev/null
    OGL_FORK_BWD=1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --graphs > $O/ab_fork1_graphs_$i.json 2> /dev/null < /dev/null
    OGL_FORK_BWD=0 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-graphs > $O/ab_fork0_eager_$i.json 2> /dev/null < /dev/null
    OGL_FORK_BWD=1 timeout -k 10 200 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --no-graphs > $O/ab_fork1_eager_$i.json 2> /dev/null < /dev/null
  done
  for i in 1 2 3; do
    OGL_POOL_PLAN=0 timeout -k 10 200 python bench.py --steps 200 --warmup 60 --no-cpu-baseline --no-e2e --graphs > $O/ab_plan0_graphs_$i.json 2> /dev/null < /dev/null
    OGL_POOL_PLAN=1 timeout -k 10 200 python bench.py --steps 200 --warmup 60 --no-cpu-baseline --no-e2e --graphs > $O/ab_plan1_graphs_$i.json 2> /dev/null < /dev/null
  done
}
# a gpurun call lasts at most 20 minutes: collect in parts (A1: bench lines + parity numbers, A2: A/Bs + probes, B: rocprofv3 traces + counters,
# C: replica / two-rank lines + fork / plan A/Bs); "all" runs them in order (for a box without that limit)
case $PART in A1) part_A1;; A2) part_A2;; B) part_B;; C) part_C;; all) part_A1; part_A2; part_B; part_C;; esac
head -c 300 $O/bench.json 2> /dev/null || true
