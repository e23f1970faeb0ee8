#!/bin/bash
# Copy what tools/collect_profiles.sh left under gpurun_out/prof_final into profiles/ under the round's names and derive the
# traffic JSONs bench.py quotes.  Usage: bash tools/assemble_profiles.sh r03 2026-10-04
set -e
R=${1:-r06}; D=${2:-$(date +%F)}
O=gpurun_out/prof_final; P=profiles; H=$(git rev-parse --short HEAD)
cp $O/bench.json $P/${R}_bench_final.json
cp $O/bench_eager.json $P/${R}_bench_eager.json
for f in $O/bench_*.json; do
  b=$(basename $f .json); b=${b#bench_}
  [ "$b" = "eager" ] || cp $f $P/${R}_bench_$b.json
done
python - "$R" <<'PY'
import glob, json, sys
R = sys.argv[1]
rows = {}
for f in sorted(glob.glob('gpurun_out/prof_final/ab_fork*.json')):
    d = json.load(open(f))
    rows[f.split('/')[-1][:-5]] = dict(ms_per_step=d['ms_per_step'], vertices_per_s=d['value'], host_enqueue_ms_per_step=d['host_enqueue_ms_per_step'],
                                       step_execution=d['config']['step_execution'][:60])
json.dump(dict(what="same box, alternating runs of `OGL_FORK_BWD=0|1 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-e2e --graphs|--no-graphs`",
               runs=rows), open('profiles/%s_ab_fork.json' % R, 'w'), indent=1)
rows = {}
for f in sorted(glob.glob('gpurun_out/prof_final/ab_plan*.json')):
    d = json.load(open(f))
    rows[f.split('/')[-1][:-5]] = dict(ms_per_step=d['ms_per_step'], vertices_per_s=d['value'])
if rows:
    json.dump(dict(what="same box, alternating runs of `OGL_POOL_PLAN=0|1 python bench.py --steps 200 --warmup 60 --no-cpu-baseline --no-e2e --graphs` "
                        "(0: the pool backward's bucket pass on the backward's critical path; 1: planned by the forward pass on the side stream)",
                   runs=rows), open('profiles/%s_ab_pool_plan.json' % R, 'w'), indent=1)
PY
[ -f $O/parity_numbers.log ] && grep -E "200-step curve|accumulated gradient|out-of-tolerance logits|relative gradient errors|weights outside|forward parity|passed|failed" $O/parity_numbers.log > $P/${R}_parity_numbers.txt || true
[ -f $O/ab_$R.txt ] && python - "$R" <<'PY' || true
import glob, json, re, sys
R = sys.argv[1]
txt = open('gpurun_out/prof_final/ab_%s.txt' % R).read()
combos = dict(re.findall(r"^(c\d+) = (.*)$", txt, flags=re.M))
runs = {}
for f in sorted(glob.glob('gpurun_out/prof_final/ab_%s/*.json' % R)):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    k = f.split('/')[-1][:-5]
    runs[k] = dict(env=combos.get(k.split('_')[0], '?'), ms_per_step=d['ms_per_step'], vertices_per_s=d['value'])
json.dump(dict(what="same box, alternating runs of `env <switches> python bench.py --steps 200 --warmup 60 --graphs --no-cpu-baseline --no-e2e` "
                    "(tools/ab_combo.sh): the round's switches against the default (OGL_X=0 is a no-op)", runs=runs),
          open('profiles/%s_ab_experiments.json' % R, 'w'), indent=1)
PY
for f in micro.txt pmc_waits.txt step_timeline_pubmed_rbr.txt step_timeline_arxiv_rbr.txt x3_phase_probe.txt x3_clock_probe.txt half_wave_probe.txt; do
  [ -f $O/$f ] && grep -v amdgpu.ids $O/$f > $P/${R}_$f || true
done
python - "$R" <<'PY' || true
import glob, json, sys
R = sys.argv[1]
rows = {}
for f in sorted(glob.glob('gpurun_out/prof_final/ab_meanbits*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception:
        continue
    rows[f.split('/')[-1][:-5]] = dict(ms_per_step=d['ms_per_step'], vertices_per_s=d['value'],
                                       mean_backward_apply_ms=(d['kernels'].get('ogl_reduce_bwd_seg_apply') or {}).get('ms_per_step'))
if rows:
    json.dump(dict(what="same box, alternating runs of `OGL_POOL_MEAN_BITS=0|1 python bench.py --aggregator meanpool --steps 100 --warmup 20`", runs=rows),
              open('profiles/%s_ab_meanpool_sign_bits.json' % R, 'w'), indent=1)
PY
[ -f $O/block_build_probe.txt ] && grep -v amdgpu.ids $O/block_build_probe.txt > $P/${R}_block_build_probe.txt || true
[ -f $O/dw_pool0_probe.txt ] && cp $O/dw_pool0_probe.txt $P/${R}_dw_pool0_probe.txt || true
cp $O/kernel_stats.csv $P/${R}_rocprofv3_kernel_stats.csv
cp $O/kernel_stats_graph.csv $P/${R}_rocprofv3_kernel_stats_graph_replay.csv
cp $O/trace_by_grid.txt $P/${R}_kernel_trace_by_grid.txt
cp $O/pmc_by_grid.txt $P/${R}_pmc_fetch_write_by_kernel_grid.txt
cp $O/pmc_mfma_busy.txt $P/${R}_pmc_mfma_busy.txt
cp $O/step_timeline.txt $P/${R}_step_timeline.txt
cp $O/step_timeline_eager_traced.txt $P/${R}_step_timeline_eager_traced.txt
for w in reddit_pbr_forward arxiv_pbr_forward; do
  cp $O/pbr_${w}_kernel_stats.csv $P/${R}_pbr_${w}_kernel_stats.csv
  cp $O/pbr_${w}_trace_by_grid.txt $P/${R}_pbr_${w}_trace_by_grid.txt
  cp $O/pbr_${w}_pmc_by_grid.txt $P/${R}_pbr_${w}_pmc_by_grid.txt
done
python tools/make_pmc_traffic.py $H $D rbr=$O/pmc_by_grid.txt > $P/${R}_pmc_traffic.json
python tools/make_pmc_traffic.py $H $D reddit_pbr_forward=$O/pbr_reddit_pbr_forward_pmc_by_grid.txt arxiv_pbr_forward=$O/pbr_arxiv_pbr_forward_pmc_by_grid.txt > $P/${R}_pbr_pmc_traffic.json
ls $P | grep -c "^${R}_"
