O=gpurun_out/prof_small; mkdir -p $O
for w in arxiv_rbr pubmed_rbr pubmed_settings arxiv_settings bitcoin_settings; do
  timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_$w.json 2> $O/bench_$w.err < /dev/null
done
for w in pubmed_rbr arxiv_rbr; do
  bash tools/ab_wl.sh OGL_SAMPLE_PIPELINE $w 3 > $O/ab_sample_pipeline_$w.txt 2>&1 || true
  bash tools/trace_wl.sh $w > /dev/null 2>&1 || true
  cp gpurun_out/trace_$w/timeline.txt $O/step_timeline_$w.txt 2> /dev/null || true
done
for w in pubmed_settings bitcoin_settings; do bash tools/ab_wl.sh OGL_SAMPLE_PIPELINE $w 3 > $O/ab_sample_pipeline_$w.txt 2>&1 || true; done
python tools/bench_brief.py $O/bench_*.json
