"""From a rocprofv3 kernel_trace.csv of a replayed run: the dispatches between the last Adam launch of one loader's last step and the
first staging launch of the next loader's first step (sampling + block build + their read-backs), with gaps.
Usage: python tools/trace_loader_gap.py <kernel_trace.csv> [which gap, default the last]"""
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
gaps = []
i = 0
while i < len(rows):
    if "k_sample_layer_batched" in rows[i][2]:
        j = i
        while j > 0 and "k_adam" not in rows[j][2]:
            j -= 1
        k = i
        while k < len(rows) and "k_stage_segments" not in rows[k][2]:
            k += 1
        if k < len(rows):
            gaps.append((j, k))
        i = k + 1
    else:
        i += 1
# merge: a loader has two sampler launches; keep gaps whose start differs
uniq = []
for g in gaps:
    if not uniq or uniq[-1][0] != g[0]:
        uniq.append(g)
pick = int(sys.argv[2]) if len(sys.argv) > 2 else len(uniq) - 1
j, k = uniq[pick]
t0 = rows[j][1]
print("loader gap %d of %d: %.1f us from the end of the last Adam launch to the start of the next step's staging launch" % (pick, len(uniq), (rows[k][0] - t0) / 1e3))
prev = t0
for s, e, n in rows[j + 1:k + 1]:
    print("%9.1f %8.1f gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, n[:90]))
    prev = max(prev, e)
