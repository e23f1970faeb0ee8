"""Timeline of ONE steady-state train step from a rocprofv3 kernel_trace.csv: every dispatch between two consecutive Adam
launches with its start offset, duration, the idle gap on the device before it (no kernel of ANY queue running), and how much
of it ran concurrently with a kernel of another queue (the forked backward).  Usage:
  python tools/step_timeline.py <kernel_trace.csv> [step index, default: the median-length step]"""
import csv
import sys


def short(n):
    n = n.replace("void ", "")
    return n if len(n) < 70 else n[:67] + "..."


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), int(r.get("Grid_Size", 0) or 0)))
    rows.sort()
    # a step ends with its LAST optimiser launch (since round 4 the optimiser may run in two parts: an early one on the side branch,
    # the rest at the end): the Adam launch that is followed by the next step's staging / weight-image launch (or by nothing)
    adam = [i for i, r in enumerate(rows) if "k_adam_multi" in r[2] or r[2].startswith("k_adam(")]     # (not k_adam_prepare)
    firsts = ("k_stage_segments", "k_x3_split_multi", "k_sample_layer", "k_block_", "k_sample_blocks_small")
    last = [i for i in adam if i + 1 >= len(rows) or any(f in rows[i + 1][2] for f in firsts)]
    if len(last) >= 2:
        adam = last
    steps = [(adam[k], adam[k + 1]) for k in range(len(adam) - 1)]
    steps = [(a, b) for a, b in steps if b - a >= 5] or steps       # (whole train steps, not the optimiser's own two launches; a 32-seed
    # step is seven dispatches since the end of round 5)
    lens = sorted((rows[b][1] - rows[a][1], k) for k, (a, b) in enumerate(steps))
    pick = int(sys.argv[2]) if len(sys.argv) > 2 else lens[len(lens) // 2][1]
    a, b = steps[pick]
    seg = rows[a + 1:b + 1]
    t0 = rows[a][1]
    print("step %d of %d: %.1f us from the end of one step's last Adam launch to the end of the next's; %d dispatches" % (pick, len(steps), (rows[b][1] - t0) / 1e3, len(seg)))
    busy_until = t0
    idle = overlap_total = 0.0
    print("%9s %9s %8s %8s %5s  %s" % ("start_us", "dur_us", "gap_us", "ovl_us", "queue", "kernel [grid]"))
    for i, (s, e, n, q, g) in enumerate(seg):
        gap = max(0, s - busy_until) / 1e3
        idle += gap
        ovl = 0
        for j, (s2, e2, n2, q2, g2) in enumerate(seg):
            if j != i and q2 != q:
                ovl += max(0, min(e, e2) - max(s, s2))
        overlap_total += ovl / 1e3
        print("%9.1f %9.1f %8.1f %8.1f %5s  %s [%d]" % ((s - t0) / 1e3, (e - s) / 1e3, gap, ovl / 1e3, q, short(n), g))
        busy_until = max(busy_until, e)
    print("device idle inside the step: %.1f us; kernel time summed: %.1f us; cross-queue overlap (counted on both sides): %.1f us"
          % (idle, sum(e - s for s, e, *_ in seg) / 1e3, overlap_total))


if __name__ == "__main__":
    main()
