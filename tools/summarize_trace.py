"""Group a rocprofv3 kernel_trace.csv / counter_collection.csv by (kernel, grid size): average duration
and, when a PMC pass is given, the per-launch counter value.  A persistent kernel launches every product on the same
grid; when one (kernel, grid) group holds launches of clearly different lengths (longest > 2.5 x shortest) it is split at
the geometric mean into "[long]" and "[short]" — e.g. the layer-0 weight gradient next to the three n1-row ones.  Usage:
  python tools/summarize_trace.py <kernel_trace.csv> [<counter_collection.csv> ...]"""
import collections
import csv
import math
import sys


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 100 else name[:97] + "..."


def classify(groups):
    """groups: key -> [(duration_us, payload)] ; returns key' -> [(duration_us, payload)] with bimodal groups split."""
    out = collections.defaultdict(list)
    for key, items in groups.items():
        durs = [d for d, _ in items]
        lo, hi = min(durs), max(durs)
        if len(durs) >= 4 and lo > 0 and hi > 2.5 * lo:
            thr = math.sqrt(lo * hi)
            for d, pl in items:
                out[(key[0] + (" [long]" if d >= thr else " [short]"), key[1])].append((d, pl))
        else:
            out[key] = list(items)
    return out


def main():
    trace = sys.argv[1]
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        key = (short(r["Kernel_Name"]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
        groups[key].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, None))
    groups = classify(groups)
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[2:]:
        cg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            cg[key].append(((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, (r["Counter_Name"], float(r["Counter_Value"]))))
        for key, items in classify(cg).items():
            for _, (cname, val) in items:
                counters[key][cname].append(val)
    tot = sum(sum(d for d, _ in v) for v in groups.values())
    print("%-108s %10s %7s %12s %12s %7s" % ("kernel", "grid", "calls", "avg_us", "total_ms", "pct"))
    for key, v in sorted(groups.items(), key=lambda kv: -sum(d for d, _ in kv[1])):
        ds = [d for d, _ in v]
        line = "%-108s %10d %7d %12.2f %12.3f %6.2f%%" % (key[0], key[1], len(ds), sum(ds) / len(ds), sum(ds) / 1e3, 100 * sum(ds) / tot)
        for cname, cv in sorted(counters.get(key, {}).items()):
            line += "  %s=%.1f/launch" % (cname, sum(cv) / len(cv))
        print(line)


if __name__ == "__main__":
    main()
