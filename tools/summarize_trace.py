"""Group a rocprofv3 kernel_trace.csv / counter_collection.csv by (kernel, grid size): average duration
and, when a PMC pass is given, the per-launch counter value.  Usage:
  python tools/summarize_trace.py <kernel_trace.csv> [<counter_collection.csv> ...]"""
import collections
import csv
import sys


def short(name):
    name = name.replace("void ", "")
    return name if len(name) < 100 else name[:97] + "..."


def main():
    trace = sys.argv[1]
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        key = (short(r["Kernel_Name"]), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
        groups[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    counters = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[2:]:
        for r in csv.DictReader(open(path)):
            key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
            counters[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    tot = sum(sum(v) for v in groups.values())
    print("%-100s %10s %7s %12s %12s %7s" % ("kernel", "grid", "calls", "avg_us", "total_ms", "pct"))
    for key, v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        line = "%-100s %10d %7d %12.2f %12.3f %6.2f%%" % (key[0], key[1], len(v), sum(v) / len(v), sum(v) / 1e3, 100 * sum(v) / tot)
        for cname, cv in sorted(counters.get(key, {}).items()):
            line += "  %s=%.1f/launch" % (cname, sum(cv) / len(cv))
        print(line)


if __name__ == "__main__":
    main()
