"""Per-kernel averages of the counters in rocprofv3 counter_collection CSVs: python tools/pmc_by_kernel.py FILE.csv [FILE.csv ...] [--match SUBSTR]"""
import collections
import csv
import sys

match = None
files = []
args = sys.argv[1:]
while args:
    a = args.pop(0)
    if a == "--match":
        match = args.pop(0)
    else:
        files.append(a)
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in files:
    per_dispatch = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        k = (r["Dispatch_Id"], r["Counter_Name"])
        per_dispatch[k] += float(r["Counter_Value"])          # (one row per XCD / instance: summed)
        names[r["Dispatch_Id"]] = r["Kernel_Name"]
    for (d, c), v in per_dispatch.items():
        n = names[d]
        if match and match not in n:
            continue
        e = acc[n.split("(")[0][:60]][c]
        e[0] += v; e[1] += 1
for n, cs in sorted(acc.items()):
    print(n, "  ".join("%s %.4g (x%d)" % (c, v[0] / v[1], v[1]) for c, v in sorted(cs.items())))
