"""Print the headline fields of bench.py JSON lines read from files given on the command line."""
import json
import sys

for path in sys.argv[1:]:
    try:
        d = json.loads(open(path).read().strip().splitlines()[-1])
    except Exception as e:      # noqa: BLE001
        print(path, "unreadable:", e)
        continue
    g = d.get("graph_mode") or {}
    print(path, d["value"], d["ms_per_step"], "graph_mode", g.get("ms_per_step"), "host", d.get("host_enqueue_ms_per_step"))
