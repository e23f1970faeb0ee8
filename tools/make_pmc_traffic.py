"""Per-launch HBM-side traffic JSONs (what bench.py quotes as `roofline.traffic`) from the (kernel, grid) PMC tables that
tools/summarize_trace.py prints (rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes; both counters are in KB;
FETCH_SIZE is doubled: gfx950 reports half the bytes of 16-B/lane coalesced reads, MI355X_MICROARCH.md).  Usage:
  python tools/make_pmc_traffic.py <head> <date> rbr=<pmc_by_grid.txt> [<workload>=<pmc_by_grid.txt> ...] > out.json"""
import json
import re
import sys


def parse(path):
    rows = []
    for line in open(path):
        m = re.match(r"^(.*?)\s+(\d+)\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)%(.*)$", line.rstrip("\n"))
        if not m:
            continue
        name, grid, calls, avg_us, total_ms, pct, rest = m.groups()
        c = dict(re.findall(r"(\w+)=([\d.]+)/launch", rest))
        rows.append(dict(name=name.strip(), grid=int(grid), calls=int(calls), avg_us=float(avg_us), total_ms=float(total_ms),
                         fetch_kb=float(c["FETCH_SIZE"]) if "FETCH_SIZE" in c else None,
                         write_kb=float(c["WRITE_SIZE"]) if "WRITE_SIZE" in c else None))
    return rows


def entry(r, **extra):
    fetch = 2 * 1024 * r["fetch_kb"]
    write = 1024 * (r["write_kb"] or 0.0)
    d = dict(kernel=r["name"], grid=r["grid"], launches=r["calls"], fetch_kb_raw=r["fetch_kb"], fetch_bytes_corrected=round(fetch),
             write_bytes=round(write), traffic_bytes=round(fetch + write), avg_launch_us=r["avg_us"])
    d.update(extra)
    return d


def first(rows, pred, full_only=False):
    """The kernel (template instantiation) with the most total time among those `pred` accepts — its launches pooled over the
    grid sizes they ran on (an aggregator's grid follows the batch's block size), counters averaged per launch."""
    groups = {}
    for r in rows:
        if r["fetch_kb"] is not None and pred(r):
            groups.setdefault(r["name"], []).append(r)
    if not groups:
        return None
    name, rs = max(groups.items(), key=lambda kv: sum(r["total_ms"] for r in kv[1]))
    if full_only:                       # a fused inference pass: its FULL chunks only (the last chunk of a pass is a partial one)
        top = max(r["grid"] for r in rs)
        rs = [r for r in rs if r["grid"] >= 0.85 * top]
    n = sum(r["calls"] for r in rs)
    avg = lambda k: sum((r[k] or 0.0) * r["calls"] for r in rs) / n          # noqa: E731
    return dict(name=name, grid=("%d..%d" % (min(r["grid"] for r in rs), max(r["grid"] for r in rs))) if len(rs) > 1 else rs[0]["grid"],
                calls=n, avg_us=round(avg("avg_us"), 2), total_ms=sum(r["total_ms"] for r in rs), fetch_kb=round(avg("fetch_kb"), 1),
                write_kb=round(avg("write_kb"), 1))


def main():
    head, date = sys.argv[1], sys.argv[2]
    out = {"_comment": "HBM-side bytes per launch from rocprofv3 PMC passes (separate --pmc FETCH_SIZE and --pmc WRITE_SIZE runs, MI355X). "
                       "FETCH_SIZE is in KB and is doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of 16-B/lane coalesced "
                       "reads); WRITE_SIZE (KB) is exact.  Built by tools/make_pmc_traffic.py from the per-(kernel, grid) tables beside it.",
           "collected": date, "head": head}
    for arg in sys.argv[3:]:
        wl, path = arg.split("=", 1)
        rows = parse(path)
        agg = first(rows, lambda r: r["name"].startswith("k_reduce_fwd_v4") or r["name"].startswith("k_reduce_fwd_max_half"), full_only=(wl != "rbr"))
        if wl == "rbr":
            if agg:
                out["k_reduce_fwd_v4_L0"] = entry(agg)
            f = first(rows, lambda r: re.match(r"k_gemm_x3p<4, 2, 2, 2, 2, false, false, false(, 0, [01])?(, (true|false))?>", r["name"]))
            if f:
                out["k_gemm_x3_fwd_pool0"] = entry(f)
            # the layer-0 weight gradient: k-major B, dy^T a transposed image (256 x 128 tile since round 3, 128 x 128 before)
            w = first(rows, lambda r: re.match(r"k_gemm_x3p<(2, 4, 2, 1, 3|4, 2, 2, 2, 2), false, true, false(, 0, 0)?(, (true|false))?>", r["name"]))
            if w:
                out["k_gemm_x3_bwwk_pool0"] = entry(w)
        else:
            d = {}
            if agg:
                d["k_reduce_fwd_v4_L0"] = entry(agg)
            t = first(rows, lambda r: r["name"].startswith("k_gemm_x3p") and r["avg_us"] > 300)
            if t:
                d["k_gemm_x3_tables"] = entry(t)
            out[wl] = d
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
