"""Where the waves of the image-GEMM launches wait: per (kernel, grid) ratios from TWO rocprofv3 PMC passes of the same command
  pass A: --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC
  pass B: --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM
(every value divided by that pass's SQ_WAVE_CYCLES: fractions of a wave-cycle, summed over the 12 waves of a block — 8 multipliers and 4
movers, which the counters cannot tell apart: the movers' share of the wave-cycles is 1/3, spent almost entirely in SQ_WAIT_ANY / VMEM).
Usage: python tools/summarize_pmc_waits.py <kernel_trace.csv> <countersA.csv> <countersB.csv>"""
import collections
import csv
import sys


def load(path):
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        key = (r["Kernel_Name"].replace("void ", "")[:100], int(r["Grid_Size"]))
        cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {n: sum(v) / len(v) for n, v in c.items()} for k, c in cnt.items()}


def main():
    trace, pa, pb = sys.argv[1:4]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        key = (r["Kernel_Name"].replace("void ", "")[:100], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    A, B = load(pa), load(pb)
    print("# fractions of SQ_WAVE_CYCLES (all 12 waves of a block together; the 4 mover waves are 1/3 of them)")
    for key in sorted(A, key=lambda k: -sum(dur.get(k, [0]))):
        if "gemm" not in key[0]:
            continue
        a, b = A[key], B.get(key, {})
        wa, wb = a.get("SQ_WAVE_CYCLES", 1.0) or 1.0, b.get("SQ_WAVE_CYCLES", 1.0) or 1.0
        d = dur.get(key, [0.0])
        f = lambda m, k, w: m.get(k, 0.0) / w                                   # noqa: E731
        print("%-62s grid %7d n %3d dur_us %7.1f | wait_any %.3f wait_inst_any %.3f wait_inst_lds %.3f | active: lds %.3f vmem %.3f valu %.3f misc %.3f"
              " | lds: insts/kcycle %.2f bank_conflict %.4f idx_active %.3f data_fifo_full %.4f cmd_fifo_full %.4f | level: lds %.2f vmem %.2f" % (
                  key[0][:62], key[1], len(d), sum(d) / len(d), f(a, "SQ_WAIT_ANY", wa), f(a, "SQ_WAIT_INST_ANY", wa), f(a, "SQ_WAIT_INST_LDS", wa),
                  f(a, "SQ_ACTIVE_INST_LDS", wa), f(a, "SQ_ACTIVE_INST_VMEM", wa), f(a, "SQ_ACTIVE_INST_VALU", wa), f(a, "SQ_ACTIVE_INST_MISC", wa),
                  1e3 * f(b, "SQ_INSTS_LDS", wb), f(b, "SQ_LDS_BANK_CONFLICT", wb), f(b, "SQ_LDS_IDX_ACTIVE", wb), f(b, "SQ_LDS_DATA_FIFO_FULL", wb),
                  f(b, "SQ_LDS_CMD_FIFO_FULL", wb), f(b, "SQ_INST_LEVEL_LDS", wb), f(b, "SQ_INST_LEVEL_VMEM", wb)))


if __name__ == "__main__":
    main()
