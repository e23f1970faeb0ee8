"""MFMA-pipe utilisation per (kernel, grid) from a rocprofv3 PMC pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace).
Usage: python tools/summarize_pmc_mfma.py <kernel_trace.csv> <counter_collection.csv>"""
import collections
import csv
import sys


def main():
    trace, pmc = sys.argv[1], sys.argv[2]
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        key = (r["Kernel_Name"].replace("void ", "")[:100], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
        dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pmc)):
        key = (r["Kernel_Name"].replace("void ", "")[:100], int(r["Grid_Size"]))
        cnt[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs); wait fractions are per wave-cycle;")
    print("# eff_clock = GRBM_GUI_ACTIVE / 8 / duration (reads high on launches shorter than ~0.3 ms: MI355X_MICROARCH.md)")
    for key, c in sorted(cnt.items(), key=lambda kv: -sum(dur.get(kv[0], [0]))):
        if "gemm" not in key[0]:
            continue
        m = {k: sum(v) / len(v) for k, v in c.items()}
        d = dur.get(key, [0.0])
        us = sum(d) / len(d)
        gui = m.get("GRBM_GUI_ACTIVE", 0.0) / 8
        wc = m.get("SQ_WAVE_CYCLES", 1.0)
        print("%-60s grid %8d n %4d dur_us %8.1f  mfma_util %.3f  wait_any/wave %.3f wait_inst/wave %.3f eff_clock_GHz %.2f" % (
            key[0][:60], key[1], len(d), us, m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024 / gui if gui else 0.0,
            m.get("SQ_WAIT_ANY", 0.0) / wc, m.get("SQ_WAIT_INST_ANY", 0.0) / wc, gui / us / 1e3 if us else 0.0))


if __name__ == "__main__":
    main()
