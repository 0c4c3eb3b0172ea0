"""Summarise a rocprofv3 SQ counter pass over bench.py into per-kernel-family means per launch.

    python tools/pmc_sq.py <pmc_dir> [<pmc_dir2> ...] > profiles/rNN/sq_counters.txt
Counters (MI355X_MICROARCH.md, rocprofv3 PMC slots): SQ_WAVE_CYCLES / SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY count
quad-cycles summed over all waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles the matrix pipe of a SIMD is busy, summed over SIMDs;
SQ_BUSY_CYCLES the cycles an SQ (one per CU... per SE on some parts) has a wave.  Derived columns:
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (SQ_BUSY_CU_CYCLES-equivalent)  -> printed against GRBM_GUI_ACTIVE/8 x 1024 SIMDs when GRBM is present
  wait%, inst_wait%, active% = share of SQ_WAVE_CYCLES (they are disjoint and add up to ~100 %)
  lds_conflict% = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE"""
import collections, csv, glob, sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from pmc_traffic import family


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[family(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    names = sorted({c for v in agg.values() for c in v})
    print("# means per launch; counters:", ", ".join(names))
    rows = []
    for k, v in agg.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        n = max(len(x) for x in v.values())
        rows.append((m.get("SQ_WAVE_CYCLES", 0.0) * n, k, n, m))
    for _, k, n, m in sorted(rows, reverse=True):
        wc = m.get("SQ_WAVE_CYCLES", 0.0)
        parts = [f"{k:34s} n={n:5d}"]
        if wc:
            parts.append(f"wave_cyc {wc:12.0f}  wait {100 * m.get('SQ_WAIT_ANY', 0) / wc:5.1f}%  inst_wait {100 * m.get('SQ_WAIT_INST_ANY', 0) / wc:5.1f}%  "
                         f"active {100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc:5.1f}%")
        if "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            parts.append(f"mfma_busy_cyc {m['SQ_VALU_MFMA_BUSY_CYCLES']:12.0f}")
            if m.get("GRBM_GUI_ACTIVE"):
                parts.append(f"mfma_busy {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (m['GRBM_GUI_ACTIVE'] / 8 * 1024):5.1f}% of SIMD-cycles")
            if m.get("SQ_BUSY_CYCLES"):
                parts.append(f"sq_busy_cyc {m['SQ_BUSY_CYCLES']:10.0f}")
        if "SQ_VALU_MFMA_COEXEC_CYCLES" in m:
            parts.append(f"valu_mfma_coexec {m['SQ_VALU_MFMA_COEXEC_CYCLES']:11.0f}")
        if m.get("SQ_LDS_IDX_ACTIVE"):
            parts.append(f"lds_active {m['SQ_LDS_IDX_ACTIVE']:11.0f}  lds_conflict {100 * m.get('SQ_LDS_BANK_CONFLICT', 0) / m['SQ_LDS_IDX_ACTIVE']:5.1f}%")
        shown = {"SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE",
                 "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"}
        for c in sorted(m):
            if c not in shown:
                parts.append(f"{c[3:] if c.startswith('SQ_') else c} {m[c]:.0f}")
        print("  ".join(parts))


if __name__ == "__main__":
    main()
