#!/usr/bin/env python3
"""Per-kernel matrix-pipe and LDS figures from one rocprofv3 --pmc pass (development / evidence tool).

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
              --kernel-trace --output-format csv -d gpurun_out/pmc_u -o u -- python3 bench.py --steps 2 --warmup 1 --no-roofline --no-cpu-baseline
    python tools/pmc_util.py gpurun_out/pmc_u profiles/r02_pmc_mfma_lds_per_kernel.csv

Columns (per launch, averaged over the launches of a kernel): duration under the counters, the raw counters, and
  mfma_share  = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024): matrix-pipe busy cycles summed over the 1024 SIMDs against the
                cycles the chip was active (GRBM_GUI_ACTIVE is summed over the 8 XCDs; check: ffn_chain_kernel reports exactly
                9,830,400 MFMAs x 16 cycles = 157,286,400);
  clock_ghz   = GRBM_GUI_ACTIVE / 8 / duration;
  lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (extra cycles over all LDS-array cycles, MI355X_MICROARCH.md, LDS).
"""
import csv, glob, os, sys
from collections import defaultdict

def main():
    d, out = sys.argv[1:3]
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *counter_collection.csv under {d}")
    per = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    dur = defaultdict(float)
    seen = set()
    for f in files:
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "ge2e" not in k:
                continue
            per[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
            if (k, r["Dispatch_Id"]) not in seen:
                seen.add((k, r["Dispatch_Id"]))
                dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    names = ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAVES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "GRBM_GUI_ACTIVE"]
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches", "avg_us_under_pmc"] + [n + "_per_launch" for n in names] + ["mfma_share_of_simd_cycles", "clock_ghz", "lds_conflict_share"])
        for k in sorted(per, key=lambda k: -dur[k]):
            n = max(len(disp[k]), 1)
            c = {m: per[k].get(m, 0.0) / n for m in names}
            # SIMD-cycles available to the launch: chip-active cycles x 1024 SIMDs (256 CUs x 4); MFMA busy is summed over SIMDs
            mf = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0) if c["GRBM_GUI_ACTIVE"] else 0.0
            ghz = c["GRBM_GUI_ACTIVE"] / 8.0 / (dur[k] / n) / 1e3 if dur[k] else 0.0
            lc = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"] if c["SQ_LDS_IDX_ACTIVE"] else 0.0
            w.writerow([k, n, round(dur[k] / n, 1)] + [round(c[m], 1) for m in names] + [round(mf, 4), round(ghz, 3), round(lc, 4)])
            print(f"{k[:72]:72s} n={n:3d} {dur[k] / n:8.1f} us  mfma {mf:6.3f}  {ghz:5.2f} GHz  lds-conflict {lc:6.3f}")

if __name__ == "__main__":
    main()
