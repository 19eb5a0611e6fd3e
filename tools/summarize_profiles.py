"""Developer tool: condense the rocprofv3 output of tools/collect_profiles.sh into the small CSVs kept under profiles/."""
import collections
import csv
import glob
import os
import re
import sys

out = sys.argv[1]


def short(name: str) -> str:
    name = re.sub(r"^void ", "", name)
    name = name.replace("hipac::", "")
    return name[:150]


for kt in ("kt_bench", "kt_wsi", "kt_simclr", "kt_simclr_amp", "kt_tiff"):
    for f in glob.glob(os.path.join(out, kt, "**", "*kernel_stats.csv"), recursive=True):
        rows = list(csv.DictReader(open(f)))
        dst = os.path.join(out, kt.replace("kt_", "") + "_kernel_stats.csv")
        with open(dst, "w", newline="") as g:
            w = csv.writer(g)
            w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in rows:
                w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
        print("wrote", dst, len(rows), "kernels")

for prefix, dst in (("pmc_", "pmc_per_kernel.csv"), ("pmcwsi_", "pmc_per_kernel_wsi.csv")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sorted(glob.glob(os.path.join(out, prefix + "*"))):
        if not os.path.isdir(d):
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in acc.values() for c in k})
    with open(os.path.join(out, dst), "w", newline="") as g:
        w = csv.writer(g)
        # derived: mfma_util = MFMA-busy cycles / (1024 SIMDs x kernel cycles), kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs
        # (MI355X_MICROARCH.md: SQ_VALU_MFMA_BUSY_CYCLES = 32 per 32x32x16 MFMA, summed over the chip);
        # hbm_bytes = (2 x FETCH_SIZE + WRITE_SIZE) KB: gfx950 reports half of the bytes of wide streaming reads
        w.writerow(["Kernel", "dispatches"] + [c + "_avg" for c in counters] + ["mfma_util", "lds_conflict_ratio", "hbm_bytes_fetch_x2"])
        for k, v in sorted(acc.items()):
            if k.startswith("at::") or "elementwise" in k or "Memset" in k or "distribution" in k:
                continue
            n = max(len(x) for x in v.values())
            avg = {c: sum(v[c]) / len(v[c]) for c in v}
            util = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * avg["GRBM_GUI_ACTIVE"] / 8) if avg.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in avg else None
            ldsr = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"] if avg.get("SQ_LDS_IDX_ACTIVE") else None
            hbm = (2 * avg["FETCH_SIZE"] + avg["WRITE_SIZE"]) * 1024 if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg else None
            w.writerow([k, n] + [f"{avg[c]:.1f}" if c in avg else "" for c in counters] +
                       ["" if util is None else f"{util:.3f}", "" if ldsr is None else f"{ldsr:.3f}", "" if hbm is None else f"{hbm:.0f}"])
    print("wrote", os.path.join(out, dst))
