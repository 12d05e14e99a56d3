"""Summarise rocprofv3 --pmc CSV output (one directory, several <NAME>_counter_collection.csv files,
one counter set per pass) into the per-kernel table committed under profiles/.

    python tools/pmc_summary.py gpurun_out/pmc8 > profiles/r01_pmc_kernels.txt
"""
import collections
import csv
import glob
import os
import sys

KEEP = ("mlp_fwd_kernel", "mlp_bwd_kernel", "wgrad256_kernel", "wgrad256_split_kernel", "layer_split_kernel", "wgrad_tiles_kernel", "wgrad_reduce_multi_kernel")


def main():
    d = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in sorted(glob.glob(os.path.join(d, "*_counter_collection.csv"))):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print("# rocprofv3 --pmc summary of %s (values are per-dispatch means; counters from separate passes)" % d)
    print("# gfx950 notes (MI355X_MICROARCH.md): SQ_* cycle counters are quad-cycles summed over waves; "
          "FETCH_SIZE / WRITE_SIZE are KB; FETCH_SIZE under-reports wide coalesced reads by 2x;")
    print("# GRBM_GUI_ACTIVE is summed over the 8 XCDs: cycles per XCD = value / 8.")
    for k in sorted(agg):
        if not os.environ.get("KEEP_ALL") and not any(x in k for x in KEEP):
            continue
        du = dur[k]
        print("\n== %s   dispatches seen: %d   duration us: mean %.1f min %.1f max %.1f" % (
            k, len(du), sum(du) / len(du), min(du), max(du)))
        for c, v in sorted(agg[k].items()):
            print("   %-28s mean %.6g   min %.6g   max %.6g   (n=%d)" % (c, sum(v) / len(v), min(v), max(v), len(v)))
        a = agg[k]
        if "GRBM_GUI_ACTIVE" in a and "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            gui = sum(a["GRBM_GUI_ACTIVE"]) / len(a["GRBM_GUI_ACTIVE"]) / 8.0
            mf = sum(a["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(a["SQ_VALU_MFMA_BUSY_CYCLES"])
            print("   -> MFMA pipe utilisation = MFMA_BUSY / (cycles_per_XCD * 1024 SIMDs) = %.1f %%" % (100 * mf / (gui * 1024)))
            d_us = [x for x in du]
            print("   -> shader clock ~ cycles_per_XCD / duration = %.2f GHz" % (gui / (sum(d_us) / len(d_us)) / 1e3))


if __name__ == "__main__":
    main()
