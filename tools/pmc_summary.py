"""Summarise rocprofv3 --pmc CSV output (one directory, several <NAME>_counter_collection.csv files,
one counter set per pass) into the per-kernel table committed under profiles/.

    python tools/pmc_summary.py gpurun_out/pmc8 > profiles/r01_pmc_kernels.txt
"""
import collections
import csv
import glob
import os
import sys

KEEP = ("mlp_fwd_kernel", "mlp_bwd_kernel", "mlp_fwd_h3_kernel", "mlp_bwd_h3_kernel", "wgrad256_kernel",
        "wgrad256_half_kernel", "wgrad_half_narrow_kernel", "rows4_kernel", "copyBuffer",
        "wgrad_tiles_kernel", "wgrad_reduce_multi_kernel", "wgrad_kernel", "vecmat_kernel", "elementwise_kernel")

# bench.py's region names of the kernels whose HBM traffic goes into profiles/pmc_traffic_r<NN>.json (P = 786432)
REGIONS = {"mlp_fwd_h3_kernel<3, true, 0>": "mlp_fwd_h3_kernel/P=786432/train",
           "mlp_fwd_h3_kernel<3, false, 0>": "mlp_fwd_h3_kernel/P=786432/infer",
           "mlp_fwd_h3_kernel<3, true, 2>": "mlp_fwd_h3_kernel<fine stage>/P=786432/train",
           "mlp_fwd_h3_kernel<3, false, 2>": "mlp_fwd_h3_kernel<fine stage>/P=786432/infer",
           "mlp_bwd_h3_kernel<3>": "mlp_bwd_h3_kernel/P=786432",
           "wgrad256_half_kernel<0>": "wgrad256_kernel<8 GEMMs, half>/P=786432"}


def step_traffic(d, steps):
    """whole-step HBM bytes: every dispatch of `tools/profile_steps.py pmc` (PMC_STEPS plain steps), both counters"""
    tot = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0}
    for c in tot:
        f = os.path.join(d, "%s_counter_collection.csv" % c)
        if not os.path.isfile(f):
            return None
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                tot[c] += float(r["Counter_Value"])
    return int((tot["WRITE_SIZE"] * 1024 + 2 * tot["FETCH_SIZE"] * 1024) / steps)


def traffic_json(agg, out_path, source, bytes_per_step=None, durations=None):
    """bytes per launch = WRITE_SIZE KB x 1024 + 2 x FETCH_SIZE KB x 1024 (gfx950 tallies wide coalesced reads at half
    their size: MI355X_MICROARCH.md, HBM section), with the calibration copy of known size beside it"""
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench

    def mean(k, c):
        v = agg[k].get(c)
        return sum(v) / len(v) if v else None
    per, by_region = {}, {}
    for k in agg:
        f, w = mean(k, "FETCH_SIZE"), mean(k, "WRITE_SIZE")
        if f is None or w is None:
            continue
        short = k.replace("void ", "").replace("scn::wg256h::", "").replace("scn::wgnh::", "")
        per[short] = {"fetch_kb_raw": f, "write_kb": w, "bytes": int(w * 1024 + 2 * f * 1024)}
        for pat, region in REGIONS.items():
            if pat in short:
                by_region[region] = per[short]["bytes"]
    # the calibration copy: the LARGEST dispatch of the runtime's blit kernel (small copies share its name)
    cal = [{"fetch_kb_raw": max(agg[k]["FETCH_SIZE"]), "write_kb": max(agg[k]["WRITE_SIZE"])} for k in agg
           if "copyBuffer" in k and agg[k].get("FETCH_SIZE") and agg[k].get("WRITE_SIZE")]
    # matrix-pipe utilisation and shader clock per region: MFMA busy cycles / (cycles per XCD x 1024 SIMDs); cycles per XCD / duration
    busy, clock = {}, {}
    for k in agg:
        a = agg[k]
        if a.get("GRBM_GUI_ACTIVE") and a.get("SQ_VALU_MFMA_BUSY_CYCLES") and durations and durations.get(k):
            gui = mean(k, "GRBM_GUI_ACTIVE") / 8.0
            short = k.replace("void ", "").replace("scn::wg256h::", "").replace("scn::wgnh::", "")
            for pat, region in REGIONS.items():
                if pat in short:
                    busy[region] = mean(k, "SQ_VALU_MFMA_BUSY_CYCLES") / (gui * 1024.0)
                    clock[region] = gui / (sum(durations[k]) / len(durations[k])) / 1e3
    rec = {"_csrc_sha16": bench.csrc_sha16(), "_source": source, "bytes_per_launch": by_region, "per_kernel": per,
           "mfma_busy": busy, "clock_ghz": clock}
    if bytes_per_step:
        rec["bytes_per_step"] = bytes_per_step
        rec["_bytes_per_step_source"] = ("the same two counters summed over every dispatch of `tools/profile_steps.py pmc` "
                                         "(plain default steps, 4096 rays) / the number of steps")
    if cal:
        c = max(cal, key=lambda v: v["write_kb"])
        rec["calibration_copy_1GiB_read_1GiB_written"] = {
            "fetch_kb_raw": c["fetch_kb_raw"], "write_kb": c["write_kb"],
            "read_bytes_counted_x2_over_known": 2 * c["fetch_kb_raw"] * 1024 / float(1 << 30),
            "write_bytes_counted_over_known": c["write_kb"] * 1024 / float(1 << 30)}
    with open(out_path, "w") as fh:
        json.dump(rec, fh, indent=1, sort_keys=True)


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
    if "--json" in sys.argv:
        bps = None
        if "--steps-dir" in sys.argv:
            bps = step_traffic(sys.argv[sys.argv.index("--steps-dir") + 1], int(sys.argv[sys.argv.index("--steps") + 1]))
        traffic_json(agg, sys.argv[sys.argv.index("--json") + 1],
                     "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, one counter each, --kernel-trace only) of "
                     "tools/profile_kernels.py at P = 786432 (tools/collect_profiles.sh)", bps, dur)
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
