"""Summarise a rocprofv3 rocpd database (ROCm 7: `rocprofv3 --kernel-trace [--pmc ...] -d DIR -o NAME`
writes NAME_results.db) into the per-kernel table committed under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof7/bench_results.db [--pmc] > profiles/r01_kernel_trace.txt
"""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size) "
                       "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace summary of %s" % path)
    print("# total kernel time %.3f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
    print("%-64s %6s %11s %11s %10s %10s %6s %5s %5s %5s %7s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us",
                                                              "pct", "vgpr", "agpr", "sgpr", "lds"))
    for r in rows:
        name = r[0].replace("(anonymous namespace)::", "")
        print("%-64s %6d %11.3f %11.1f %10.1f %10.1f %6.1f %5d %5d %5d %7d" % (
            name[:64], r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot, r[6] or 0, r[7] or 0,
            r[8] or 0, r[9] or 0))
    if "--pmc" in sys.argv:
        tabs = [t[0] for t in cur.execute("select name from sqlite_master where type in ('table','view')")]
        cand = [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()]
        print("# counter tables:", cand)
        for t in cand:
            cols = [c[1] for c in cur.execute("pragma table_info(%s)" % t)]
            print("#", t, cols)


if __name__ == "__main__":
    main()
