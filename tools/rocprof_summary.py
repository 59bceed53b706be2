#!/usr/bin/env python3
"""Turn a rocprofv3 (rocpd sqlite) kernel trace into the text summaries kept under profiles/.
    python tools/rocprof_summary.py <results.db> <out_prefix> [--dispatches N]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    out = sys.argv[2]
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out + "_kernel_stats.csv", "w") as f:
        f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
        for n, c, t, a, p in rows:
            f.write('"%s",%d,%.0f,%.1f,%.3f\n' % (n, c, t * 1e3, a * 1e3, p))
    if "--dispatches" in sys.argv:
        k = int(sys.argv[sys.argv.index("--dispatches") + 1])
        q = "select name, grid_x, grid_y, workgroup_x, duration, vgpr_count, lds_size from kernels order by start desc limit %d" % k
        rows = list(cur.execute(q))[::-1]
        with open(out + "_last_dispatches.csv", "w") as f:
            f.write("Name,GridX,GridY,WG,DurationNs,VGPR,LDS\n")
            for r in rows:
                f.write('"%s",%d,%d,%d,%d,%d,%d\n' % (r[0][:60], r[1], r[2], r[3], r[4], r[5], r[6]))
    print("wrote", out + "_kernel_stats.csv")


if __name__ == "__main__":
    main()
