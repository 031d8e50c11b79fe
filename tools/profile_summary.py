"""Export the per-kernel summary (rocprofv3 --kernel-trace --stats; the rocpd database's top_kernels view)
as a small CSV for profiles/.

    python tools/profile_summary.py <results.db> <out.csv>
"""
import csv
import re
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(sys.argv[2], "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for name, calls, total, avg, pct in rows:
            w.writerow([name, calls, "%.3f" % total, "%.3f" % avg, "%.3f" % pct])
    fam = {}
    for name, calls, total, avg, pct in rows:
        m = re.search(r"::(\w+)", name)
        key = m.group(1) if m else name.split("(")[0]
        c, t = fam.get(key, (0, 0.0))
        fam[key] = (c + calls, t + total)
    for key, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:16]:
        print("%-28s calls %6d  total %10.1f us  avg %8.2f us" % (key, c, t, t / c))


if __name__ == "__main__":
    main()
