"""Per-kernel statistics from a rocprofv3 (ROCm 7.2) rocpd SQLite result file -- the same table
`rocprofv3 --stats` prints (calls, total / average / min / max duration, share of GPU time)."""
import sqlite3
import sys


def main(db_path, out=sys.stdout):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [c[1] for c in cur.execute("pragma table_info('kernels')")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute("select %s, start, end from kernels" % name_col).fetchall()
    agg = {}
    for name, s, e in rows:
        name = name.replace("(anonymous namespace)::", "").replace("void ", "")
        name = name.split("(")[0]
        d = (e - s) / 1e3  # ns -> us
        a = agg.setdefault(name, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    total = sum(a[1] for a in agg.values())
    out.write("%-100s %8s %14s %12s %12s %12s %7s\n" % ("Name", "Calls", "TotalDur(us)", "Avg(us)", "Min(us)", "Max(us)", "Pct"))
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        out.write("%-100s %8d %14.1f %12.2f %12.2f %12.2f %6.2f%%\n" % (name[:100], a[0], a[1], a[1] / a[0], a[2], a[3],
                                                                       100.0 * a[1] / total))


if __name__ == "__main__":
    main(sys.argv[1])
