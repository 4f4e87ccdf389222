"""Aggregate rocprofv3 counter_collection CSVs (one counter per run) per kernel name."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, "prof_" + counter, "**", "*counter_collection.csv"), recursive=True)
    agg = defaultdict(lambda: [0.0, 0])
    for f in files:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "?").split("(")[0][:90]
                agg[name][0] += float(row.get("Counter_Value", 0) or 0)
                agg[name][1] += 1
    print("== %s (rocprofv3 units: KB per dispatch; gfx950 FETCH_SIZE under-reports wide streaming reads 2x)" % counter)
    for name, (tot, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:14]:
        print("%-92s dispatches %6d  total %14.1f  per-dispatch %12.2f" % (name, n, tot, tot / max(n, 1)))
