"""Aggregate rocprofv3 counter_collection CSVs (one counter per run) per kernel name, and the
kernel-trace stats CSVs, from the directory tree scripts/gpu_pmc.sh writes."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for f in sorted(glob.glob(os.path.join(root, "**", "*kernel_stats.csv"), recursive=True)):
    print("== " + f)
    with open(f) as fh:
        for i, row in enumerate(csv.reader(fh)):
            if i < 12:
                print("   " + " | ".join(x[:70] for x in row))
for d in sorted(glob.glob(os.path.join(root, "*_*"))):
    base = os.path.basename(d)
    counter = base.rsplit("_", 1)[0]
    if counter not in ("FETCH_SIZE", "WRITE_SIZE"):
        continue
    agg = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row.get("Kernel_Name", "?").replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
                agg[name][0] += float(row.get("Counter_Value", 0) or 0)
                agg[name][1] += 1
    print("== %s  (sum over dispatches / per dispatch; rocprofv3 reports KB; on gfx950 FETCH_SIZE counts wide "
          "streaming reads at half their bytes -- MI355X_MICROARCH.md HBM section)" % base)
    for name, (tot, n) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:12]:
        print("   %-82s dispatches %6d  total %16.1f  per-dispatch %14.2f" % (name, n, tot, tot / max(n, 1)))
