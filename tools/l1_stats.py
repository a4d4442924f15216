"""per-pass kernel table of the Level-1 post-pass from a rocprofv3 --kernel-trace --stats csv (tools/bench_level1.py runs 3 passes)
usage: python tools/l1_stats.py <kernel_stats.csv> [passes]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
passes = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
tot = 0.0
for r in rows:
    if "cxp_" in r["Name"]:
        per = float(r["TotalDurationNs"]) / passes / 1e6
        tot += per
        print("%-28s calls/pass %4.1f  ms/pass %.3f" % (r["Name"].split("(")[0][:28], int(r["Calls"]) / passes, per))
print("sum of post-pass kernels per pass: %.3f ms" % tot)
