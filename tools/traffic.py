#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of tools/pmc.sh into profiles/traffic.json (HBM bytes per launch of the
Level-0 kernels, corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE are in KiB-like
units of 1024 B; on gfx950 FETCH_SIZE counts wide coalesced streaming reads at half their bytes)."""
import collections, csv, glob, json, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
dest = sys.argv[2] if len(sys.argv) > 2 else "profiles"
out = {}
vals = collections.defaultdict(dict)
for name in ("fetch", "write"):
    files = sorted(glob.glob(os.path.join(root, name, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:          # the latest pass only
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            kk = k.split("(")[0].replace("void ", "").split("<")[0].strip()
            if kk.startswith("cx_k_"):
                vals[kk][c] = sum(v[1:]) / max(len(v) - 1, 1)
for kk, d in vals.items():
    fetch_raw = d.get("FETCH_SIZE", 0.0) * 1024.0
    write = d.get("WRITE_SIZE", 0.0) * 1024.0
    out[kk + "_512"] = {
        "fetch_bytes_raw": fetch_raw, "write_bytes": write,
        # 16-B-per-lane streaming loads (grid rows, cell records) are counted at 1/2; narrower gathers are
        # uncalibrated, so both bounds are given
        "hbm_bytes_low": fetch_raw + write, "hbm_bytes_high": 2.0 * fetch_raw + write,
        "note": "FETCH_SIZE*1024 (+ x2 upper bound for 16-B/lane streams on gfx950) + WRITE_SIZE*1024, mean of launches 2..N",
    }
json.dump(out, open(os.path.join(dest, "traffic_detail.json"), "w"), indent=1)
# the number bench.py reports as roofline.traffic: HBM bytes of ONE extraction = all Level-0 kernels of the staged pipeline.
# The stream kernel's loads are 16 B per lane (FETCH_SIZE counts them at half: x2, MI355X_MICROARCH.md); the emit kernels
# gather 4-16 B per lane (uncalibrated: counted as reported, a lower bound) -- both sums are kept
level0 = ["cx_k_stream", "cx_k_scan_list", "cx_k_emit_vertices", "cx_k_emit_triangles_q"]
summary = {k: v["hbm_bytes_high"] if k.startswith("cx_k_stream") else v["hbm_bytes_low"] for k, v in out.items()}
summary["level0_512"] = sum(summary.get(k + "_512", 0.0) for k in level0)
summary["level0_512_upper"] = sum(out[k + "_512"]["hbm_bytes_high"] for k in level0 if k + "_512" in out)
json.dump(summary, open(os.path.join(dest, "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
