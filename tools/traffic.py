#!/usr/bin/env python3
"""Turn the rocprofv3 PMC passes of tools/profile_round.sh into profiles/traffic.json: HBM-side bytes per launch of the Level-0
kernels.

Round 4: the bytes come from the L2's fabric REQUEST counters by size -- TCC_EA0_RDREQ_32B / _64B / _128B for reads,
TCC_EA0_WRREQ_64B and TCC_EA0_WRREQ (the rest are 32-byte writes) for writes -- each request counted at its own size.  Rounds 1-3
derived them from FETCH_SIZE / WRITE_SIZE: on gfx950 FETCH_SIZE tallies a 128-byte request at 64 bytes (MI355X_MICROARCH.md), so
only the stream kernel's reads were doubled there and the GATHERS of the emit stages -- which are 128-byte requests too (tools/micro/
sector_fetch.hip: every load flavour fetches whole 128-byte lines) -- were under-counted by half: 1.40 GB per extraction was
reported where 1.66 GB moved.  FETCH_SIZE / WRITE_SIZE are kept in the detail file for comparison."""
import collections, csv, glob, json, os, sys
root = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
dest = sys.argv[2] if len(sys.argv) > 2 else "profiles"
vals = collections.defaultdict(dict)
for name in ("fetch", "write", "ea_rd", "ea_wr"):
    files = sorted(glob.glob(os.path.join(root, name, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:          # the latest pass only
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            acc[(row["Kernel_Name"], row["Counter_Name"])].append(float(row["Counter_Value"]))
        for (k, c), v in acc.items():
            kk = k.split("(")[0].replace("void ", "").split("<")[0].strip()
            if kk.startswith("cx_k_"):
                vals[kk][c] = sum(v[1:]) / max(len(v) - 1, 1)      # mean of launches 2..N
out = {}
for kk, d in vals.items():
    rd = 32.0 * d.get("TCC_EA0_RDREQ_32B_sum", 0.0) + 64.0 * d.get("TCC_EA0_RDREQ_64B_sum", 0.0) + 128.0 * d.get("TCC_EA0_RDREQ_128B_sum", 0.0)
    w64 = d.get("TCC_EA0_WRREQ_64B_sum", 0.0)
    wr = 64.0 * w64 + 32.0 * max(d.get("TCC_EA0_WRREQ_sum", 0.0) - w64, 0.0)
    out[kk + "_512"] = {
        "read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr,
        "requests": {c: d[c] for c in sorted(d) if c.startswith("TCC_EA0")},
        "FETCH_SIZE_x1024": d.get("FETCH_SIZE", 0.0) * 1024.0, "WRITE_SIZE_x1024": d.get("WRITE_SIZE", 0.0) * 1024.0,
        "note": "bytes = fabric requests of the L2 (TCC_EA0_*), each at its size; mean of launches 2..N of tools/prof_step.py 512",
    }
json.dump(out, open(os.path.join(dest, "traffic_detail.json"), "w"), indent=1)
level0 = ["cx_k_stream", "cx_k_scan_list", "cx_k_emit_vertices", "cx_k_emit_triangles_q"]
summary = {k: v["hbm_bytes"] for k, v in out.items()}
summary["level0_512"] = sum(summary.get(k + "_512", 0.0) for k in level0)
summary["level0_512_reads"] = sum(out[k + "_512"]["read_bytes"] for k in level0 if k + "_512" in out)
summary["level0_512_writes"] = sum(out[k + "_512"]["write_bytes"] for k in level0 if k + "_512" in out)
json.dump(summary, open(os.path.join(dest, "traffic.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
