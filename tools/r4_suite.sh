#!/bin/bash
# full GPU suite + bench line + A/B of stream-kernel variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout -k 10 1100 python3 -m pytest tests/ -q -m gpu > gpurun_out/r4/suite.txt 2>&1
tail -15 gpurun_out/r4/suite.txt | cut -c1-300
python3 bench.py --steps 20 --warmup 5 > gpurun_out/r4/bench1.json 2> gpurun_out/r4/bench1.err
tail -3 gpurun_out/r4/bench1.err
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/r4/bench1.json"))
r = d["roofline"]
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", r["frac"], "measured_peak", r.get("measured_peak_GBps"), "copy", r.get("measured_copy_GBps"), "frac_meas", r.get("frac_of_measured"), "frac_nec", r.get("frac_necessary"))
print("single", r["single_stream"]["ms_per_step"], [(k["name"], round(k["ms"], 4)) for k in r["single_stream"]["kernels"]])
print("levels", d.get("ms_all_levels"), "level1", d.get("level1_ms"), "api", d.get("api_ms"), d["config"].get("field_checksum"), d["config"]["grids_rotated"])
print(d["cpu_baseline"])
PY
