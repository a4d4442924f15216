#!/usr/bin/env python3
"""A/B timing of 3-D Level-0 builds (tools/variants.sh): python tools/ab3d.py default v1 ...; per-kernel medians (HIP events)."""
import os, subprocess, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for rep in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, CX_DEBUG="1", TAG="%-8s" % name)
        if name != "default":
            env["CX_LIB_PATH"] = os.path.join(ROOT, "contourist_amd", "lib", "variants", "lib_%s.so" % name)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "quick_time.py"), "512"], env=env, capture_output=True, text=True)
        print("\n".join(l for l in out.stdout.splitlines() if "full" in l), flush=True)
