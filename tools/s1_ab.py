#!/usr/bin/env python3
"""A/B of stream-kernel builds (tools/variants.sh): each variant in its own process; stream kernel alone and the staged pipeline"""
import os, subprocess, sys
ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for rep in range(2):
    for name in sys.argv[1:]:
        env = dict(os.environ, CX_DEBUG="1", TAG=name)
        if name != "default":
            env["CX_LIB_PATH"] = os.path.join(ROOT, "contourist_amd", "lib", "variants", "lib_%s.so" % name)
        for tool, arg in (("stream_ab.py", []), ("time_modes.py", ["512", "staged"])):
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + arg, env=env, capture_output=True, text=True)
            lines = [l for l in out.stdout.strip().splitlines() if "stream" in l]
            print(name, "|", " || ".join(lines[-2:]), flush=True)
