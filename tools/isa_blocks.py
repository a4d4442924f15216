#!/usr/bin/env python3
"""Static census of a kernel's gfx950 ISA per basic block (tools only).
usage: tools/isa_blocks.py <kernel-name-substring> [-v]   (compiles cx_march3d.hip with -save-temps into /tmp/cx_isa)"""
import os, re, subprocess, sys
src = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "contourist_amd", "csrc", sys.argv[3] if len(sys.argv) > 3 else "cx_march3d.hip")
os.makedirs("/tmp/cx_isa", exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/tmp/cx_isa/m.o",
                       "-save-temps", "-Wno-unused-function"] + os.environ.get("CX_EXTRA_FLAGS", "").split(), cwd="/tmp/cx_isa", stderr=subprocess.DEVNULL)
s = open("/tmp/cx_isa/%s-hip-amdgcn-amd-amdhsa-gfx950.s" % os.path.basename(src).replace(".hip", "")).read()
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)\.Lfunc_end', s, re.S | re.M):
    if sys.argv[1] not in m.group(1):
        continue
    print(m.group(1))
    cur, blocks = ["entry", []], []
    for l in m.group(2).split("\n"):
        b = re.match(r'^(\.LBB\d+_\d+):', l)
        if b:
            blocks.append(cur); cur = [b.group(1), []]
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur[1].append(l.strip())
    blocks.append(cur)
    for name, ins in blocks:
        c = lambda p: sum(1 for x in ins if x.startswith(p))
        br = [x.split()[-1] for x in ins if x.startswith(("s_cbranch", "s_branch"))]
        print("%-12s v %4d s %4d ds %3d vmem %3d  %s" % (name, c("v_"), c("s_"), c("ds_"), c("global_") + c("buffer_") + c("scratch_"), " ".join(br)))
        if "-v" in sys.argv:
            for x in ins:
                print("      " + x)
    break
