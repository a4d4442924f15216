"""hipcc recipe for the gfx950 shared library behind the C ABI (include/contourist_hip.h).

Output: contourist_amd/lib/libcontourist_hip.so (git-ignored; travels with the gpurun snapshot).
hipcc cross-compiles for gfx950 without a GPU present."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libcontourist_hip.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "contourist_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, cmds = [], []
    headers = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "contourist_hip.h")]
    newest_header = max(os.path.getmtime(h) for h in headers)
    extra = os.environ.get("CX_EXTRA_FLAGS", "").split()
    for src in sources():
        obj = os.path.join(HERE, "lib", os.path.basename(src) + ".o")
        objs.append(obj)
        # objects built with extra flags are never reused (and leave a stamp so that a plain build replaces them)
        stamp = obj + ".flags"
        had_flags = os.path.exists(stamp)
        if (not force and not extra and not had_flags and not verbose and os.path.exists(obj)
                and os.path.getmtime(obj) > max(os.path.getmtime(src), newest_header)):
            continue
        cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function"] + extra
        if verbose:
            cmd.append("-Rpass-analysis=kernel-resource-usage")
        if extra:
            open(stamp, "w").write(" ".join(extra))
        elif had_flags:
            os.remove(stamp)
        cmds.append(cmd)
    if cmds:
        # translation units are independent: compile them side by side (8 cores here, 16 on a GPU box)
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 4)) as pool:
            list(pool.map(subprocess.check_call, cmds))
    subprocess.check_call([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force=True, verbose="-v" in sys.argv))
