"""gcc recipe for the C oracle (test infrastructure).  Output: oracle/libmarch_oracle.so
(git-ignored; travels to the GPU box with the gpurun snapshot)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "march_oracle.c")
SRC4 = os.path.join(HERE, "march4d_oracle.c")
OUT = os.path.join(HERE, "libmarch_oracle.so")


def build(force=False):
    srcs = [s for s in (SRC, SRC4) if os.path.exists(s)]
    if not force and os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return OUT
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c11", "-fno-fast-math", "-ffp-contract=off",
           "-o", OUT] + srcs + ["-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
