"""gcc recipe for the C restatement (test infrastructure).  ``python -m oracle.build``."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "rt_oracle.c")
LIB = os.path.join(HERE, "liboracle_rt.so")


def build(force: bool = False) -> str:
    if (not force and os.path.exists(LIB)
            and os.path.getmtime(LIB) >= os.path.getmtime(SRC)):
        return LIB
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-shared", "-fPIC", "-Wall",
           SRC, "-o", LIB, "-lm"]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
