"""Compile the C restatement (oracle/geometry_ref.c) into oracle/liboracle.so with gcc."""
import os, subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "geometry_ref.c")
LIB = os.path.join(HERE, "liboracle.so")


def build(force: bool = False) -> str:
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC", "-o", LIB, SRC, "-lm"]
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
