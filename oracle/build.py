"""Builds the oracle's C restatement (oracle/libdemon_oracle.so) with gcc.  Test infrastructure."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "demon_oracle.c")
OUT = os.path.join(HERE, "libdemon_oracle.so")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=c99", "-o", OUT, SRC, "-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force=True))
