"""TEST INFRASTRUCTURE ONLY — compile the C oracle into oracle/_cbuild/libpn_oracle.so."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "pn_oracle.c")
OUT_DIR = os.path.join(HERE, "_cbuild")
LIB = os.path.join(OUT_DIR, "libpn_oracle.so")


def build(verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ["gcc", "-O2", "-fopenmp", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
           "-std=c11", SRC, "-o", LIB, "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("oracle build failed:\n" + r.stdout + r.stderr)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build()
