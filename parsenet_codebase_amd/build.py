"""Build the gfx950 shared library in-tree with hipcc.

One translation unit per kernel family under ``csrc/``; objects are compiled in
parallel and linked into ``libparsenet_hip.so`` next to this file so that the
library travels with the source tree (no JIT cache, no site-packages install).
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libparsenet_hip.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# PN_EXTRA_HIPCC_FLAGS: developer hook (e.g. "-DMS_TIMING" for the phase timers of meanshift.hip)
FLAGS = os.environ.get("PN_EXTRA_HIPCC_FLAGS", "").split() + [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    # the parity contract fixes the rounding of every distance: FMAs appear only
    # where the source writes fmaf explicitly
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(path):
    h = hashlib.sha1()
    for name in sorted(os.listdir(CSRC)):
        if name.endswith(".h"):
            with open(os.path.join(CSRC, name), "rb") as f:
                h.update(f.read())
    with open(path, "rb") as f:
        h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _compile(src):
    path = os.path.join(CSRC, src)
    obj = os.path.join(OBJ, src.replace(".hip", ".o"))
    stamp = obj + ".sha1"
    dig = _digest(path)
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    if force:
        for f in os.listdir(OBJ):
            os.remove(os.path.join(OBJ, f))
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        results = list(ex.map(_compile, srcs))
    objs = [o for o, _ in results]
    rebuilt = any(c for _, c in results)
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", LIB)
    elif verbose:
        print("up to date", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
