"""Builds fasterrcnn_amd/csrc/libfrcnn_hip.so in-tree with hipcc for gfx950 (no GPU needed)."""
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def build(verbose=True, jobs=8, clean=False):
    """clean=True: every object is recompiled from the tracked sources (`make -B`), so the library cannot be a stale one that merely looks
    current to make; the default is make's incremental build."""
    cmd = ["make", "-C", CSRC, "-j%d" % jobs] + (["-B"] if clean else [])
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if verbose or res.returncode != 0:
        sys.stdout.write(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libfrcnn_hip.so failed (exit %d)" % res.returncode)
    return os.path.join(CSRC, "libfrcnn_hip.so")


if __name__ == "__main__":
    print(build())
