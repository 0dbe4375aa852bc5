"""Builds libpda_pointnet2.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libpda_pointnet2.so")


def build(force=False, verbose=False):
    cmd = ["make", "-C", CSRC, "-j8"]
    if force:
        cmd.append("-B")
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, stdout=out)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce %s" % LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True))
