"""In-tree build of libpcr_hip.so with hipcc for gfx950 (no JIT cache: the .so travels with the repo)."""

import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(jobs=8, verbose=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), f"-j{jobs}"]
    if not verbose:
        cmd.append("-s")
    subprocess.run(cmd, check=True)
    return os.path.join(_HERE, "libpcr_hip.so")


if __name__ == "__main__":
    print(build(verbose=True))
