"""Build recipe for libogl_hip.so (hipcc, gfx950 only, in-tree so the .so travels to the GPU box)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libogl_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    return any(os.path.getmtime(p) > t for p in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = ["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
           "-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc")] + sources() + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
