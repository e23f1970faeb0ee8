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


HOST_LIB = os.path.join(HERE, "libogl_host.so")
HOST_SRC = os.path.join(HERE, "csrc_host", "host_replay.c")


def build_host(force=False, verbose=False):
    """The optional host helper (replay-buffer arithmetic as a C loop; pure-Python fallback when it is absent)."""
    if not force and os.path.exists(HOST_LIB) and os.path.getmtime(HOST_LIB) >= os.path.getmtime(HOST_SRC):
        return HOST_LIB
    cmd = ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", HOST_SRC, "-lm", "-o", HOST_LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return HOST_LIB


def build(force=False, verbose=False):
    build_host(force, verbose)
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
