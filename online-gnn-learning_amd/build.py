"""Build recipe for libogl_hip.so (hipcc, gfx950 only, in-tree so the .so travels to the GPU box).

The library carries the hash of the sources it was built from (``ogl_source_hash()``, csrc/graph.hip, fed by
``-DOGL_SOURCE_HASH``): ``build()`` recomputes the hash of ``csrc/*`` + ``include/*`` + the compiler flags and REBUILDS when the
in-tree binary answers anything else (or cannot be loaded) — file times are not trusted, a prebuilt ``.so`` that travelled with
the tree proves nothing by existing.  Objects are compiled one hipcc process per source file, in parallel, and cached under
``build/`` by the hash of what went into them, so touching one kernel file recompiles one object."""
import concurrent.futures
import glob
import hashlib
import os
import subprocess
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libogl_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17"]


def sources():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(HERE, "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")))


def _digest(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    return h.hexdigest()


def source_hash():
    """Hash of everything that determines the library: every kernel source, every header, the flags."""
    return _digest(sources() + headers(), " ".join(FLAGS))[:16]


def built_hash(path=None):
    """What the in-tree library says it was built from (None: missing, unloadable, or older than the stamp)."""
    path = path or LIB
    if not os.path.exists(path):
        return None
    # read from the file's bytes, not through dlopen: a process that already has an older build of the same path loaded
    # would be handed that mapping again
    with open(path, "rb") as f:
        blob = f.read()
    at = blob.find(b"OGL_SOURCE_STAMP:")
    if at < 0:
        return None
    tail = blob[at + 17:at + 17 + 64].split(b"\0", 1)[0]
    return tail.decode("ascii", "replace")


def needs_build():
    return built_hash() != source_hash()


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


def _compile_one(src, stamp, verbose):
    """One object, cached by the hash of (this source, all headers, flags[, the stamp for the file that embeds it])."""
    embeds = os.path.basename(src) == "graph.hip"
    key = _digest([src] + headers(), " ".join(FLAGS) + (stamp if embeds else ""))[:16]
    obj = os.path.join(OBJ_DIR, "%s.%s.o" % (os.path.splitext(os.path.basename(src))[0], key))
    if os.path.exists(obj):
        return obj, False
    for old in glob.glob(os.path.join(OBJ_DIR, os.path.splitext(os.path.basename(src))[0] + ".*.o")):
        os.remove(old)
    cmd = ["hipcc"] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", os.path.join(HERE, "csrc"), "-c", src, "-o", obj + ".tmp"]
    if embeds:
        cmd.insert(1, '-DOGL_SOURCE_HASH="%s"' % stamp)
    if verbose:
        print(" ".join(cmd), flush=True)
    t0 = time.perf_counter()
    subprocess.run(cmd, check=True)
    os.replace(obj + ".tmp", obj)
    if verbose:
        print("compiled %s in %.1f s" % (os.path.basename(src), time.perf_counter() - t0), flush=True)
    return obj, True


def build(force=False, verbose=False, jobs=None):
    build_host(force, verbose)
    want = source_hash()
    have = built_hash()
    if not force and have == want:
        if verbose:
            print("libogl_hip.so is current: built from source hash %s" % want)
        return LIB
    if verbose:
        print("rebuilding libogl_hip.so: sources hash to %s, the in-tree library %s" % (
            want, "is missing or carries no stamp" if have is None else "was built from %s" % have))
    os.makedirs(OBJ_DIR, exist_ok=True)
    if force:
        for old in glob.glob(os.path.join(OBJ_DIR, "*.o")):
            os.remove(old)
    jobs = jobs or max(1, min(len(sources()), (os.cpu_count() or 2) - 1, 6))
    with concurrent.futures.ThreadPoolExecutor(jobs) as pool:
        objs = list(pool.map(lambda s: _compile_one(s, want, verbose), sources()))
    cmd = ["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared"] + [o for o, _ in objs] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    got = built_hash()
    if got != want:
        raise RuntimeError("libogl_hip.so was rebuilt but reports source hash %r, expected %r" % (got, want))
    return LIB


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
