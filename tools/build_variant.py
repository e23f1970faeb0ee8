"""A DIAGNOSTIC build of the kernel library beside the product one: the same sources with extra -D switches, linked to
tools/debug/libogl_hip_<name>.so (git-ignored, travels to the GPU box).  Nothing in the package loads it; a probe script points
`_lib.LIB_PATH` at it before the first call.  Usage: python tools/build_variant.py phase -DOGL_X3_PHASE_STAMPS"""
import concurrent.futures
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(name, defines):
    spec = importlib.util.spec_from_file_location("_ogl_build", os.path.join(ROOT, "online-gnn-learning_amd", "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    out_dir = os.path.join(ROOT, "tools", "debug")
    obj_dir = os.path.join(out_dir, "obj_" + name)
    os.makedirs(obj_dir, exist_ok=True)
    stamp = b.source_hash()

    def one(src):
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        cmd = ["hipcc"] + b.FLAGS + list(defines) + ['-DOGL_SOURCE_HASH="%s"' % stamp, "-I", os.path.join(ROOT, "include"),
                                                     "-I", os.path.join(b.HERE, "csrc"), "-c", src, "-o", obj]
        subprocess.run(cmd, check=True)
        return obj

    with concurrent.futures.ThreadPoolExecutor(6) as pool:
        objs = list(pool.map(one, b.sources()))
    lib = os.path.join(out_dir, "libogl_hip_%s.so" % name)
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", lib], check=True)
    return lib


if __name__ == "__main__":
    print(build(sys.argv[1], sys.argv[2:]))
