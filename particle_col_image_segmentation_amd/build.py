"""Build libpcseg.so (hipcc, gfx950 only) in-tree next to the sources."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["stencil.hip", "ccl.hip", "reduce.hip", "edt.hip", "watershed.hip", "tables.hip", "frontend.hip"]
LIB = os.path.join(HERE, "libpcseg.so")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + ["common.h", "tile_ops.h"]]
    deps.append(os.path.join(HERE, "..", "include", "pcseg.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, out_dir=None, extra_flags=None):
    """``out_dir`` / ``extra_flags``: an A/B variant of the library (``-D`` tunables) built beside the product's, e.g. into
    ``ab/<name>/libpcseg.so`` -- loaded with ``PCSEG_LIB=<path>`` (``_lib.load``), see profiles/r04/make_ab.sh."""
    lib_path = LIB if out_dir is None else os.path.join(out_dir, "libpcseg.so")
    if out_dir is None and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    obj_dir = os.path.join(HERE, "build") if out_dir is None else os.path.join(out_dir, "build")
    os.makedirs(obj_dir, exist_ok=True)
    flags = (extra_flags if extra_flags is not None else os.environ.get("PCSEG_EXTRA_FLAGS", "")).split()
    for src in SOURCES:
        obj = os.path.join(obj_dir, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + flags + [
            "-c", os.path.join(CSRC, src), "-o", obj]  # PCSEG_EXTRA_FLAGS: -D tunables for A/B builds (profiles/ab_compare.sh)
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
