"""Builds the gfx950 shared library in-tree (``sgcdet_amd/csrc/libsgcdet_amd.so``).

hipcc cross-compiles without a GPU, so this runs in the CPU-only build container; the
resulting ``.so`` travels to the GPU box with the repo snapshot.  No torch headers are
involved: the library is plain HIP behind the C ABI of ``include/sgcdet_amd.h``.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsgcdet_amd.so")
ARCH = "gfx950"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        [os.path.join(HERE, "..", "include", "sgcdet_amd.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for src in sources():
        obj = src[:-4] + ".o"
        cmd = [hipcc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj,
               "-Wall", "-Wno-unused-function", *extra_flags]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, pr in procs:
        out, _ = pr.communicate()
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode(), file=sys.stderr)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-o", LIB]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
