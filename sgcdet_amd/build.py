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
# Packed-FP32 VALU instructions are switched off for every kernel of this library.  Measured on MI355X
# (tools/hazard/pk_mfma_repro.hip, a 150-line reproducer): `v_pk_fma_f32` with an op_sel operand swizzle returns
# wrong values in lanes 48-63 while waves of the bf16 implicit-GEMM kernel (v_mfma_f32_32x32x16_bf16) run on the
# same SIMD from another stream.  The compiler emits exactly that instruction for `x * W - 0.5` pairs in the
# gather kernels, which corrupted ~0.3 % of the gathered rows whenever two scenes were in flight.  Without
# the packed forms the kernels are bit-identical to their serial results under any overlap.
NO_PACKED_FP32 = ("-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops")
# per-file extra flags (experiments: SGC_FLAGS_<stem>="..." in the environment)
FILE_FLAGS = {}
for _k, _v in os.environ.items():
    if _k.startswith("SGC_FLAGS_"):
        FILE_FLAGS[_k[len("SGC_FLAGS_"):] + ".hip"] = _v.split()


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
               "-Wall", "-Wno-unused-function", *NO_PACKED_FP32, *extra_flags,
               *FILE_FLAGS.get(os.path.basename(src), ())]
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
