"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Importers allowed: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  The product package ``sgcdet_amd`` never imports this.

``oracle.ops()`` returns a ``TensorOps`` front end (CPU tensors) over
``libsgc_oracle.so`` (scalar checker) or ``libsgc_oracle_omp.so`` (OpenMP build used
for the CPU baseline timing).  Parity status: see the header of ``sgc_oracle.c``
("parity unpinned" by the reference's own tests; pinned by the grid_sample identity
and by container-generated golden vectors).
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CACHE = {}


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    targets = [os.path.join(_HERE, n) for n in ("libsgc_oracle.so", "libsgc_oracle_omp.so")]
    src = os.path.join(_HERE, "sgc_oracle.c")
    stale = force or any((not os.path.exists(t)) or os.path.getmtime(t) < os.path.getmtime(src) for t in targets)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
    return targets


def library(omp=False):
    key = "omp" if omp else "scalar"
    if key not in _CACHE:
        from sgcdet_amd._abi import Library
        path = os.path.join(_HERE, "libsgc_oracle_omp.so" if omp else "libsgc_oracle.so")
        if not os.path.exists(path):
            build()
        _CACHE[key] = Library(path)
    return _CACHE[key]


def ops(omp=False):
    from sgcdet_amd.tensor_api import TensorOps
    return TensorOps(library(omp), "cpu")
