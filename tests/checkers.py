"""Loaders of the CHECKERS (test infrastructure): the CPU restatement oracle/libsf3d_oracle.so and the wrapped, unmodified
reference oracle/_ref/libsf3d_ref*.so.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use this module;
the product package (criteria3d_amd/) holds no reference to anything under oracle/."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from criteria3d_amd.capi import SF3D

ROOT = Path(__file__).resolve().parent.parent
ORACLE_LIB = ROOT / "oracle" / "libsf3d_oracle.so"
REFERENCE_LIB = ROOT / "oracle" / "_ref" / "libsf3d_ref.so"
QT_CORE = Path(os.environ.get("SF3D_QT_CORE", "/opt/conda/lib/libQt5Core.so.5"))


def load_oracle() -> SF3D:
    """CPU restatement of the reference algorithm (oracle/sf3d_oracle.cpp)."""
    return SF3D(ORACLE_LIB)


ORACLE_FM_LIB = ROOT / "oracle" / "libsf3d_oracle_fm.so"


def load_oracle_fastmath() -> SF3D:
    """The fast-math TWIN of the oracle (oracle/Makefile `oracle-fm`): the same restatement compiled with the PRODUCT's elementary
    functions (criteria3d_amd/csrc/sf3d_fastmath.inc) instead of the C library's.  NOT a checker of results and never the pin: it
    tells arithmetic sensitivity of a scenario from a kernel effect (tests/test_gpu_sensitivity.py) and nothing else."""
    if not ORACLE_FM_LIB.exists():
        import subprocess
        subprocess.run(["make", "-C", str(ROOT / "oracle"), "oracle-fm"], check=True, stdout=subprocess.DEVNULL)
    return SF3D(ORACLE_FM_LIB)


def load_oracle_copy(tag: str) -> SF3D:
    """A second, independent instance of the oracle in this process (the library keeps its model in globals): the file is copied
    under another name, so the loader maps it again with globals of its own.  For tests that let two oracle runs go side by side."""
    import shutil
    import tempfile
    d = Path(tempfile.mkdtemp(prefix="sf3d_oracle_"))
    f = d / f"libsf3d_oracle_{tag}.so"
    shutil.copy(ORACLE_LIB, f)
    return SF3D(f)


def _reference(name: str) -> SF3D:
    if QT_CORE.exists():
        C.CDLL(str(QT_CORE), mode=C.RTLD_GLOBAL)   # linked by soname, deliberately not on the rpath
    return SF3D(REFERENCE_LIB.with_name(name))


def load_reference() -> SF3D:
    """The wrapped, unmodified reference (built by oracle/Makefile `ref`)."""
    return _reference("libsf3d_ref.so")


def load_reference_ndebug() -> SF3D:
    """The same unmodified sources built with -DNDEBUG (oracle/Makefile `ref-ndebug`): golden vectors with Urban / Road nodes."""
    return _reference("libsf3d_ref_ndebug.so")


def load_reference_tuned() -> SF3D:
    """The same unmodified sources built -O3 -march=x86-64-v3 (oracle/Makefile `ref-tuned`): CPU baseline timing only."""
    return _reference("libsf3d_ref_tuned.so")
