"""Build recipes: the HIP product library (gfx950), the C++ drop-in shim, and the test oracle.

Everything is built IN-TREE so the shared objects travel with the repository snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / "criteria3d_amd" / "csrc"
INCLUDE = ROOT / "include"
PRODUCT_LIB = CSRC / "libsf3d_hip.so"
SHIM_LIB = ROOT / "shim" / "libsoilFluxes3D_mi355x.so"

HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"

# -ffp-contract=off: products and sums must round separately, as in the reference's x86-64 -O2
# build (no FMA contraction), so trajectories stay within the 1e-6 parity band.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
             "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _run(cmd, **kw):
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, **kw)
    if proc.returncode != 0:
        raise RuntimeError(f"command failed ({proc.returncode}): {' '.join(map(str, cmd))}\n{proc.stdout}")
    return proc.stdout


def _stale(target: Path, sources) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(s).stat().st_mtime > t for s in sources)


def build_product(force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950: kernels + C ABI -> criteria3d_amd/csrc/libsf3d_hip.so"""
    srcs = [CSRC / "sf3d_solver.hip", CSRC / "sf3d_api.cpp"]
    deps = srcs + sorted(CSRC.glob("*.inc")) + sorted(CSRC.glob("*.h")) + [INCLUDE / "sf3d.h"]      # every part of the translation unit (the Makefile rule uses the same wildcard)
    if force or _stale(PRODUCT_LIB, deps):
        extra = os.environ.get("SF3D_EXTRA_HIPFLAGS", "").split()      # tuning experiments, e.g. -DSF3D_PROPS_WAVES=4
        cmd = [HIPCC, *HIP_FLAGS, *extra, f"-I{INCLUDE}", f"-I{CSRC}", "-x", "hip", *map(str, srcs), "-o", str(PRODUCT_LIB)]
        _run(cmd)
    return PRODUCT_LIB


def build_shim(force: bool = False) -> Path:
    """The C++ drop-in: the reference's 70 soilFluxes3D::v2 symbols forwarding to the C ABI."""
    src = ROOT / "shim" / "sf3d_cxx_shim.cpp"
    if not src.exists():
        return SHIM_LIB
    deps = [src, ROOT / "shim" / "soilFluxes3D_api.h", INCLUDE / "sf3d.h"]
    if force or _stale(SHIM_LIB, deps):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", f"-I{INCLUDE}", f"-I{ROOT / 'shim'}", str(src),
              "-o", str(SHIM_LIB), f"-L{CSRC}", "-lsf3d_hip", f"-Wl,-rpath,{CSRC}"])
    return SHIM_LIB


V1_LIB = ROOT / "shim" / "libsoilFluxes3D_v1_mi355x.so"


def build_v1_alias(force: bool = False) -> Path:
    """alias layer for the retired soilFluxes3D::v1 names (initializeFluxes, ...)"""
    src = ROOT / "shim" / "sf3d_v1_alias.cpp"
    deps = [src, ROOT / "shim" / "soilFluxes3D_v1_api.h", INCLUDE / "sf3d.h"]
    if force or _stale(V1_LIB, deps):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", f"-I{INCLUDE}", f"-I{ROOT / 'shim'}", str(src),
              "-o", str(V1_LIB), f"-L{CSRC}", "-lsf3d_hip", f"-Wl,-rpath,{CSRC}"])
    demo = ROOT / "shim" / "v1_alias_demo"
    dsrc = ROOT / "tests" / "v1_alias_demo.cpp"
    if force or _stale(demo, [dsrc, V1_LIB]):
        _run(["g++", "-std=c++17", "-O2", f"-I{ROOT / 'shim'}", str(dsrc), "-o", str(demo), f"-L{ROOT / 'shim'}",
              "-lsoilFluxes3D_v1_mi355x", f"-L{CSRC}", "-lsf3d_hip", f"-Wl,-rpath,{ROOT / 'shim'}", f"-Wl,-rpath,{CSRC}"])
    return V1_LIB


REFERENCE = Path("/root/reference")


def build_v2_demo(force: bool = False) -> Path:
    """tests/v2_caller_demo.cpp: a C++ caller of the soilFluxes3D::v2 API linked against the shim.  With the reference
    mounted it is compiled against the reference's OWN soilFluxes3D.h (the caller-side header of bin/CRITERIA3D);
    elsewhere (GPU box) the prebuilt binary is used, or shim/soilFluxes3D_api.h if it has to be rebuilt."""
    demo = ROOT / "shim" / "v2_caller_demo"
    dsrc = ROOT / "tests" / "v2_caller_demo.cpp"
    build_shim()
    if force or _stale(demo, [dsrc, SHIM_LIB]):
        ref_inc = REFERENCE / "agrolib" / "soilFluxes3D"
        if (ref_inc / "soilFluxes3D.h").exists():
            inc = ["-DSF3D_USE_REFERENCE_HEADER", f"-I{ref_inc}", f"-I{REFERENCE / 'agrolib' / 'mathFunctions'}"]
        else:
            inc = [f"-I{ROOT / 'shim'}"]
        _run(["g++", "-std=c++17", "-O2", *inc, str(dsrc), "-o", str(demo), f"-L{ROOT / 'shim'}", "-lsoilFluxes3D_mi355x",
              f"-L{CSRC}", "-lsf3d_hip", f"-Wl,-rpath,{ROOT / 'shim'}", f"-Wl,-rpath,{CSRC}"])
    return demo


STATIC_LIB = ROOT / "shim" / "libsoilFluxes3D.a"
QT_INC = Path(os.environ.get("SF3D_QT_INC", "/opt/conda/include/qt"))
QT_CORE = Path(os.environ.get("SF3D_QT_CORE", "/opt/conda/lib/libQt5Core.so.5"))


def build_static_dropin(force: bool = False):
    """INTEGRATION.md section 2, executed: the shim compiled against the application's OWN headers (soilFluxes3D.h, types.h,
    lineal/linealiaLib.h -> <QLibrary>) with the LinealiaLib stub, archived as libsoilFluxes3D.a - the file name
    bin/CRITERIA3D/CRITERIA3D.pro:65,93 links - and tests/v2_caller_demo.cpp (plus the LinealiaLib::instance().load() call of
    main.cpp:81) linked against that archive the way the .pro change says.  Needs the reference headers and Qt: returns None
    elsewhere (the GPU box runs the prebuilt demo)."""
    ref_inc = REFERENCE / "agrolib" / "soilFluxes3D"
    if not (ref_inc / "soilFluxes3D.h").exists() or not (QT_INC / "QtCore").exists() or not QT_CORE.exists():
        return None
    src = ROOT / "shim" / "sf3d_cxx_shim.cpp"
    obj = ROOT / "shim" / "sf3d_cxx_shim_static.o"
    inc = [f"-I{ref_inc}", f"-I{ref_inc / 'lineal'}", f"-I{REFERENCE / 'agrolib' / 'mathFunctions'}", f"-I{INCLUDE}",
           f"-I{QT_INC}", f"-I{QT_INC / 'QtCore'}"]
    if force or _stale(STATIC_LIB, [src, INCLUDE / "sf3d.h"]):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-c", str(src), "-DSF3D_USE_REFERENCE_HEADER", "-DSF3D_WITH_LINEALIA_STUB", *inc, "-o", str(obj)])
        if STATIC_LIB.exists():
            STATIC_LIB.unlink()
        _run(["ar", "rcs", str(STATIC_LIB), str(obj)])
    demo = ROOT / "shim" / "v2_static_demo"
    dsrc = ROOT / "tests" / "v2_caller_demo.cpp"
    build_product()
    if force or _stale(demo, [dsrc, STATIC_LIB, PRODUCT_LIB]):
        # the application's link line after the INTEGRATION.md change: -lsoilFluxes3D (the archive) + the HIP library + Qt
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-DSF3D_USE_REFERENCE_HEADER", "-DSF3D_DEMO_LINEALIA", *inc, str(dsrc), "-o", str(demo),
              f"-L{ROOT / 'shim'}", "-l:libsoilFluxes3D.a", f"-L{CSRC}", "-lsf3d_hip", f"-Wl,-rpath,{CSRC}", str(QT_CORE)])
        # (no rpath to the image's conda Qt: that directory also holds an older libstdc++ that must not shadow the system one -
        # tests run the demo with LD_PRELOAD="<system libstdc++> <libQt5Core>", a real deployment has Qt on the loader path)
    return STATIC_LIB


def build_oracle(with_reference: bool = True) -> None:
    """Test infrastructure: the CPU restatement and (when /root/reference is present) oracle/_ref."""
    _run(["make", "-C", str(ROOT / "oracle"), "oracle"])
    _run(["make", "-C", str(ROOT / "oracle"), "oracle-fm"])        # the fast-math twin (a diagnostic of tests/test_gpu_sensitivity.py, never a pin)
    if with_reference:
        _run(["make", "-C", str(ROOT / "oracle"), "ref"])
        _run(["make", "-C", str(ROOT / "oracle"), "ref-tuned"])
        _run(["make", "-C", str(ROOT / "oracle"), "ref-ndebug"])


RCCL_MOCK = ROOT / "tests" / "librccl_mock.so"


def build_rccl_mock(force: bool = False) -> Path:
    """Test infrastructure (tests/rccl_mock.cpp): the nine RCCL entry points the opt-in RCCL exchange resolves, carried by shared
    memory, so that its sequencing runs with the ranks of a test sharing one GPU.  Host code only: g++ against the HIP runtime."""
    src = ROOT / "tests" / "rccl_mock.cpp"
    rocm = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))
    if force or _stale(RCCL_MOCK, [src]):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", f"-I{rocm / 'include'}", str(src), "-o", str(RCCL_MOCK),
              f"-L{rocm / 'lib'}", "-lamdhip64", "-lrt", "-lpthread", f"-Wl,-rpath,{rocm / 'lib'}"])
    return RCCL_MOCK


def build_all(force: bool = False) -> None:
    build_product(force)
    build_shim(force)
    build_v1_alias(force)
    build_v2_demo(force)
    build_static_dropin(force)
    build_rccl_mock(force)
    build_oracle()
