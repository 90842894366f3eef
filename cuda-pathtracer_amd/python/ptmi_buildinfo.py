"""Identity of the built product, used to tell whether a committed rocprofv3 profile still describes the code that runs.

A profile under profiles/ carries two stamps (tools/pmc_summary.py writes them, bench.py checks them):
  lib_sha256         SHA-256 of cuda-pathtracer_amd/libptmi.so as it was profiled
  kernel_src_sha256  SHA-256 over the sources that are compiled into the render kernels (and the build flags)
The counters of a kernel stay valid while kernel_src_sha256 matches, even if an unrelated part of the library changed.
"""
import hashlib
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(_HERE)
ROOT = os.path.dirname(PKG)

# everything that reaches build/kernels.o (csrc/kernels.hip and what it includes) + the flags it is compiled with
KERNEL_SOURCES = [
    os.path.join(PKG, "csrc", "kernels.hip"), os.path.join(PKG, "csrc", "pt_device.h"), os.path.join(PKG, "csrc", "pt_vec.h"),
    os.path.join(PKG, "csrc", "device_scene.h"), os.path.join(PKG, "csrc", "wide_bvh.h"), os.path.join(ROOT, "include", "ptmi_math.h"),
    os.path.join(PKG, "Makefile"),
    # the host loop that decides how the kernels are launched (chunks, run-ahead, segments per launch)
    os.path.join(PKG, "host", "application_state.cpp"),
    # the builder of the opt-in fast tree: the tree's shape decides what ptmi_bounce_wide fetches
    os.path.join(PKG, "host", "wide_bvh.cpp"),
]
# everything that reaches build/radiosity.o (the radiosity pre-pass kernels) + its flags + the host code that launches them
SOLVER_SOURCES = [
    os.path.join(PKG, "csrc", "radiosity.hip"), os.path.join(PKG, "csrc", "pt_device.h"), os.path.join(PKG, "csrc", "pt_vec.h"),
    os.path.join(PKG, "csrc", "device_scene.h"), os.path.join(PKG, "csrc", "wide_bvh.h"), os.path.join(ROOT, "include", "ptmi_math.h"),
    os.path.join(PKG, "Makefile"), os.path.join(PKG, "host", "application_state.cpp"),
]


def _sha(paths):
    h = hashlib.sha256()
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def lib_sha256():
    p = os.environ.get("PTMI_LIB") or os.path.join(PKG, "libptmi.so")
    if not os.path.exists(p):
        return None
    with open(p, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()        # = `sha256sum libptmi.so`


def kernel_src_sha256():
    return _sha(KERNEL_SOURCES)


def solver_src_sha256():
    return _sha(SOLVER_SOURCES)


def stamps():
    return {"lib_sha256": lib_sha256(), "kernel_src_sha256": kernel_src_sha256(), "solver_src_sha256": solver_src_sha256()}


def profile_is_current(profile, solver=False):
    """(current?, which stamp matched).  solver: the profile describes the radiosity pre-pass kernels (csrc/radiosity.hip)"""
    if profile.get("lib_sha256") and profile.get("lib_sha256") == lib_sha256():
        return True, "lib"
    if solver:
        if profile.get("solver_src_sha256") and profile.get("solver_src_sha256") == solver_src_sha256():
            return True, "solver_sources"
        return False, None
    if profile.get("kernel_src_sha256") and profile.get("kernel_src_sha256") == kernel_src_sha256():
        return True, "kernel_sources"
    return False, None
