"""Build the native libraries of pdwt_amd IN-TREE (pdwt_amd/lib/):

  libpdwt_hip.so   hand-written gfx950 kernels + the C-ABI of include/pdwt_hip.h   (hipcc)
  libpdwt.so       host C++ `Wavelets` class, DTYPE=float                           (g++)
  libpdwtd.so      host C++ `Wavelets` class, DTYPE=double (-DDOUBLEPRECISION)      (g++)

The two host libraries mirror the reference's product shape (Makefile:29-39).  hipcc cross-compiles
for gfx950 without a GPU, so this runs in the build container; the .so files travel to the GPU box
with the repo snapshot.  Usage:  python -m pdwt_amd.build [--force]
"""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "lib")
INC = os.path.join(ROOT, "include")

# (the slowest translation units first: the pool starts jobs in this order, and the build's wall time is the longest chain)
HIP_SOURCES = ["dwt_lat.hip", "swt_fused_l2_fwd.hip", "swt_fused_l2_inv.hip", "swt_fused_l2_inv1.hip", "swt_fused_l2_inv2.hip", "swt_fused_l2_inv4.hip", "dwt_lds.hip", "dwt_casc_inv3.hip", "dwt_casc.hip", "dwt_casc_invw.hip", "swt_fused_inv.hip", "swt_fused_invp.hip", "swt_fused_fwd.hip", "swt_fused_fwd_long.hip", "swt_fused_inv_long.hip", "swt_fused_f64_fwd.hip", "swt_fused_f64_inv.hip", "dwt1d_fused.hip", "dwt1d_fused_nt.hip", "dwt_stream.hip", "cols_ring_dwt_f32.hip", "cols_ring_dwt_f64.hip", "cols_ring_swt_f32.hip", "cols_ring_swt_f64.hip", "rows_tr.hip", "runtime.hip", "coeffs.hip", "dwt.hip", "swt.hip", "haar.hip", "utils.hip", "nonsep.hip", "collective.hip", "selfcheck.hip", "filters.cpp"]
HOST_SOURCES = ["wt.cpp", "wt_capi.cpp"]
# (kept for reference; an object's real dependencies are the files its source includes, transitively: _deps())
HIP_DEPS = ["common.hpp", "dwt_stream.hpp", "stream_dev.hpp", "dwt_casc.hpp", "casc_dev.hpp", "dwt_lds.hpp", "swt_fused.hpp", "swt_fused.inc", "swt_fused_l2.inc", "swt_fused_f64.inc", "dwt1d_fused.hpp", "cols_ring.hpp", "cols_ring.inc", "rows_tr.hpp", "tapreg.hpp", "filters_table.inc"]
ARCH = "gfx950"


def _newer(srcs, out):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(s) > t for s in srcs)


def _deps(path, seen=None):
    """`path` and every file it #include "..."s, transitively (csrc/ and include/ only): what an object really depends on, so that touching
    one header rebuilds the translation units that use it and nothing else (the whole library is ~8 minutes on 8 cores)."""
    import re
    seen = set() if seen is None else seen
    if path in seen or not os.path.exists(path):
        return seen
    seen.add(path)
    for m in re.finditer(r'^\s*#\s*include\s+"([^"]+)"', open(path, errors="replace").read(), re.M):
        inc = m.group(1)
        for base in (os.path.dirname(path), CSRC, INC):
            cand = os.path.normpath(os.path.join(base, inc))
            if os.path.exists(cand):
                _deps(cand, seen)
                break
    return seen


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def build_hip(force=False):
    """Compile every HIP translation unit to an object (in parallel), then link libpdwt_hip.so."""
    from concurrent.futures import ThreadPoolExecutor
    out = os.path.join(LIB, "libpdwt_hip.so")
    objdir = os.path.join(PKG, "build")
    extra = os.environ.get("PDWT_HIPCC_FLAGS", "").split()
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)
    # the library is current (e.g. on the GPU box, where the object directory does not travel): nothing to do
    alldeps = set()
    for src in HIP_SOURCES:
        _deps(os.path.join(CSRC, src), alldeps)
    have_objs = all(os.path.exists(os.path.join(objdir, os.path.splitext(s)[0] + ".o")) for s in HIP_SOURCES)
    if not force and not extra and not have_objs and not _newer(sorted(alldeps), out):
        return out  # (no object directory to check against: the library is newer than every source)
    jobs, objs = [], []
    for src in HIP_SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or extra or _newer(sorted(_deps(sp)), obj):
            jobs.append([hipcc(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall"] + extra + ["-c", sp, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(max(8, (os.cpu_count() or 8) // 2), len(jobs))) as ex:
            list(ex.map(_run, jobs))
    if jobs or not os.path.exists(out):
        _run([hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_host(force=False):
    outs = []
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    deps = srcs + [os.path.join(INC, "wt.h"), os.path.join(INC, "pdwt_hip.h")]
    for name, flags in (("libpdwt.so", []), ("libpdwtd.so", ["-DDOUBLEPRECISION"])):
        out = os.path.join(LIB, name)
        if force or _newer(deps + [os.path.join(LIB, "libpdwt_hip.so")], out):
            _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall"] + flags + ["-o", out] + srcs
                 + ["-L" + LIB, "-lpdwt_hip", "-Wl,-rpath,$ORIGIN"])
        outs.append(out)
    return outs


def build_demo(force=False):
    """The drop-in acceptance program (examples/demo.cpp) against include/wt.h + libpdwt.so / libpdwtd.so."""
    src = os.path.join(ROOT, "examples", "demo.cpp")
    outs = []
    bsrc = os.path.join(ROOT, "examples", "batch_demo.cpp")
    for name, lib, flags, src in (("demo", "pdwt", [], src), ("demod", "pdwtd", ["-DDOUBLEPRECISION"], src),
                                  ("batch_demo", "pdwt", [], bsrc), ("batch_demod", "pdwtd", ["-DDOUBLEPRECISION"], bsrc)):
        out = os.path.join(LIB, name)
        if force or _newer([src, os.path.join(INC, "wt.h"), os.path.join(INC, "wt_batch.h"), os.path.join(LIB, "lib%s.so" % lib)], out):
            _run(["g++", "-O2", "-std=c++17", "-Wall"] + flags + ["-I" + INC, src, "-L" + LIB, "-l" + lib, "-lpdwt_hip",
                  "-Wl,-rpath,$ORIGIN", "-o", out])
        outs.append(out)
    return outs


def build_all(force=False):
    return [build_hip(force)] + build_host(force) + build_demo(force)


if __name__ == "__main__":
    for p in build_all("--force" in sys.argv):
        print(p)
