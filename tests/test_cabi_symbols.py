"""CPU-only: the C-ABI libraries load without a GPU and export every symbol include/pdwt_hip.h
declares; host-side logic that needs no device (filter lookup, band geometry) behaves like the
reference's."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import pdwt_amd
from pdwt_amd import _native as N
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "pdwt_hip.h")).read()
    names = set(re.findall(r"\b(pdwt_[a-z0-9_]+)\s*\(", src))
    # expand the PDWT_DECL_DRIVERS macro: pdwt_xxx_##S
    macro = set(re.findall(r"\b(pdwt_[a-z0-9_]+)_##S\(", src))
    out = {n for n in names if not n.endswith("_")}
    for m in macro:
        out |= {m + "_f32", m + "_f64"}
    return out


def test_every_declared_symbol_is_exported():
    L = pdwt_amd.hip()
    declared = _declared_symbols()
    assert len(declared) > 60
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    listed = set(N.PLAIN_SYMBOLS) | {"pdwt_%s_%s" % (t, s) for t in N.TYPED_SYMBOLS for s in ("f32", "f64")}
    assert declared == listed, (declared ^ listed)


def test_host_libraries_load_and_report_dtype():
    assert N.host(np.float32).pdwt_wavelets_sizeof_dtype() == 4
    assert N.host(np.float64).pdwt_wavelets_sizeof_dtype() == 8


def test_filter_lookup_matches_oracle_and_reference_semantics():
    L = pdwt_amd.hip()
    assert L.pdwt_num_wavelets() == 72
    for i in range(72):
        name = L.pdwt_wavelet_name(i).decode()
        f = N.Filters64()
        h = L.pdwt_compute_filters_separable_f64(name.encode(), 0, C.byref(f))
        ho, taps, _ = orc.filters(name, np.float64)
        assert h == ho == f.hlen and 2 <= h <= 40 and h % 2 == 0
        for k in ("L", "H", "IL", "IH"):
            assert np.array_equal(np.array(getattr(f, k)[:h]), taps[k])
        f32 = N.Filters32()
        assert L.pdwt_compute_filters_separable_f32(name.upper().encode(), 1, C.byref(f32)) == h  # case-insensitive
        assert np.array_equal(np.array(f32.L[:h], dtype=np.float32), taps["L"].astype(np.float32))
    assert L.pdwt_compute_filters_separable_f32(b"nosuch", 0, None) == -2  # src/separable.cu:42-45
    for alias in (b"haar", b"db1", b"bior1.1", b"rbior1.1"):
        assert L.pdwt_compute_filters_separable_f32(alias, 0, None) == 2  # src/separable.cu:24-28


@pytest.mark.parametrize("ndims,do_swt", [(2, 0), (1, 0), (2, 1), (1, 1)])
def test_band_geometry(ndims, do_swt):
    L = pdwt_amd.hip()
    for Nr, Nc, lev in ((512, 512, 3), (63, 65, 2), (5, 1000, 4), (4096, 4096, 3)):
        info = N.Info(ndims, Nr, Nc, lev, do_swt, 8)
        shapes = orc.band_shapes(Nr, Nc, lev, do_swt, ndims)
        assert L.pdwt_num_bands(info) == len(shapes)
        for k, (r, c) in enumerate(shapes):
            br, bc = C.c_int(), C.c_int()
            assert L.pdwt_band_size(info, k, C.byref(br), C.byref(bc)) == r * c
            assert (br.value, bc.value) == (r, c)
    assert L.pdwt_band_size(N.Info(2, 8, 8, 1, 0, 2), 4, None, None) < 0
    assert L.pdwt_tmp_elems(N.Info(2, 100, 50, 1, 0, 2)) >= 2 * 100 * 50


def test_product_fails_loudly_without_gpu():
    if pdwt_amd.hip().pdwt_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        pdwt_amd.Wavelets(np.zeros((8, 8), np.float32), "db2", 1)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under pdwt_amd/ may reference it."""
    pkg = os.path.join(ROOT, "pdwt_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h")):
                txt = open(os.path.join(d, f), errors="ignore").read()
                assert "import oracle" not in txt and "from oracle" not in txt and "libpdwt_oracle" not in txt and "orc_" not in txt, f


def test_header_is_plain_c(tmp_path):
    """The drop-in boundary is a C ABI: include/pdwt_hip.h must compile as C99 (no C++-isms, no device headers), and
    include/wt.h with a plain host C++ compiler in both precisions (the reference header needs the CUDA toolkit)."""
    import shutil
    import subprocess
    inc = os.path.join(ROOT, "include")
    c = tmp_path / "c99.c"
    c.write_text('#include "pdwt_hip.h"\nint main(void) { pdwt_info i; (void)i; return pdwt_device_count() < -1; }\n')
    subprocess.check_call([shutil.which("gcc") or "gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", inc, "-fsyntax-only", str(c)])
    cpp = tmp_path / "w.cpp"
    cpp.write_text('#include "wt.h"\nint main() { return sizeof(Wavelets) > 0 ? 0 : 1; }\n')
    for flags in ([], ["-DDOUBLEPRECISION"]):
        subprocess.check_call([shutil.which("g++") or "g++", "-std=c++11", "-Wall", "-I", inc, "-fsyntax-only", str(cpp)] + flags)


def test_rccl_abi_constants_match_the_header_when_present():
    """collective.hip states the few RCCL declarations it needs itself (no build dependency on the RCCL headers: the library is
    dlopen'ed).  Where the image ships rccl.h, its enumerators and prototypes must be the ones stated there."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no rccl.h in this image")
    h = open(hdr).read()
    src = open(os.path.join(ROOT, "pdwt_amd", "csrc", "collective.hip")).read()
    assert not re.search(r"^\s*#\s*include\s*<rccl", src, re.M)
    for name, val in (("ncclSuccess", 0), ("ncclInvalidArgument", 4), ("ncclSum", 0), ("ncclDouble", 8)):
        m = re.search(r"\b%s\s*=\s*(\d+)" % name, h)
        assert m and int(m.group(1)) == val, name
        assert re.search(r"\b%s = %d\b" % (name, val), src), name
    flat = re.sub(r"\s+", " ", h)
    assert "ncclResult_t ncclCommInitAll(ncclComm_t* comm, int ndev, const int* devlist);" in flat
    assert ("ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, "
            "ncclComm_t comm, hipStream_t stream);") in flat
    assert "ncclResult_t ncclCommDestroy(ncclComm_t comm);" in flat
    assert "ncclResult_t ncclGroupStart();" in flat and "ncclResult_t ncclGroupEnd();" in flat


def test_every_run_time_switch_is_documented_and_readable_without_a_gpu():
    """pdwt_debug_get / pdwt_debug_set (include/pdwt_hip.h): every knob of the table in runtime.hip answers on a box without a GPU (the
    table is host state, read from the environment once) and has its row in INTEGRATION.md section E; an unknown key is PDWT_EINVAL."""
    src = open(os.path.join(ROOT, "pdwt_amd", "csrc", "runtime.hip")).read()
    table = re.findall(r'\{"([a-z0-9_]+)",\s*"(PDWT_[A-Z0-9_]+)",\s*(-?\d+)\}', src)
    assert len(table) >= 40
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    L = pdwt_amd.hip()
    for key, env, default in table:
        v = C.c_int(-12345)
        assert L.pdwt_debug_get(key.encode(), C.byref(v)) == 0, key
        if env not in os.environ:
            assert v.value == int(default), (key, v.value, default)
        assert "`%s`" % key in doc, "knob %s has no row in INTEGRATION.md section E" % key
        assert env == "PDWT_" + key.upper(), (key, env)
    v = C.c_int(0)
    assert L.pdwt_debug_get(b"no_such_knob", C.byref(v)) == -1
    assert L.pdwt_debug_set(b"no_such_knob", 1) == -1
    # a set is visible to the next get, and restoring it leaves the table as it was
    assert L.pdwt_debug_set(b"nonsep_tiled", 0) == 0 and L.pdwt_debug_get(b"nonsep_tiled", C.byref(v)) == 0 and v.value == 0
    assert L.pdwt_debug_set(b"nonsep_tiled", 1) == 0


def test_hot_kernels_keep_their_taps_in_scalar_registers():
    """Static guard (no GPU): the kernels of the BASELINE configs and of the common wavelets must not park filter taps in VGPR lanes and
    read them back per use (v_readlane: what slowed the first long-bank SWT inverse 2x and the 12-16-tap inverse levels by 5-11 %), nor
    spill vector registers to scratch.  tools/isa_audit.py disassembles the code objects inside the built libpdwt_hip.so."""
    import importlib.util
    if not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no llvm-objdump")
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_audit.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rows = {mod.demangle(n): (c, m) for n, c, m in mod.audit()}
    assert len(rows) > 500
    hot = {  # kernel (as c++filt prints it) -> largest tolerated share of v_readlane among its instructions
        "k_fwd2d_casc<8, 2, 16, true>": 0.01, "k_inv2d_casc3<8, 16, true, true>": 0.01, "k_fwd2d_stream<8, 2>": 0.01,      # C2
        "k_swt_fwd_fused<14, 0>": 0.01, "k_swt_inv_fused4<14, 1>": 0.01, "k_swt_inv_fused4<14, 2>": 0.01,                  # C3
        "k_swt_inv_fusedp<14, 4>": 0.01, "k_swt_inv_fusedp<14, 8>": 0.01, "k_swt_inv_fusedp<14, 16>": 0.01,
        "dflt::k_fwd1d_fused<float, 16, true>": 0.06, "dflt::k_inv1d_fused_pf<float, 16>": 0.06,  # C4 (read-lanes in the per-row set-up only, none in the item loops)
        "nt::k_fwd1d_fused<float, 16, true>": 0.06, "nt::k_inv1d_fused_pf<float, 16>": 0.06,      # ... the same file compiled with non-temporal loads (round 6)
        "k_fwd2d_lat<40>": 0.0, "k_inv2d_lat<40>": 0.0,                                # C5 levels 1-2 (round 6: lattice level kernels; no scalar spill at all)
        "k_fwd2d_f64lds<double, 40>": 0.06, "k_inv2d_f64lds<double, 40, 256>": 0.06,  # C5 levels 3-6 (~25 per 300 FMAs: per-step bookkeeping scalars, not taps)
        "k_inv2d_stream<8>": 0.01, "k_inv2d_stream<12>": 0.01, "k_inv2d_stream<16>": 0.01,                                  # db4 ... db8 / sym8 levels
        "k_swt_inv_fused2<24, 1>": 0.03, "k_swt_inv_fused2<32, 1>": 0.06, "k_swt_inv_fused2<32, 2>": 0.06,                  # db11 ... db16 SWT
    }
    for k, lim in hot.items():
        assert k in rows, "kernel %s is not in the library any more: update this list" % k
        c, m = rows[k]
        tot = sum(c.values())
        share = c["v_readlane_b32"] / tot
        assert share <= lim, (k, c["v_readlane_b32"], tot)
        assert m.get("vgpr_spill_count", 0) == 0 and not any(op.startswith("scratch_") for op in c), (k, "scratch spills")
