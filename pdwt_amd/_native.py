"""ctypes bindings of the in-tree native libraries (pdwt_amd/lib/*.so).

There is NO CPU fallback: if a library is missing or no HIP device is visible, the product path
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# PDWT_LIBDIR: an alternative set of the in-tree libraries (diagnostic builds: tools/build_trace.sh, tools/ab_libs.sh)
LIBDIR = os.environ.get("PDWT_LIBDIR") or os.path.join(_PKG, "lib")


class Info(C.Structure):
    """== pdwt_info (include/pdwt_hip.h) == reference w_info (src/utils.h:9-19)."""
    _fields_ = [("ndims", C.c_int), ("Nr", C.c_int), ("Nc", C.c_int), ("nlevels", C.c_int), ("do_swt", C.c_int), ("hlen", C.c_int)]

    def __repr__(self):
        return "Info(ndims=%d, Nr=%d, Nc=%d, nlevels=%d, do_swt=%d, hlen=%d)" % (self.ndims, self.Nr, self.Nc, self.nlevels, self.do_swt, self.hlen)


def _filters_struct(ct):
    class F(C.Structure):
        _fields_ = [("hlen", C.c_int), ("L", ct * 40), ("H", ct * 40), ("IL", ct * 40), ("IH", ct * 40)]
    return F


Filters32 = _filters_struct(C.c_float)
Filters64 = _filters_struct(C.c_double)

DRIVERS = ["forward_separable", "forward_separable_1d", "inverse_separable", "inverse_separable_1d",
           "forward_swt_separable", "forward_swt_separable_1d", "inverse_swt_separable", "inverse_swt_separable_1d"]
HAAR_DRIVERS = ["haar_forward2d", "haar_inverse2d", "haar_forward1d", "haar_inverse1d"]

# every symbol include/pdwt_hip.h declares (checked by tests/test_cabi_symbols.py)
PLAIN_SYMBOLS = ["pdwt_device_count", "pdwt_set_device", "pdwt_get_device", "pdwt_device_name", "pdwt_malloc", "pdwt_free",
                 "pdwt_memset", "pdwt_memcpy_h2d", "pdwt_memcpy_d2h", "pdwt_memcpy_d2d", "pdwt_memcpy_d2d_foreign", "pdwt_set_stream", "pdwt_sync", "pdwt_get_stream",
                 "pdwt_last_error_string", "pdwt_event_create", "pdwt_event_record", "pdwt_event_sync", "pdwt_event_elapsed_ms",
                 "pdwt_event_destroy", "pdwt_ktime_enable", "pdwt_ktime_reset", "pdwt_ktime_read", "pdwt_kernel_name",
                 "pdwt_kernel_count", "pdwt_graph_allowed", "pdwt_graph_capture_begin", "pdwt_graph_capture_end", "pdwt_graph_launch",
                 "pdwt_graph_destroy", "pdwt_num_wavelets", "pdwt_wavelet_name", "pdwt_num_bands", "pdwt_band_size", "pdwt_tmp_elems", "pdwt_debug_set", "pdwt_debug_get", "pdwt_clock_probe_enable", "pdwt_clock_probe_read", "pdwt_clock_probe_dump", "pdwt_probe_bandwidth", "pdwt_selfcheck_vmcnt_order", "pdwt_rccl_available", "pdwt_rccl_allreduce_sum_f64", "pdwt_sum_result_index", "pdwt_sum_spare_index",
                 "pdwt_batch2d_create_f32", "pdwt_batch2d_forward_f32", "pdwt_batch2d_inverse_f32", "pdwt_batch2d_destroy",
                 "pdwt_batch2d_create_f64", "pdwt_batch2d_forward_f64", "pdwt_batch2d_inverse_f64", "pdwt_batch2d_destroy_f64",
                 "pdwt_sum_scratch_doubles", "pdwt_sum_scratch_read"]
TYPED_SYMBOLS = (["compute_filters_separable", "create_coeffs_buffer", "free_coeffs_buffer", "copy_coeffs_buffer",
                  "soft_thresh", "soft_thresh_sum", "norm1", "norm1_as_double", "norm1_enqueue", "hard_thresh", "proj_linf", "shrink", "group_soft_thresh",
                  "norm2sq", "norm2sq_as_double", "add_coeffs", "circshift", "forward_nonseparable", "inverse_nonseparable",
                  "forward_swt_nonseparable", "inverse_swt_nonseparable"] + DRIVERS + HAAR_DRIVERS)

_hip = None
_host = {}


def _require(path):
    if not os.path.exists(path):
        raise RuntimeError("%s is missing: build the native libraries first (python -m pdwt_amd.build); "
                           "pdwt_amd has no CPU fallback" % path)
    return path


def hip():
    """libpdwt_hip.so with argument/return types set."""
    global _hip
    if _hip is not None:
        return _hip
    L = C.CDLL(_require(os.path.join(LIBDIR, "libpdwt_hip.so")), mode=C.RTLD_GLOBAL)
    vp, ci, sz = C.c_void_p, C.c_int, C.c_size_t
    L.pdwt_malloc.restype = vp
    L.pdwt_malloc.argtypes = [sz]
    L.pdwt_free.argtypes = [vp]
    L.pdwt_memset.argtypes = [vp, ci, sz]
    L.pdwt_set_stream.argtypes = [vp, ci]
    for n in ("pdwt_memcpy_h2d", "pdwt_memcpy_d2h", "pdwt_memcpy_d2d", "pdwt_memcpy_d2d_foreign"):
        getattr(L, n).argtypes = [vp, vp, sz]
    L.pdwt_get_stream.restype = vp
    L.pdwt_last_error_string.restype = C.c_char_p
    L.pdwt_device_name.argtypes = [C.c_char_p, ci]
    L.pdwt_event_create.restype = vp
    for n in ("pdwt_event_record", "pdwt_event_sync", "pdwt_event_destroy"):
        getattr(L, n).argtypes = [vp]
    L.pdwt_probe_bandwidth.argtypes = [vp, vp, C.c_size_t, ci]
    L.pdwt_selfcheck_vmcnt_order.restype = C.c_longlong
    L.pdwt_event_elapsed_ms.restype = C.c_float
    L.pdwt_event_elapsed_ms.argtypes = [vp, vp]
    L.pdwt_ktime_read.argtypes = [ci, C.POINTER(ci), C.POINTER(C.c_double)]
    L.pdwt_kernel_name.restype = C.c_char_p
    L.pdwt_kernel_name.argtypes = [ci]
    L.pdwt_wavelet_name.restype = C.c_char_p
    L.pdwt_wavelet_name.argtypes = [ci]
    L.pdwt_num_bands.argtypes = [Info]
    L.pdwt_band_size.restype = C.c_longlong
    L.pdwt_band_size.argtypes = [Info, ci, C.POINTER(ci), C.POINTER(ci)]
    L.pdwt_clock_probe_enable.argtypes = [ci]
    L.pdwt_batch2d_create_f32.restype = vp
    L.pdwt_batch2d_create_f32.argtypes = [ci, vp, vp, vp, Info]
    L.pdwt_batch2d_forward_f32.argtypes = [vp, vp]
    L.pdwt_batch2d_inverse_f32.argtypes = [vp, vp]
    L.pdwt_batch2d_destroy.argtypes = [vp]
    L.pdwt_batch2d_create_f64.restype = vp
    L.pdwt_batch2d_create_f64.argtypes = [ci, vp, vp, vp, Info]
    L.pdwt_batch2d_forward_f64.argtypes = [vp, vp]
    L.pdwt_batch2d_inverse_f64.argtypes = [vp, vp]
    L.pdwt_batch2d_destroy_f64.argtypes = [vp]
    L.pdwt_clock_probe_dump.argtypes = [vp, ci]
    L.pdwt_sum_scratch_doubles.restype = sz
    L.pdwt_sum_result_index.restype = sz
    L.pdwt_sum_spare_index.restype = sz
    L.pdwt_sum_scratch_read.argtypes = [vp, C.POINTER(C.c_double)]
    L.pdwt_clock_probe_read.argtypes = [ci, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.pdwt_debug_set.argtypes = [C.c_char_p, ci]
    L.pdwt_debug_get.argtypes = [C.c_char_p, C.POINTER(ci)]
    L.pdwt_tmp_elems.restype = sz
    L.pdwt_tmp_elems.argtypes = [Info]
    for sfx, ct, FT in (("f32", C.c_float, Filters32), ("f64", C.c_double, Filters64)):
        P = C.POINTER(ct)
        PP = C.POINTER(P)
        f = getattr(L, "pdwt_compute_filters_separable_" + sfx)
        f.argtypes = [C.c_char_p, ci, C.POINTER(FT)]
        f = getattr(L, "pdwt_create_coeffs_buffer_" + sfx)
        f.restype = PP
        f.argtypes = [Info]
        getattr(L, "pdwt_free_coeffs_buffer_" + sfx).argtypes = [PP, Info]
        getattr(L, "pdwt_copy_coeffs_buffer_" + sfx).argtypes = [PP, PP, Info]
        getattr(L, "pdwt_soft_thresh_" + sfx).argtypes = [PP, ct, Info, ci, ci]
        getattr(L, "pdwt_norm1_" + sfx).argtypes = [PP, Info, P]
        getattr(L, "pdwt_norm1_as_double_" + sfx).argtypes = [PP, Info, C.POINTER(C.c_double)]
        getattr(L, "pdwt_norm1_enqueue_" + sfx).argtypes = [PP, Info, vp]
        getattr(L, "pdwt_soft_thresh_sum_" + sfx).argtypes = [PP, ct, Info, ci, ci, vp]
        for n in ("hard_thresh", "group_soft_thresh"):
            getattr(L, "pdwt_%s_%s" % (n, sfx)).argtypes = [PP, ct, Info, ci, ci]
        for n in ("proj_linf", "shrink"):
            getattr(L, "pdwt_%s_%s" % (n, sfx)).argtypes = [PP, ct, Info, ci]
        getattr(L, "pdwt_norm2sq_" + sfx).argtypes = [PP, Info, P]
        getattr(L, "pdwt_norm2sq_as_double_" + sfx).argtypes = [PP, Info, C.POINTER(C.c_double)]
        getattr(L, "pdwt_add_coeffs_" + sfx).argtypes = [PP, PP, Info, ct]
        getattr(L, "pdwt_circshift_" + sfx).argtypes = [vp, vp, Info, ci, ci, ci]
        for d in DRIVERS:
            getattr(L, "pdwt_%s_%s" % (d, sfx)).argtypes = [vp, PP, vp, Info, C.POINTER(FT)]
        for d in HAAR_DRIVERS:
            getattr(L, "pdwt_%s_%s" % (d, sfx)).argtypes = [vp, PP, vp, Info]
    _hip = L
    return L


def host(dtype):
    """libpdwt.so (float32) / libpdwtd.so (float64): the C++ Wavelets class behind a C handle API."""
    dt = np.dtype(dtype)
    if dt not in _host:
        hip()  # resolve libpdwt_hip.so first (same directory, RTLD_GLOBAL)
        name = {np.dtype(np.float32): "libpdwt.so", np.dtype(np.float64): "libpdwtd.so"}[dt]
        L = C.CDLL(_require(os.path.join(LIBDIR, name)))
        ct = C.c_float if dt == np.float32 else C.c_double
        vp, ci = C.c_void_p, C.c_int
        assert L.pdwt_wavelets_sizeof_dtype() == dt.itemsize
        L.pdwt_wavelets_new.restype = vp
        L.pdwt_wavelets_new.argtypes = [vp, ci, ci, C.c_char_p, ci, ci, ci, ci, ci, ci]
        L.pdwt_wavelets_copy.restype = vp
        L.pdwt_wavelets_copy.argtypes = [vp]
        L.pdwt_wavelets_delete.argtypes = [vp]
        for n in ("forward", "inverse", "print_informations"):
            getattr(L, "pdwt_wavelets_" + n).argtypes = [vp]
        L.pdwt_wavelets_soft_threshold.argtypes = [vp, ct, ci, ci]
        L.pdwt_wavelets_norm1.restype = ct
        L.pdwt_wavelets_norm1.argtypes = [vp]
        L.pdwt_wavelets_norm1_f64.restype = C.c_double
        L.pdwt_wavelets_norm1_f64.argtypes = [vp]
        L.pdwt_wavelets_set_norm_cache.argtypes = [vp, ci]
        L.pdwt_wavelets_norm1_begin.argtypes = [vp]
        L.pdwt_wavelets_norm1_end.restype = C.c_double
        L.pdwt_wavelets_norm1_end.argtypes = [vp]
        L.pdwt_wavelets_norm2sq.restype = ct
        L.pdwt_wavelets_norm2sq.argtypes = [vp]
        for n in ("hard_threshold", "group_soft_threshold"):
            getattr(L, "pdwt_wavelets_" + n).argtypes = [vp, ct, ci, ci]
        for n in ("shrink", "proj_linf"):
            getattr(L, "pdwt_wavelets_" + n).argtypes = [vp, ct, ci]
        L.pdwt_wavelets_circshift.argtypes = [vp, ci, ci, ci]
        L.pdwt_wavelets_set_filters_forward.argtypes = [vp, C.c_char_p, C.c_uint, vp, vp]
        L.pdwt_wavelets_set_filters_inverse.argtypes = [vp, vp, vp]
        L.pdwt_wavelets_set_filters_forward4.argtypes = [vp, C.c_char_p, C.c_uint, vp, vp, vp, vp]
        L.pdwt_wavelets_set_filters_inverse4.argtypes = [vp, vp, vp, vp, vp]
        L.pdwt_wavelets_add_wavelet.argtypes = [vp, vp, ct]
        L.pdwt_wavelets_shifts.argtypes = [vp, C.POINTER(ci), C.POINTER(ci)]
        L.pdwt_wavelets_get_image.argtypes = [vp, vp]
        L.pdwt_wavelets_set_image.argtypes = [vp, vp, ci]
        L.pdwt_wavelets_get_coeff.argtypes = [vp, vp, ci]
        L.pdwt_wavelets_set_coeff.argtypes = [vp, vp, ci, ci]
        L.pdwt_wavelets_state.argtypes = [vp]
        L.pdwt_wavelets_set_state.argtypes = [vp, ci]
        L.pdwt_wavelets_info.argtypes = [vp, C.POINTER(Info)]
        for n in ("image_int_ptr", "coeffs_table_ptr", "tmp_int_ptr"):
            getattr(L, "pdwt_wavelets_" + n).restype = C.c_ssize_t
            getattr(L, "pdwt_wavelets_" + n).argtypes = [vp]
        L.pdwt_images_new.restype = vp
        L.pdwt_images_new.argtypes = [vp, ci, ci, ci, C.c_char_p, ci, ci]
        L.pdwt_images_new_swt.restype = vp
        L.pdwt_images_new_swt.argtypes = [vp, ci, ci, ci, C.c_char_p, ci, ci, ci]
        L.pdwt_images_at.restype = vp
        L.pdwt_images_at.argtypes = [vp, ci]
        for n in ("delete", "ok", "batched", "forward", "inverse"):
            getattr(L, "pdwt_images_" + n).argtypes = [vp]
        L.pdwt_wavelets_coeff_int_ptr.restype = C.c_ssize_t
        L.pdwt_wavelets_coeff_int_ptr.argtypes = [vp, ci]
        _host[dt] = L
    return _host[dt]


def require_gpu():
    n = hip().pdwt_device_count()
    if n <= 0:
        raise RuntimeError("pdwt_amd: no HIP device visible (MI355X required; there is no CPU fallback)")
    return n
