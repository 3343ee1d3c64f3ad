"""Python view of the C++ ``Wavelets`` class (include/wt.h) through its C handle API.

Method names, argument meaning, defaults and the state machine are the reference's
(src/wt.h:20-76, src/wt.cu); numpy arrays stand in for the host float*/double* buffers.
"""
import ctypes as C

import numpy as np

from . import _native as N

W_INIT, W_FORWARD, W_INVERSE, W_THRESHOLD, W_CREATION_ERROR, W_FORWARD_ERROR, W_INVERSE_ERROR, W_THRESHOLD_ERROR = range(8)


class DeviceArray:
    """Zero-copy view of device memory owned by a ``Wavelets`` instance (``d_image`` or one band of ``d_coeffs``).

    Exposes ``__cuda_array_interface__`` (v2; PyTorch-ROCm, CuPy-ROCm and numba consume it for HIP memory too):
    ``torch.as_tensor(W.coeff_view(1), device="cuda")`` is a tensor over the band itself, no copy.  The view keeps
    its owner alive.  The library works on its own stream: call ``Wavelets.sync()`` before the consumer reads and
    synchronise the consumer's stream before the next transform (the reference's
    ``image_int_ptr`` / ``coeff_int_ptr`` interop, src/wt.cu:660-667, with shape and dtype attached)."""

    def __init__(self, owner, ptr, shape, dtype):
        self._owner, self.ptr, self.shape, self.dtype = owner, int(ptr), tuple(int(v) for v in shape), np.dtype(dtype)

    @property
    def __cuda_array_interface__(self):
        return {"shape": self.shape, "typestr": self.dtype.str, "data": (self.ptr, False), "version": 2, "strides": None}

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        if N.hip().pdwt_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr), out.nbytes) != 0:
            raise RuntimeError("device-to-host copy failed")
        return out


def _sync_producer():
    """A device buffer handed to the library was produced on SOME stream of the caller: wait for torch's current
    stream (the usual producer; it need not be the NULL stream the C side waits for) before the pointer is used."""
    import sys
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available():
        torch.cuda.current_stream().synchronize()


def _device_source(img):
    """(pointer, shape, dtype) of an object that lives in device memory (torch tensor on the GPU or anything with
    ``__cuda_array_interface__``), else None."""
    if hasattr(img, "is_cuda") and hasattr(img, "data_ptr"):  # torch tensor, without importing torch here
        if not img.is_cuda:
            return None
        if not img.is_contiguous():
            raise ValueError("device tensors must be contiguous")
        dt = {"torch.float32": np.float32, "torch.float64": np.float64}.get(str(img.dtype))
        if dt is None:
            raise TypeError("device tensors must be float32 or float64")
        return int(img.data_ptr()), tuple(img.shape), np.dtype(dt)
    cai = getattr(img, "__cuda_array_interface__", None)
    if cai is not None:
        if cai.get("strides") not in (None,):
            raise ValueError("device arrays must be C-contiguous")
        return int(cai["data"][0]), tuple(cai["shape"]), np.dtype(cai["typestr"])
    return None


class Wavelets:
    def __init__(self, img, wname, levels, do_separable=1, do_cycle_spinning=0, do_swt=0, ndim=2, dtype=None, shape=None, device_ptr=None,
                 norm_cache=True):
        """Wavelets(img, Nr, Nc, wname, levels, memisonhost, do_separable, do_cycle_spinning, do_swt, ndim)
        (src/wt.h:42).  ``img`` is a 2-D (or 1-D) numpy array; or pass ``device_ptr`` + ``shape`` for
        an image already in HBM (memisonhost=0), or img=None + shape for a zero image.

        ``norm_cache``: the C++ class reduces the bands on every ``norm1()`` unless ``set_norm_cache(1)`` was called
        (include/wt.h: its ``d_coeffs`` is a public member, a caller's kernel may write a band unseen).  This wrapper
        hands band pointers out ONLY through ``coeff_int_ptr`` / ``coeff_view``, which switch the shortcut off for the
        instance, so it is safe by construction here and on by default: ``soft_threshold`` leaves sum|c| behind for the
        ``norm1`` that follows.  ``norm_cache=False`` gives the reference behaviour."""
        N.require_gpu()
        dev = _device_source(img) if img is not None else None
        if dev is not None:  # image already in HBM: memisonhost = 0, the constructor copies device-to-device
            device_ptr, shape, dtype = dev[0], (dev[1] if len(dev[1]) == 2 else (1, dev[1][0])), dev[2]
            img = None
        if img is not None:
            img = np.asarray(img)
            if dtype is None:
                dtype = img.dtype if img.dtype in (np.float32, np.float64) else np.float32
            img = np.ascontiguousarray(img, dtype=dtype)
            if img.ndim == 1:
                img = img[None, :]
            shape = img.shape
        elif shape is None:
            raise ValueError("img or shape is required")
        self.dtype = np.dtype(dtype or np.float32)
        self.shape = (int(shape[0]), int(shape[1]))
        self._L = N.host(self.dtype)
        self._ct = C.c_float if self.dtype == np.float32 else C.c_double
        if device_ptr is not None:
            _sync_producer()
            src, on_host = C.c_void_p(int(device_ptr)), 0
        elif img is not None:
            src, on_host = img.ctypes.data_as(C.c_void_p), 1
        else:
            src, on_host = None, 1
        self._h = self._L.pdwt_wavelets_new(src, self.shape[0], self.shape[1], wname.encode(), int(levels), on_host, int(do_separable),
                                            int(do_cycle_spinning), int(do_swt), int(ndim))
        if not self._h:
            raise MemoryError("Wavelets allocation failed")
        self.wname = wname
        if norm_cache:
            self._L.pdwt_wavelets_set_norm_cache(self._h, 1)

    def set_norm_cache(self, on=True):
        """Wavelets::set_norm_cache (include/wt.h): norm1() right after soft_threshold() without a second pass."""
        self._L.pdwt_wavelets_set_norm_cache(self._h, 1 if on else 0)

    @classmethod
    def _from_handle(cls, other, h):
        o = cls.__new__(cls)
        o.dtype, o.shape, o._L, o._ct, o.wname, o._h = other.dtype, other.shape, other._L, other._ct, other.wname, h
        return o

    def copy(self):  # copy constructor, src/wt.cu:191-222
        return Wavelets._from_handle(self, self._L.pdwt_wavelets_copy(self._h))

    def close(self):
        if getattr(self, "_h", None) and not getattr(self, "_borrowed", False):  # (a view into an ImageBatch belongs to the batch)
            self._L.pdwt_wavelets_delete(self._h)
        self._h = None

    __del__ = close

    # -- introspection ---------------------------------------------------------------------
    @property
    def info(self):
        i = N.Info()
        self._L.pdwt_wavelets_info(self._h, C.byref(i))
        return i

    @property
    def state(self):
        return self._L.pdwt_wavelets_state(self._h)

    @state.setter
    def state(self, s):
        self._L.pdwt_wavelets_set_state(self._h, int(s))

    @property
    def nbands(self):
        return N.hip().pdwt_num_bands(self.info)

    def band_shape(self, num):
        r, c = C.c_int(), C.c_int()
        n = N.hip().pdwt_band_size(self.info, int(num), C.byref(r), C.byref(c))
        if n <= 0:
            raise IndexError(num)
        return r.value, c.value

    # -- the Wavelets API ------------------------------------------------------------------
    def forward(self):
        self._L.pdwt_wavelets_forward(self._h)

    def inverse(self):
        self._L.pdwt_wavelets_inverse(self._h)

    def soft_threshold(self, beta, do_thresh_appcoeffs=0, normalize=0):
        self._L.pdwt_wavelets_soft_threshold(self._h, self._ct(beta), int(do_thresh_appcoeffs), int(normalize))

    def norm1(self):
        return self.dtype.type(self._L.pdwt_wavelets_norm1(self._h))

    def hard_threshold(self, beta, do_thresh_appcoeffs=0, normalize=0):
        self._L.pdwt_wavelets_hard_threshold(self._h, self._ct(beta), int(do_thresh_appcoeffs), int(normalize))

    def group_soft_threshold(self, beta, do_thresh_appcoeffs=0, normalize=0):
        self._L.pdwt_wavelets_group_soft_threshold(self._h, self._ct(beta), int(do_thresh_appcoeffs), int(normalize))

    def shrink(self, beta, do_thresh_appcoeffs=1):
        self._L.pdwt_wavelets_shrink(self._h, self._ct(beta), int(do_thresh_appcoeffs))

    def proj_linf(self, beta, do_thresh_appcoeffs=1):
        self._L.pdwt_wavelets_proj_linf(self._h, self._ct(beta), int(do_thresh_appcoeffs))

    def circshift(self, sr, sc, inplace=1):
        self._L.pdwt_wavelets_circshift(self._h, int(sr), int(sc), int(inplace))

    def norm2sq(self):
        return self.dtype.type(self._L.pdwt_wavelets_norm2sq(self._h))

    def set_filters_forward(self, name, lo, hi):
        """Custom analysis bank (reference Wavelets::set_filters_forward, separable form)."""
        a = np.ascontiguousarray(lo, dtype=self.dtype)
        b = np.ascontiguousarray(hi, dtype=self.dtype)
        assert a.size == b.size
        return self._L.pdwt_wavelets_set_filters_forward(self._h, name.encode(), a.size, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))

    def set_filters_inverse(self, lo, hi):
        a = np.ascontiguousarray(lo, dtype=self.dtype)
        b = np.ascontiguousarray(hi, dtype=self.dtype)
        assert a.size == b.size == self.info.hlen
        return self._L.pdwt_wavelets_set_filters_inverse(self._h, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p))

    def set_filters_forward_nonseparable(self, name, k_ll, k_lh, k_hl, k_hh):
        """Four genuinely non-separable len x len analysis kernels (do_separable=0 instances only)."""
        ks = [np.ascontiguousarray(k, dtype=self.dtype) for k in (k_ll, k_lh, k_hl, k_hh)]
        n = ks[0].shape[0]
        assert all(k.shape == (n, n) for k in ks)
        return self._L.pdwt_wavelets_set_filters_forward4(self._h, name.encode(), n, *[k.ctypes.data_as(C.c_void_p) for k in ks])

    def set_filters_inverse_nonseparable(self, k_ll, k_lh, k_hl, k_hh):
        ks = [np.ascontiguousarray(k, dtype=self.dtype) for k in (k_ll, k_lh, k_hl, k_hh)]
        n = self.info.hlen
        assert all(k.shape == (n, n) for k in ks)
        return self._L.pdwt_wavelets_set_filters_inverse4(self._h, *[k.ctypes.data_as(C.c_void_p) for k in ks])

    def add_wavelet(self, other, alpha=1.0):
        """self += alpha * other on every band (reference Wavelets::add_wavelet)."""
        return self._L.pdwt_wavelets_add_wavelet(self._h, other._h, self._ct(alpha))

    @property
    def current_shift(self):
        r, c = C.c_int(), C.c_int()
        self._L.pdwt_wavelets_shifts(self._h, C.byref(r), C.byref(c))
        return r.value, c.value

    def get_tmp(self):
        """First Nr*Nc elements of d_tmp (where circshift(..., inplace=0) leaves its result)."""
        out = np.empty(self.shape, dtype=self.dtype)
        rc = N.hip().pdwt_memcpy_d2h(out.ctypes.data_as(C.c_void_p), C.c_void_p(self._L.pdwt_wavelets_tmp_int_ptr(self._h)), out.nbytes)
        if rc != 0:
            raise RuntimeError("get_tmp failed")
        return out

    def norm1_f64(self):
        """Sum of |c| over all bands, in double (pdwt_norm1_as_double_*): shard-combinable."""
        return float(self._L.pdwt_wavelets_norm1_f64(self._h))

    def norm1_begin(self):
        """Enqueue the reduction of norm1() on the instance's device without waiting for it (include/wt.h: norm1_begin)."""
        self._L.pdwt_wavelets_norm1_begin(self._h)

    def norm1_end(self):
        """Wait for the reduction norm1_begin() started and return the sum in double (alone: the same as norm1_f64())."""
        return float(self._L.pdwt_wavelets_norm1_end(self._h))

    def get_image(self):
        out = np.empty(self.shape, dtype=self.dtype)
        n = self._L.pdwt_wavelets_get_image(self._h, out.ctypes.data_as(C.c_void_p))
        if n != min(out.size, 2**31 - 1):  # the C++ method returns an int element count: saturated past 2^31 - 1
            raise RuntimeError("get_image failed")
        return out

    def sync(self):
        """Wait for the library stream (everything enqueued by this process on the current device)."""
        return N.hip().pdwt_sync()

    def image_view(self):
        """``d_image`` as a zero-copy DeviceArray (the reference's image_int_ptr with shape and dtype)."""
        return DeviceArray(self, self.image_int_ptr(), self.shape, self.dtype)

    def coeff_view(self, num):
        """Band ``num`` of ``d_coeffs`` as a zero-copy DeviceArray (the reference's coeff_int_ptr)."""
        return DeviceArray(self, self.coeff_int_ptr(num), self.band_shape(num), self.dtype)

    def set_image(self, img, mem_is_on_device=0):
        dev = _device_source(img)
        if dev is not None:
            assert dev[2] == self.dtype and int(np.prod(dev[1])) == self.shape[0] * self.shape[1]
            img, mem_is_on_device = dev[0], 1
        if mem_is_on_device:
            _sync_producer()
            self._L.pdwt_wavelets_set_image(self._h, C.c_void_p(int(img)), 1)
        else:
            a = np.ascontiguousarray(img, dtype=self.dtype)
            assert a.size == self.shape[0] * self.shape[1]
            self._L.pdwt_wavelets_set_image(self._h, a.ctypes.data_as(C.c_void_p), 0)

    def get_coeff(self, num):
        r, c = self.band_shape(num)
        out = np.empty((r, c), dtype=self.dtype)
        n = self._L.pdwt_wavelets_get_coeff(self._h, out.ctypes.data_as(C.c_void_p), int(num))
        if n != min(out.size, 2**31 - 1):
            raise RuntimeError("get_coeff(%d) failed (state=%d)" % (num, self.state))
        return out

    def set_coeff(self, arr, num):
        r, c = self.band_shape(num)
        dev = _device_source(arr)
        if dev is not None:
            assert dev[2] == self.dtype and int(np.prod(dev[1])) == r * c
            _sync_producer()
            self._L.pdwt_wavelets_set_coeff(self._h, C.c_void_p(dev[0]), int(num), 1)
            return
        a = np.ascontiguousarray(arr, dtype=self.dtype)
        assert a.size == r * c
        self._L.pdwt_wavelets_set_coeff(self._h, a.ctypes.data_as(C.c_void_p), int(num), 0)

    @property
    def coeffs(self):
        return [self.get_coeff(i) for i in range(self.nbands)]

    def print_informations(self):
        self._L.pdwt_wavelets_print_informations(self._h)

    def image_int_ptr(self):
        return self._L.pdwt_wavelets_image_int_ptr(self._h)

    def coeff_int_ptr(self, num):
        return self._L.pdwt_wavelets_coeff_int_ptr(self._h, int(num))


class ImageBatch:
    """A batch of equally sized 2-D images on one GPU transformed together (include/wt_batch.h: WaveletsImages): every level of ALL
    images runs in one launch when the geometry is inside the streaming level kernels (``batched``), otherwise image after image.
    ``imgs``: array (B, Nr, Nc), numpy (host) or a contiguous device tensor.  ``batch[b]`` is the ordinary ``Wavelets`` view of image
    b (coefficients, thresholds ... between forward() and inverse()); results equal the per-image transforms bit for bit.
    ``do_swt=1``: the undecimated transform (batched in float32 when every level is inside the fused SWT level kernels)."""

    def __init__(self, imgs, wname, levels, dtype=None, do_swt=0):
        N.require_gpu()
        dev = _device_source(imgs)
        if dev is not None:
            ptr, shape, dt = dev
            _sync_producer()
            src, on_host = C.c_void_p(ptr), 0
        else:
            imgs = np.asarray(imgs)
            dt = np.dtype(dtype or (imgs.dtype if imgs.dtype in (np.float32, np.float64) else np.float32))
            self._keep = np.ascontiguousarray(imgs, dtype=dt)
            shape, src, on_host = self._keep.shape, self._keep.ctypes.data_as(C.c_void_p), 1
        assert len(shape) == 3, "imgs must be (B, Nr, Nc)"
        self.dtype, self.shape, self.wname = np.dtype(dt), tuple(int(v) for v in shape), wname
        self._L = N.host(self.dtype)
        self._h = self._L.pdwt_images_new_swt(src, self.shape[0], self.shape[1], self.shape[2], wname.encode(), int(levels), on_host, int(bool(do_swt)))
        if not self._h or not self._L.pdwt_images_ok(self._h):
            raise RuntimeError("ImageBatch creation failed")

    @property
    def batched(self):
        return bool(self._L.pdwt_images_batched(self._h))

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, b):
        if not 0 <= b < self.shape[0]:
            raise IndexError(b)
        w = Wavelets.__new__(Wavelets)
        w.dtype, w.shape, w._L, w.wname = self.dtype, self.shape[1:], self._L, self.wname
        w._ct = C.c_float if self.dtype == np.float32 else C.c_double
        w._h, w._borrowed, w._owner = self._L.pdwt_images_at(self._h, int(b)), True, self
        return w

    def forward(self):
        self._L.pdwt_images_forward(self._h)

    def inverse(self):
        self._L.pdwt_images_inverse(self._h)

    def get_images(self):
        return np.stack([self[b].get_image() for b in range(self.shape[0])])

    def sync(self):
        return N.hip().pdwt_sync()

    def close(self):
        if getattr(self, "_h", None):
            self._L.pdwt_images_delete(self._h)
            self._h = None

    __del__ = close
