"""ctypes front-end of the CPU oracle (oracle/libpdwt_oracle.so).

TEST INFRASTRUCTURE ONLY -- see the header of oracle/pdwt_oracle.c.  Importable from tests/,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of bench.py; never from pdwt_amd/.

``OracleWavelets`` mirrors the reference's ``Wavelets`` class (src/wt.h:20-76, src/wt.cu) on the
host so that parity tests read like calls on the real thing: same constructor arguments, same
level clamping (src/wt.cu:155-165), same variant dispatch (src/wt.cu:247-266,283-301), same band
numbering for ``get_coeff`` (src/wt.cu:475-508), same state machine (src/wt.h:8-17).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

W_INIT, W_FORWARD, W_INVERSE, W_THRESHOLD, W_CREATION_ERROR = 0, 1, 2, 3, 4


class Info(C.Structure):  # == w_info, src/utils.h:9-19
    _fields_ = [("ndims", C.c_int), ("Nr", C.c_int), ("Nc", C.c_int), ("nlevels", C.c_int), ("do_swt", C.c_int), ("hlen", C.c_int)]


def _filters_struct(ct):
    class F(C.Structure):
        _fields_ = [("hlen", C.c_int), ("L", ct * 40), ("H", ct * 40), ("IL", ct * 40), ("IH", ct * 40)]
    return F


Filters32 = _filters_struct(C.c_float)
Filters64 = _filters_struct(C.c_double)


def _filters2d_struct(ct):
    class F2(C.Structure):  # four hlen x hlen kernels (LL, LH, HL, HH), src/nonseparable.cu:7-11
        _fields_ = [("hlen", C.c_int), ("K", (ct * 1600) * 4)]
    return F2


Filters2D32 = _filters2d_struct(C.c_float)
Filters2D64 = _filters2d_struct(C.c_double)


def build(force=False):
    so = os.path.join(_HERE, "libpdwt_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("pdwt_oracle.c", "pdwt_oracle_impl.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_norm1_f32.restype = C.c_double
        _LIB.orc_norm1_f64.restype = C.c_double
        # parity tests run many small problems: a 256-thread OpenMP team per loop is pure overhead.
        # The full-size tests raise this to usable_cores(); bench.py's cpu_baseline leg times its own thread counts in its own process.
        _LIB.orc_set_num_threads(min(16, os.cpu_count() or 1))
    return _LIB


def div2(n):  # w_div2, src/utils.cu:24-27
    return (n + 1) // 2


def ilog2(i):  # w_ilog2, src/utils.cu:14-20
    l = 0
    while i > 1:
        i >>= 1
        l += 1
    return l


def set_num_threads(n):
    return lib().orc_set_num_threads(int(n))


def usable_cores(cap=64):
    """CPUs this process is really granted: the affinity mask capped by the container's cgroup CPU quota (a 256-thread host that grants
    16 CPUs throttles any larger OpenMP team), at most `cap`.  What the full-size parity tests give the oracle's team."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, cap))


def max_threads():
    return lib().orc_get_max_threads()


def filters(wname, dtype=np.float32, do_swt=0):
    """(hlen, dict L/H/IL/IH as numpy) or raises KeyError for unknown names (rc -2)."""
    f = Filters32() if np.dtype(dtype) == np.float32 else Filters64()
    fn = lib().orc_compute_filters_f32 if np.dtype(dtype) == np.float32 else lib().orc_compute_filters_f64
    h = fn(wname.encode(), int(do_swt), C.byref(f))
    if h <= 0:
        raise KeyError(wname)
    return h, {k: np.array(getattr(f, k)[:h], dtype=dtype) for k in ("L", "H", "IL", "IH")}, f


def band_shapes(Nr, Nc, nlevels, do_swt, ndims):
    """Shapes of the logical bands [A_L, details...] in PDWT order (src/common.cu:400-445)."""
    shapes = []
    r, c = Nr, Nc
    for _ in range(nlevels):
        if not do_swt:
            if ndims == 2:
                r = div2(r)
            c = div2(c)
        shapes += [(r, c)] * (3 if ndims == 2 else 1)
    return [(r, c)] + shapes


class OracleWavelets:
    """Host mirror of the reference's Wavelets class, computing with the C oracle."""

    def __init__(self, img, wname, levels, do_swt=0, ndim=2, dtype=None, custom_filters=None, do_separable=1):
        img = np.asarray(img)
        if dtype is None:
            dtype = img.dtype if img.dtype in (np.float32, np.float64) else np.float32
        self.dtype = np.dtype(dtype)
        self.sfx = "f32" if self.dtype == np.float32 else "f64"
        if img.ndim == 1:
            img = img[None, :]
        Nr, Nc = img.shape
        self.state = W_INIT
        if levels < 1:  # src/wt.cu:111-114
            levels = 1
        if Nr == 1:  # src/wt.cu:133-136
            ndim = 1
        self.wname = wname
        self.do_swt = int(do_swt)
        self.do_separable = 1 if (ndim == 1 or Nr == 1) else int(do_separable)  # src/wt.cu:137-141: ignored in 1-D
        self._F2 = None  # custom non-separable kernels: (forward Filters2D, inverse Filters2D)
        if custom_filters is not None:
            hlen, self._F = custom_filters
        else:
            alias = wname.lower() in ("haar", "db1", "bior1.1", "rbior1.1")
            if alias and not do_swt:  # src/separable.cu:24-28
                hlen, self._F = 2, None
            else:
                hlen, _, self._F = filters(wname, self.dtype, do_swt)
        N = min(Nr, Nc) if ndim == 2 else Nc
        wmaxlev = ilog2(N // (hlen - 1))  # src/wt.cu:155-165
        if levels > wmaxlev:
            levels = wmaxlev
        self.info = Info(ndim, Nr, Nc, levels, self.do_swt, hlen)
        self.image = np.ascontiguousarray(img, dtype=self.dtype).copy()
        self.tmp = np.zeros(2 * Nr * Nc + 64, dtype=self.dtype)
        shapes = band_shapes(Nr, Nc, levels, self.do_swt, ndim)
        self.shapes = shapes
        # band 0 is allocated at level-1 size (src/common.cu:421-423, 441-443)
        r0, c0 = (Nr, Nc) if self.do_swt else ((div2(Nr) if ndim == 2 else Nr), div2(Nc))
        self._bufs = [np.zeros(r0 * c0, dtype=self.dtype)] + [np.zeros(s[0] * s[1], dtype=self.dtype) for s in shapes[1:]]
        ct = C.c_float if self.dtype == np.float32 else C.c_double
        self._PT = C.POINTER(ct)
        self._ctab = (self._PT * len(self._bufs))(*[b.ctypes.data_as(self._PT) for b in self._bufs])

    # -- helpers ---------------------------------------------------------------------------
    def _p(self, a):
        return a.ctypes.data_as(self._PT)

    def _driver(self, direction):
        """Variant selection of Wavelets::forward/inverse (src/wt.cu:247-266, 283-301)."""
        i = self.info
        if i.hlen == 2 and not i.do_swt:
            return "haar_%s%dd" % (direction, i.ndims), False
        return direction + ("_swt" if i.do_swt else "") + "_separable" + ("_1d" if i.ndims == 1 else ""), True

    def _filters2d(self, direction):
        """The four 2-D kernels of the non-separable path: custom ones if set, else the outer products of the 1-D
        bank (w_compute_filters(wname, +1 | -1), src/nonseparable.cu:32-83; src/wt.cu:296 reloads them for the inverse)."""
        if self._F2 is not None:
            return self._F2[0 if direction == "forward" else 1]
        F2 = (Filters2D32 if self.dtype == np.float32 else Filters2D64)()
        lo, hi = (self._F.L, self._F.H) if direction == "forward" else (self._F.IL, self._F.IH)
        getattr(lib(), "orc_outer_filters_" + self.sfx)(lo, hi, self.info.hlen, C.byref(F2))
        return F2

    def set_filters_forward_nonseparable(self, name, k_ll, k_lh, k_hl, k_hh):  # src/wt.cu:560-581, nonseparable.cu:86-95
        ks = [np.asarray(k, dtype=self.dtype) for k in (k_ll, k_lh, k_hl, k_hh)]
        n = ks[0].shape[0]
        F2f = (Filters2D32 if self.dtype == np.float32 else Filters2D64)()
        F2f.hlen = n
        for b in range(4):
            flat = ks[b].reshape(-1)
            for i in range(n * n):
                F2f.K[b][i] = flat[i]
        self._F2 = (F2f, None)
        self.info.hlen = n
        self.wname = name
        return 0

    def set_filters_inverse_nonseparable(self, k_ll, k_lh, k_hl, k_hh):  # src/wt.cu:584-602
        ks = [np.asarray(k, dtype=self.dtype) for k in (k_ll, k_lh, k_hl, k_hh)]
        n = self.info.hlen
        F2i = (Filters2D32 if self.dtype == np.float32 else Filters2D64)()
        F2i.hlen = n
        for b in range(4):
            flat = ks[b].reshape(-1)
            for i in range(n * n):
                F2i.K[b][i] = flat[i]
        self._F2 = (self._F2[0], F2i)
        return 0

    def _call(self, direction):
        i = self.info
        if not self.do_separable and i.ndims == 2 and not (i.hlen == 2 and not i.do_swt and self._F2 is None):
            # Wavelets::forward/inverse, do_separable == 0 branch (src/wt.cu:257-260, 294-299)
            fn = getattr(lib(), "orc_%s%s_nonseparable_%s" % (direction, "_swt" if i.do_swt else "", self.sfx))
            F2 = self._filters2d(direction)
            rc = fn(self._p(self.image), self._ctab, self._p(self.tmp), self.info, C.byref(F2))
            assert rc == 0
            return
        name, needs_filters = self._driver(direction)
        fn = getattr(lib(), "orc_%s_%s" % (name, self.sfx))
        args = [self._p(self.image), self._ctab, self._p(self.tmp), self.info]
        if needs_filters:
            args.append(C.byref(self._F))
        rc = fn(*args)
        assert rc == 0

    # -- Wavelets API ----------------------------------------------------------------------
    def forward(self):  # src/wt.cu:236-271
        self._call("forward")
        self.state = W_FORWARD

    def inverse(self):  # src/wt.cu:273-307
        if self.state == W_INVERSE:
            return
        self._call("inverse")
        self.state = W_INVERSE

    def soft_threshold(self, beta, do_thresh_appcoeffs=0, normalize=0):  # src/wt.cu:310-317
        if self.state == W_INVERSE:
            return
        fn = getattr(lib(), "orc_soft_thresh_" + self.sfx)
        ct = C.c_float if self.dtype == np.float32 else C.c_double
        fn(self._ctab, ct(beta), self.info, int(do_thresh_appcoeffs), int(normalize))

    def norm1(self):  # src/wt.cu:398-418
        v = getattr(lib(), "orc_norm1_" + self.sfx)(self._ctab, self.info)
        return self.dtype.type(v)

    def norm1_f64(self):
        return float(getattr(lib(), "orc_norm1_" + self.sfx)(self._ctab, self.info))

    # -- remaining coefficient utilities: numpy restatements, elementwise in DTYPE ----------
    def _band(self, num):
        """writable view of the coefficients of band `num`"""
        r, c = self.shapes[num]
        return self._bufs[num][: r * c]

    def _walk(self, beta, do_thresh_appcoeffs, normalize, app_normalized):
        """Band walk shared by w_call_{soft,hard}_thresh / w_call_proj_linf / w_shrink (src/common.cu:219-315):
        yields (band view, beta for that band); beta is divided by sqrt(2) per level when normalize > 0, the
        approximation band takes beta/sqrt(2)^L only where the reference really passes it (soft threshold)."""
        T = self.dtype.type
        beta = T(beta)
        per = 3 if self.info.ndims == 2 else 1
        L = self.info.nlevels
        if do_thresh_appcoeffs:
            beta2 = beta
            if normalize > 0 and app_normalized:  # src/common.cu:231-235
                nl2 = L // 2
                beta2 = T(beta2 / T(1 << nl2))
                if nl2 * 2 != L:
                    beta2 = T(np.float64(beta2) / 1.4142135623730951)
            yield self._band(0), beta2
        for lev in range(L):
            if normalize > 0:
                beta = T(np.float64(beta) / 1.4142135623730951)  # src/common.cu:244
            for b in range(per):
                yield self._band(per * lev + 1 + b), beta

    def hard_threshold(self, beta, do_thresh_appcoeffs=0, normalize=0):  # src/wt.cu:320-327, common.cu:57-94, 252-283
        if self.state == W_INVERSE:
            return
        # the approximation band gets the UN-normalised beta (common.cu:270 passes beta, not beta2)
        for v, b in self._walk(beta, do_thresh_appcoeffs, normalize, app_normalized=False):
            v *= (np.abs(v) - b > 0).astype(self.dtype)  # max(W_SIGN(|v|-b), 0) * v

    def proj_linf(self, beta, do_thresh_appcoeffs=1):  # src/wt.cu:350-357, common.cu:96-131, 286-315
        if self.state == W_INVERSE:
            return
        for v, b in self._walk(beta, do_thresh_appcoeffs, 0, app_normalized=False):
            v[...] = np.copysign(np.minimum(np.abs(v), b), v)

    def shrink(self, beta, do_thresh_appcoeffs=1):  # src/wt.cu:341-348, common.cu:346-371: scal by 1/(1+beta)
        if self.state == W_INVERSE:
            return
        T = self.dtype.type
        f = T(T(1) / (T(1) + T(beta)))
        for v, _ in self._walk(beta, do_thresh_appcoeffs, 0, app_normalized=False):
            v *= f

    def group_soft_threshold(self, beta, do_thresh_appcoeffs=0, normalize=0):  # src/wt.cu:331-338, common.cu:134-198, 318-343
        if self.state == W_INVERSE:
            return
        T = self.dtype.type
        beta = T(beta)
        per = 3 if self.info.ndims == 2 else 1
        L = self.info.nlevels
        for lev in range(L):
            if normalize > 0:
                beta = T(np.float64(beta) / 1.4142135623730951)
            bands = [self._band(per * lev + 1 + b) for b in range(per)]
            if do_thresh_appcoeffs and lev == L - 1:
                bands.append(self._band(0))
            nrm = np.zeros_like(bands[0])
            for v in bands:
                nrm += v * v
            nrm = np.sqrt(nrm)
            with np.errstate(divide="ignore", invalid="ignore"):
                res = np.where(nrm == 0, T(0), np.maximum(T(1) - beta / nrm, T(0))).astype(self.dtype)
            for v in bands:
                v *= res

    def norm2sq(self, ref_quirk_1d=False):
        """src/wt.cu:370-395: sum of squares over all bands.  The reference's 1-D branch adds asum of the details
        (src/wt.cu:389, a bug -- SURVEY B-4): restated only when ``ref_quirk_1d``."""
        acc = 0.0
        for k in range(len(self.shapes)):
            v = self._band(k).astype(np.float64)
            if ref_quirk_1d and self.info.ndims == 1 and k > 0:
                acc += np.abs(v).sum()
            else:
                acc += (v * v).sum()
        return self.dtype.type(acc)

    def add_wavelet(self, other, alpha=1.0):  # src/wt.cu:624-657, common.cu:499-526 (whole bands)
        if self.info.nlevels != other.info.nlevels or self.wname.lower() != other.wname.lower():
            return -1
        if self.state == W_INVERSE or other.state == W_INVERSE:
            return 1
        if (self.info.Nr, self.info.Nc, self.info.ndims) != (other.info.Nr, other.info.Nc, other.info.ndims):
            return -2
        if bool(self.info.do_swt) != bool(other.info.do_swt):
            return -3
        a = self.dtype.type(alpha)
        for k in range(len(self.shapes)):
            self._band(k)[...] += a * other._band(k)
        return 0

    def circshift(self, sr, sc, inplace=1):  # src/wt.cu:364-366, common.cu:202-211, 378-396
        Nr, Nc = self.info.Nr, self.info.Nc
        if self.info.ndims == 1:
            sr = 0
        out = np.roll(self.image, (sr % Nr, sc % Nc), axis=(0, 1))
        if inplace:
            self.image[...] = out
        else:
            self.tmp[: Nr * Nc] = out.reshape(-1)

    def set_filters_forward(self, name, lo, hi):  # src/wt.cu:560-581
        lo = np.asarray(lo, dtype=self.dtype)
        hi = np.asarray(hi, dtype=self.dtype)
        if lo.size > 40:
            return -1
        if self._F is None:
            self._F = Filters32() if self.dtype == np.float32 else Filters64()
        for i in range(40):
            self._F.L[i] = lo[i] if i < lo.size else 0
            self._F.H[i] = hi[i] if i < hi.size else 0
        self._F.hlen = lo.size
        self.info.hlen = lo.size
        self.wname = name
        return 0

    def set_filters_inverse(self, lo, hi):  # src/wt.cu:584-602
        lo = np.asarray(lo, dtype=self.dtype)
        hi = np.asarray(hi, dtype=self.dtype)
        for i in range(40):
            self._F.IL[i] = lo[i] if i < lo.size else 0
            self._F.IH[i] = hi[i] if i < hi.size else 0
        return 0

    def get_image(self):  # src/wt.cu:421-424
        return self.image.copy()

    def set_image(self, img):  # src/wt.cu:427-433
        self.image[...] = np.asarray(img, dtype=self.dtype).reshape(self.image.shape)
        self.state = W_INIT

    def get_coeff(self, num):  # src/wt.cu:475-508
        r, c = self.shapes[num]
        return self._bufs[num][: r * c].reshape(r, c).copy()

    def set_coeff(self, arr, num):  # src/wt.cu:436-468
        r, c = self.shapes[num]
        self._bufs[num][: r * c] = np.asarray(arr, dtype=self.dtype).reshape(-1)

    @property
    def coeffs(self):
        return [self.get_coeff(i) for i in range(len(self.shapes))]
