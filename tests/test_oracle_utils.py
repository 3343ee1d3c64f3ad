"""CPU: the oracle's restatements of the coefficient utilities, custom filters and circular shift (SURVEY.md 8f
rows 1-2) against PyWavelets-based golden vectors (tests/golden/make_golden_utils.py)."""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.helpers import band_err, load_golden

TOLU = 1e-9  # f64 end to end; thresholds are evaluated on oracle coefficients that differ from pywt's by ~1e-11


def _fwd(d, dtype=np.float64):
    O = orc.OracleWavelets(d["input"].astype(dtype), d["wname"], d["levels"])
    O.forward()
    return O


def _cmp(O, d, prefix, tol=TOLU):
    for k in range(d["nbands"]):
        assert band_err(O.get_coeff(k), d["%s%d" % (prefix, k)]) <= tol, (prefix, k)


@pytest.mark.parametrize("op", ["hard", "soft", "proj", "shrink", "group", "softnorm"])
def test_oracle_utilities_match_closed_forms(op):
    d = load_golden("utils64x96_db4_L3")
    beta = float(d["beta"])
    O = _fwd(d)
    _cmp(O, d, "band")
    {"hard": lambda: O.hard_threshold(beta), "soft": lambda: O.soft_threshold(beta), "proj": lambda: O.proj_linf(beta),
     "shrink": lambda: O.shrink(beta), "group": lambda: O.group_soft_threshold(beta), "softnorm": lambda: O.soft_threshold(beta, 0, 1)}[op]()
    _cmp(O, d, op)


def test_oracle_norms():
    d = load_golden("utils64x96_db4_L3")
    O = _fwd(d)
    assert abs(O.norm2sq() - d["norm2sq"]) <= 1e-10 * d["norm2sq"]
    assert abs(O.norm1() - d["norm1"]) <= 1e-10 * d["norm1"]
    # Parseval: the periodised orthogonal transform preserves the squared norm
    assert abs(O.norm2sq() - (d["input"] ** 2).sum()) <= 1e-9 * d["norm2sq"]


def test_oracle_hard_threshold_appcoeff_quirk():
    """w_call_hard_thresh passes the UN-normalised beta to the approximation band (src/common.cu:262-270)."""
    d = load_golden("utils64x96_db4_L3")
    O = _fwd(d)
    a = O.get_coeff(0)
    beta = float(np.median(np.abs(a)))
    O.hard_threshold(beta, 1, 1)
    assert np.array_equal(O.get_coeff(0), np.where(np.abs(a) > beta, a, 0 * a))


def test_oracle_add_wavelet_and_state_rules():
    d = load_golden("utils64x96_db4_L3")
    A, B = _fwd(d), _fwd(d)
    assert A.add_wavelet(B, 0.5) == 0
    for k in range(d["nbands"]):
        assert band_err(A.get_coeff(k), 1.5 * d["band%d" % k]) <= TOLU
    C = orc.OracleWavelets(d["input"], "db3", d["levels"])
    C.forward()
    assert A.add_wavelet(C) == -1
    B.inverse()
    assert A.add_wavelet(B) == 1
    A.inverse()
    before = A._band(1).copy()
    A.hard_threshold(1e9)  # refused after inverse (src/wt.cu:321-324): nothing is zeroed
    assert np.array_equal(A._band(1), before)


def test_oracle_custom_filters():
    d = load_golden("custom80x64_bior33_L2")
    O = orc.OracleWavelets(d["input"], "haar", d["levels"])  # any table entry, then replaced (the reference's usage)
    assert O.set_filters_forward("custom_bior33", d["dec_lo"], d["dec_hi"]) == 0
    assert O.set_filters_inverse(d["rec_lo"], d["rec_hi"]) == 0
    assert O.info.hlen == 8
    O.forward()
    _cmp(O, d, "band", 1e-12)
    O.inverse()
    assert band_err(O.get_image(), d["recon"]) <= 1e-12
    assert O.set_filters_forward("too_long", np.zeros(41), np.zeros(41)) == -1


def test_oracle_circshift_is_the_cycle_spinning_shift():
    d = load_golden("shift48x72_db3_L2")
    O = orc.OracleWavelets(d["input"], d["wname"], d["levels"])
    O.circshift(int(d["sr"]), int(d["sc"]), 1)
    assert np.array_equal(O.image, d["shifted"])
    O.forward()
    _cmp(O, d, "band", 1e-12)
    O.inverse()
    O.circshift(-int(d["sr"]), -int(d["sc"]), 1)
    assert band_err(O.image, d["input"]) <= 1e-12
    P = orc.OracleWavelets(d["input"], d["wname"], d["levels"])
    P.circshift(3, 4, 0)  # result in tmp, image untouched
    assert np.array_equal(P.image, d["input"])
    assert np.array_equal(P.tmp[: 48 * 72].reshape(48, 72), np.roll(d["input"], (3, 4), axis=(0, 1)))


def test_oracle_nonseparable_is_separable_with_h_and_v_exchanged():
    """src/nonseparable.cu builds LH = outer(l, h): low-pass along y, high-pass along x -- the band the separable path calls
    V (the CHECKME at :72-78).  The 2-D restatement therefore equals the golden (pywt) bands with H and V exchanged."""
    for name in ("u64_db4_L3_f64", "odd63x65_db2_L2", "swt64_db3_L3"):
        d = load_golden(name)
        kw = dict(do_swt=1) if d["kind"] == "swt2" else {}
        O = orc.OracleWavelets(d["input"], d["wname"], d["levels"], do_separable=0, **kw)
        O.forward()
        L = d["levels"]
        assert band_err(O.get_coeff(0), d["band0"]) <= 1e-10
        for l in range(L):
            assert band_err(O.get_coeff(3 * l + 1), d["band%d" % (3 * l + 2)]) <= 1e-10, (name, l, "H <- V")
            assert band_err(O.get_coeff(3 * l + 2), d["band%d" % (3 * l + 1)]) <= 1e-10, (name, l, "V <- H")
            assert band_err(O.get_coeff(3 * l + 3), d["band%d" % (3 * l + 3)]) <= 1e-10
        O.inverse()
        assert band_err(O.get_image(), d["recon"]) <= 1e-10


NSEP_CASES = ["nsep_dec_h6_48x64_L2", "nsep_dec_h4_33x47_L2_odd", "nsep_dec_h5_40x56_L1", "nsep_swt_h4_32x48_L2", "nsep_swt_h5_40x56_L2"]


@pytest.mark.parametrize("name", NSEP_CASES)
def test_oracle_custom_nonseparable_kernels_vs_independent_direct_sums(name):
    """Four arbitrary hlen x hlen kernels (src/nonseparable.cu:114-225, 304-401): the oracle's sample-by-sample restatement against the
    array-wise float64 evaluation of the same defining sums in tests/golden/make_golden_nonsep.py (no pywt counterpart exists)."""
    d = load_golden(name)
    swt, L = int(d["swt"]), d["levels"]
    O = orc.OracleWavelets(d["input"], "db2", L, do_separable=0, do_swt=swt)
    assert O.info.nlevels == L
    assert O.set_filters_forward_nonseparable("custom2d", *[d["kf%d" % q] for q in range(4)]) == 0
    assert O.set_filters_inverse_nonseparable(*[d["ki%d" % q] for q in range(4)]) == 0
    O.forward()
    for k in range(d["nbands"]):
        assert band_err(O.get_coeff(k), d["band%d" % k]) <= 1e-12, (name, k)
    O.inverse()
    assert band_err(O.get_image(), d["recon"]) <= 1e-12, name
