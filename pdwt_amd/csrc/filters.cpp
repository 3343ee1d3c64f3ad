// filters.cpp -- wavelet name -> filter bank (host only).
// Replaces w_compute_filters_separable (reference src/separable.cu:19-54): same lookup semantics
// (case-insensitive linear scan over the 72 names of src/filters.cpp:5919-6002; the Haar aliases
// short-circuit to hlen = 2 when !do_swt, src/separable.cu:24-28; unknown name -> -2), but the taps
// are handed back to the caller instead of being uploaded to process-global constant memory
// (SURVEY.md Appendix B-1).  Values: generated from PyWavelets by tools/gen_filters.py (B-7).
#include <string.h>
#include <strings.h>

#include "../../include/pdwt_hip.h"

namespace {
struct Bank {
    const char* name;
    int hlen;
    double L[PDWT_MAX_FILTER_WIDTH], H[PDWT_MAX_FILTER_WIDTH], IL[PDWT_MAX_FILTER_WIDTH], IH[PDWT_MAX_FILTER_WIDTH];
};
#define PDWT_FILTER(name, hlen, ...) {name, hlen, __VA_ARGS__},
const Bank g_banks[] = {
#include "filters_table.inc"
};
#undef PDWT_FILTER
constexpr int kNumBanks = sizeof(g_banks) / sizeof(g_banks[0]);
static_assert(kNumBanks == 72, "reference table has 72 entries (src/filters.cpp:5919-6002)");

bool is_haar_alias(const char* w)
{
    // "rbior1.1" is the reference's spelling (src/separable.cu:25); "rbio1.1" is pywt's.
    return !strcasecmp(w, "haar") || !strcasecmp(w, "db1") || !strcasecmp(w, "bior1.1") || !strcasecmp(w, "rbior1.1") ||
           !strcasecmp(w, "rbio1.1");
}

const Bank* find(const char* wname)
{
    for (int i = 0; i < kNumBanks; i++)
        if (!strcasecmp(wname, g_banks[i].name)) return &g_banks[i];
    return nullptr;
}

template <typename F, typename T>
int compute(const char* wname, int do_swt, F* out)
{
    if (!wname) return PDWT_EINVAL;
    (void)do_swt;  // the Haar aliases resolve to the same 2-tap bank either way; callers pick the kernels
    const Bank* b = find(is_haar_alias(wname) ? "haar" : wname);
    if (!b) return PDWT_EUNKNOWN;
    if (out) {
        memset(out, 0, sizeof(*out));
        out->hlen = b->hlen;
        for (int i = 0; i < b->hlen; i++) {
            out->L[i] = (T)b->L[i];
            out->H[i] = (T)b->H[i];
            out->IL[i] = (T)b->IL[i];
            out->IH[i] = (T)b->IH[i];
        }
    }
    return b->hlen;
}
}  // namespace

extern "C" {
int pdwt_compute_filters_separable_f32(const char* wname, int do_swt, pdwt_filters_f32* out)
{
    return compute<pdwt_filters_f32, float>(wname, do_swt, out);
}
int pdwt_compute_filters_separable_f64(const char* wname, int do_swt, pdwt_filters_f64* out)
{
    return compute<pdwt_filters_f64, double>(wname, do_swt, out);
}
int pdwt_num_wavelets(void) { return kNumBanks; }
const char* pdwt_wavelet_name(int idx) { return (idx >= 0 && idx < kNumBanks) ? g_banks[idx].name : nullptr; }
}
