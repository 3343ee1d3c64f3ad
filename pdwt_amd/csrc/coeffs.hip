// coeffs.hip -- coefficient-band buffers (include/pdwt_hip.h "Coefficient buffers").
// Layout contract = reference w_create_coeffs_buffer[_1d] (src/common.cu:400-445): a host array of
// device pointers [A_L, H1,V1,D1, ...] (2D) / [A_L, D1, ...] (1D); band sizes by repeated
// ceil-half; band 0 allocated at level-1 size.  MI355X-first difference: ONE hipMalloc for all
// bands (256-byte aligned sub-ranges) instead of 3L+1 cudaMalloc+memset pairs -- one allocation,
// one memset, and the bands of a transform are contiguous in HBM.
#include <stdlib.h>

#include "common.hpp"

namespace pdwt {

int band_geometry(const pdwt_info& w, BandGeom* g)
{
    if (w.Nr < 1 || w.Nc < 1 || w.nlevels < 1 || w.nlevels > 32 || (w.ndims != 1 && w.ndims != 2)) return PDWT_EINVAL;
    int nr = w.Nr, nc = w.Nc;
    const int per = (w.ndims == 2) ? 3 : 1;
    g->nbands = per * w.nlevels + 1;
    // band 0 allocation: level-1 size (src/common.cu:404-408,421-423 / 433-443)
    int r0 = w.Nr, c0 = w.Nc;
    if (!w.do_swt) {
        if (w.ndims == 2) r0 = div2(r0);
        c0 = div2(c0);
    }
    for (int lev = 0; lev < w.nlevels; lev++) {
        if (!w.do_swt) {
            if (w.ndims == 2) nr = div2(nr);
            nc = div2(nc);
        }
        for (int b = 0; b < per; b++) {
            const int k = per * lev + 1 + b;
            g->Nr[k] = nr;
            g->Nc[k] = nc;
            g->alloc_elems[k] = (size_t)nr * nc;
        }
    }
    g->Nr[0] = nr;
    g->Nc[0] = nc;
    g->alloc_elems[0] = (size_t)r0 * c0;
    return PDWT_OK;
}

template <typename T>
static T** create(pdwt_info w)
{
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK) return nullptr;
    size_t off[3 * 32 + 2];
    size_t total = 0;
    for (int k = 0; k < g.nbands; k++) {
        off[k] = total;
        total += (g.alloc_elems[k] * sizeof(T) + 255) & ~(size_t)255;
    }
    char* base = (char*)pdwt_malloc(total);
    if (!base) return nullptr;
    if (pdwt_memset(base, 0, total) != PDWT_OK) {
        (void)hipFree(base);
        return nullptr;
    }
    // slot [-1] of the host table remembers the allocation base for free()
    T** tab = (T**)calloc((size_t)g.nbands + 1, sizeof(T*));
    if (!tab) {
        (void)hipFree(base);
        return nullptr;
    }
    tab[0] = (T*)base;
    for (int k = 0; k < g.nbands; k++) tab[k + 1] = (T*)(base + off[k]);
    return tab + 1;
}

template <typename T>
static int destroy(T** c)
{
    if (!c) return PDWT_OK;
    int rc = pdwt_free((void*)c[-1]);
    free(c - 1);
    return rc;
}

template <typename T>
static int copy(T** dst, T** src, pdwt_info w)
{
    if (!dst || !src) return PDWT_EINVAL;
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK) return PDWT_EINVAL;
    for (int k = 0; k < g.nbands; k++) {
        int rc = pdwt_memcpy_d2d(dst[k], src[k], (size_t)g.Nr[k] * g.Nc[k] * sizeof(T));
        if (rc != PDWT_OK) return rc;
    }
    return PDWT_OK;
}

}  // namespace pdwt

using namespace pdwt;

extern "C" {
float** pdwt_create_coeffs_buffer_f32(pdwt_info w) { return create<float>(w); }
double** pdwt_create_coeffs_buffer_f64(pdwt_info w) { return create<double>(w); }
int pdwt_free_coeffs_buffer_f32(float** c, pdwt_info) { return destroy(c); }
int pdwt_free_coeffs_buffer_f64(double** c, pdwt_info) { return destroy(c); }
int pdwt_copy_coeffs_buffer_f32(float** d, float** s, pdwt_info w) { return copy(d, s, w); }
int pdwt_copy_coeffs_buffer_f64(double** d, double** s, pdwt_info w) { return copy(d, s, w); }

int pdwt_num_bands(pdwt_info w)
{
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK) return PDWT_EINVAL;
    return g.nbands;
}
long long pdwt_band_size(pdwt_info w, int num, int* bnr, int* bnc)
{
    BandGeom g;
    if (band_geometry(w, &g) != PDWT_OK || num < 0 || num >= g.nbands) return PDWT_EINVAL;
    if (bnr) *bnr = g.Nr[num];
    if (bnc) *bnc = g.Nc[num];
    return (long long)g.Nr[num] * g.Nc[num];
}
}
