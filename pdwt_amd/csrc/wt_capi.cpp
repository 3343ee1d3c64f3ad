// wt_capi.cpp -- flat C handle API over the C++ `Wavelets` class, for FFI callers (ctypes, Cython).
// This is the shape the reference's external Python binding (pypwt, README.md:24) wraps: one opaque
// handle per Wavelets instance, methods as functions.  Compiled into libpdwt.so (float) and
// libpdwtd.so (double) next to wt.cpp; DTYPE follows -DDOUBLEPRECISION.
#include <new>

#include "../../include/wt.h"
#include "../../include/wt_batch.h"

#define W(h) (static_cast<Wavelets*>(h))

double w_last_norm1_double(void);  // wt.cpp: the double behind the calling thread's last norm1()

extern "C" {
int pdwt_wavelets_sizeof_dtype(void) { return (int)sizeof(DTYPE); }

void* pdwt_wavelets_new(DTYPE* img, int Nr, int Nc, const char* wname, int levels, int memisonhost, int do_separable, int do_cycle_spinning,
                        int do_swt, int ndim)
{
    return new (std::nothrow) Wavelets(img, Nr, Nc, wname, levels, memisonhost, do_separable, do_cycle_spinning, do_swt, ndim);
}
void* pdwt_wavelets_copy(void* h) { return h ? new (std::nothrow) Wavelets(*W(h)) : nullptr; }
void pdwt_wavelets_delete(void* h) { delete W(h); }

void pdwt_wavelets_forward(void* h) { W(h)->forward(); }
void pdwt_wavelets_inverse(void* h) { W(h)->inverse(); }
void pdwt_wavelets_soft_threshold(void* h, DTYPE beta, int do_thresh_appcoeffs, int normalize) { W(h)->soft_threshold(beta, do_thresh_appcoeffs, normalize); }
DTYPE pdwt_wavelets_norm1(void* h) { return W(h)->norm1(); }
void pdwt_wavelets_set_norm_cache(void* h, int on) { W(h)->set_norm_cache(on); }
void pdwt_wavelets_norm1_begin(void* h) { W(h)->norm1_begin(); }
double pdwt_wavelets_norm1_end(void* h) { return W(h)->norm1_end(); }
/* norm1() before its rounding to DTYPE (per-shard partial sums are combined in double) */
double pdwt_wavelets_norm1_f64(void* h)
{
    (void)W(h)->norm1();
    return w_last_norm1_double();
}
void pdwt_wavelets_hard_threshold(void* h, DTYPE beta, int do_thresh_appcoeffs, int normalize) { W(h)->hard_threshold(beta, do_thresh_appcoeffs, normalize); }
void pdwt_wavelets_group_soft_threshold(void* h, DTYPE beta, int do_thresh_appcoeffs, int normalize) { W(h)->group_soft_threshold(beta, do_thresh_appcoeffs, normalize); }
void pdwt_wavelets_shrink(void* h, DTYPE beta, int do_thresh_appcoeffs) { W(h)->shrink(beta, do_thresh_appcoeffs); }
void pdwt_wavelets_proj_linf(void* h, DTYPE beta, int do_thresh_appcoeffs) { W(h)->proj_linf(beta, do_thresh_appcoeffs); }
void pdwt_wavelets_circshift(void* h, int sr, int sc, int inplace) { W(h)->circshift(sr, sc, inplace); }
DTYPE pdwt_wavelets_norm2sq(void* h) { return W(h)->norm2sq(); }
int pdwt_wavelets_set_filters_forward(void* h, char* name, unsigned len, DTYPE* f1, DTYPE* f2) { return W(h)->set_filters_forward(name, len, f1, f2); }
int pdwt_wavelets_set_filters_inverse(void* h, DTYPE* f1, DTYPE* f2) { return W(h)->set_filters_inverse(f1, f2); }
int pdwt_wavelets_set_filters_forward4(void* h, char* name, unsigned len, DTYPE* f1, DTYPE* f2, DTYPE* f3, DTYPE* f4) { return W(h)->set_filters_forward(name, len, f1, f2, f3, f4); }
int pdwt_wavelets_set_filters_inverse4(void* h, DTYPE* f1, DTYPE* f2, DTYPE* f3, DTYPE* f4) { return W(h)->set_filters_inverse(f1, f2, f3, f4); }
int pdwt_wavelets_add_wavelet(void* h, void* other, DTYPE alpha) { return W(h)->add_wavelet(*W(other), alpha); }
void pdwt_wavelets_shifts(void* h, int* sr, int* sc) { *sr = W(h)->current_shift_r; *sc = W(h)->current_shift_c; }
int pdwt_wavelets_get_image(void* h, DTYPE* out) { return W(h)->get_image(out); }
void pdwt_wavelets_set_image(void* h, DTYPE* img, int mem_is_on_device) { W(h)->set_image(img, mem_is_on_device); }
int pdwt_wavelets_get_coeff(void* h, DTYPE* out, int num) { return W(h)->get_coeff(out, num); }
void pdwt_wavelets_set_coeff(void* h, DTYPE* in, int num, int mem_is_on_device) { W(h)->set_coeff(in, num, mem_is_on_device); }
void pdwt_wavelets_print_informations(void* h) { W(h)->print_informations(); }
int pdwt_wavelets_state(void* h) { return (int)W(h)->state; }
void pdwt_wavelets_set_state(void* h, int s) { W(h)->state = (w_state)s; }
void pdwt_wavelets_info(void* h, w_info* out) { *out = W(h)->winfos; }
intptr_t pdwt_wavelets_image_int_ptr(void* h) { return W(h)->image_int_ptr(); }
intptr_t pdwt_wavelets_coeff_int_ptr(void* h, int num) { return W(h)->coeff_int_ptr(num); }
intptr_t pdwt_wavelets_coeffs_table_ptr(void* h)
{
    (void)W(h)->coeff_int_ptr(0);  // raw band pointers leave the class: it stops tracking the coefficients' norm
    return (intptr_t)W(h)->d_coeffs;
}
intptr_t pdwt_wavelets_tmp_int_ptr(void* h) { return (intptr_t)W(h)->d_tmp; }

/* batch of equally sized 2-D images, every level of all images in one launch (include/wt_batch.h: WaveletsImages) */
void* pdwt_images_new(DTYPE* imgs, int B, int Nr, int Nc, const char* wname, int levels, int memisonhost)
{
    return new (std::nothrow) WaveletsImages(imgs, B, Nr, Nc, wname, levels, memisonhost);
}
void* pdwt_images_new_swt(DTYPE* imgs, int B, int Nr, int Nc, const char* wname, int levels, int memisonhost, int do_swt)
{
    return new (std::nothrow) WaveletsImages(imgs, B, Nr, Nc, wname, levels, memisonhost, do_swt);
}
void pdwt_images_delete(void* h) { delete static_cast<WaveletsImages*>(h); }
int pdwt_images_ok(void* h) { return static_cast<WaveletsImages*>(h)->ok() ? 1 : 0; }
int pdwt_images_batched(void* h) { return static_cast<WaveletsImages*>(h)->batched() ? 1 : 0; }
void pdwt_images_forward(void* h) { static_cast<WaveletsImages*>(h)->forward(); }
void pdwt_images_inverse(void* h) { static_cast<WaveletsImages*>(h)->inverse(); }
void* pdwt_images_at(void* h, int b) { return static_cast<WaveletsImages*>(h)->img[(size_t)b]; } /* borrowed: owned by the batch */
}
