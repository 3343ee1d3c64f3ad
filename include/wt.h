/*
 * wt.h -- the `Wavelets` class of PDWT, MI355X-native build (drop-in for the reference's src/wt.h).
 *
 * Same public surface as the reference class (src/wt.h:20-76): same constructor signature and
 * defaults, same method names / argument meaning / return values, same public data members in the
 * same order (src/wt.h:24-34), same state enum (src/wt.h:8-17), precision chosen at compile time by
 * -DDOUBLEPRECISION exactly like the reference (src/filters.h:16-30) -> libpdwt.so / libpdwtd.so.
 * A program written against the reference header (e.g. its src/demo.cpp) recompiles unchanged.
 *
 * Unlike the reference header this one pulls in NO device toolkit header (the reference reaches
 * <cublas.h> through utils.h -> filters.h): all device work goes through the C-ABI of
 * include/pdwt_hip.h, implemented by hand-written gfx950 kernels in libpdwt_hip.so.
 */
#ifndef WT_H
#define WT_H

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#ifndef DOUBLEPRECISION
#define DTYPE float
#else
#define DTYPE double
#endif

#ifndef PDWT_UINT_DEFINED
#define PDWT_UINT_DEFINED
typedef unsigned int uint;
#endif

/* geometry + transform description, passed by value to every driver (reference src/utils.h:9-19) */
struct w_info {
    int ndims;   /* 2 = image, 1 = (batched) 1D signals stored as rows */
    int Nr;      /* rows  (1D: batch size) */
    int Nc;      /* columns (1D: samples)  */
    int nlevels; /* decomposition levels (after clamping to the maximum the size allows) */
    int do_swt;  /* stationary (undecimated) transform */
    int hlen;    /* filter length */
};

/* size helpers of the reference's utils (src/utils.cu:4-34), kept because callers of wt.h use them */
int w_iDivUp(int a, int b);
int w_ipow2(int a);
int w_ilog2(int i);
void w_div2(int* N);
void w_swap_ptr(DTYPE** a, DTYPE** b);

/* Devices (the reference has no multi-GPU support, TODO.txt:15).  An instance lives on the device that is current when it
 * is constructed -- select it with w_set_device() first -- and every method switches to that device for its duration, so
 * one host thread can drive instances on several GPUs in turn (see wt_batch.h for the batch split built on this). */
int w_set_device(int dev);   /* 0 on success */
int w_get_device(void);
int w_device_count(void);

/* life cycle of an instance; guards e.g. a second inverse() (band 0 is consumed by the first) */
typedef enum w_state {
    W_INIT,            /* constructed / image replaced, coefficients not computed */
    W_FORWARD,         /* forward() done, coefficients valid */
    W_INVERSE,         /* inverse() done: d_image rewritten, coefficients no longer meaningful */
    W_THRESHOLD,       /* coefficients modified (never set by the reference either) */
    W_CREATION_ERROR,  /* constructor failed (unknown wavelet, bad ndim, allocation) */
    W_FORWARD_ERROR,   /* a forward launch failed */
    W_INVERSE_ERROR,   /* an inverse launch failed */
    W_THRESHOLD_ERROR  /* a threshold launch failed */
} w_state;

class Wavelets {
  public:
    /* data members: order and types of src/wt.h:24-34 */
    DTYPE* d_image;        /* device: input image / reconstruction */
    DTYPE** d_coeffs;      /* host array of device pointers: [A, H1,V1,D1, ...] or [A, D1, ...] */
    DTYPE* d_tmp;          /* device scratch */
    int current_shift_r;
    int current_shift_c;
    char wname[128];
    int do_separable;
    int do_cycle_spinning;
    w_info winfos;
    w_state state;

    Wavelets();
    Wavelets(DTYPE* img, int Nr, int Nc, const char* wname, int levels, int memisonhost = 1, int do_separable = 1,
             int do_cycle_spinning = 0, int do_swt = 0, int ndim = 2);
    Wavelets(const Wavelets& W);
    ~Wavelets();

    void forward();
    void soft_threshold(DTYPE beta, int do_thresh_appcoeffs = 0, int normalize = 0);
    void hard_threshold(DTYPE beta, int do_thresh_appcoeffs = 0, int normalize = 0);
    void group_soft_threshold(DTYPE beta, int do_thresh_appcoeffs = 0, int normalize = 0);
    void shrink(DTYPE beta, int do_thresh_appcoeffs = 1);
    void proj_linf(DTYPE beta, int do_thresh_appcoeffs = 1);
    void circshift(int sr, int sc, int inplace = 1);
    void inverse();
    DTYPE norm2sq();
    DTYPE norm1();
    int get_image(DTYPE* img);
    void print_informations();
    int get_coeff(DTYPE* coeff, int num);
    void set_image(DTYPE* img, int mem_is_on_device = 0);
    void set_coeff(DTYPE* coeff, int num, int mem_is_on_device = 0);
    int set_filters_forward(char* filtername, uint len, DTYPE* filter1, DTYPE* filter2, DTYPE* filter3 = NULL, DTYPE* filter4 = NULL);
    int set_filters_inverse(DTYPE* filter1, DTYPE* filter2, DTYPE* filter3 = NULL, DTYPE* filter4 = NULL);
    int add_wavelet(Wavelets W, DTYPE alpha = 1.0f);
    intptr_t image_int_ptr(void);
    intptr_t coeff_int_ptr(int num);

    /* ADDITION (not in the reference).  By default norm1() reduces the bands every time it is called, exactly like the
     * reference (src/wt.cu:398-418).  set_norm_cache(1) lets soft_threshold() accumulate sum|c| of the values it writes in
     * the same pass and the norm1() that follows return it without touching the bands again.  Only opt in when nothing but
     * the methods of this class writes the bands: a kernel or copy of the caller's that writes through d_coeffs[k] is
     * invisible to the class (coeff_int_ptr() / set_coeff() are not -- they switch the shortcut off).  INTEGRATION.md B. */
    void set_norm_cache(int on = 1);
    /* ADDITIONS: norm1() in two halves, the value in double.  norm1_begin() only enqueues the reduction on the instance's
     * device, norm1_end() waits for it and returns the sum before its rounding to DTYPE; wt_batch.h starts every shard's
     * reduction before it reads the first one and adds the doubles. */
    void norm1_begin();
    double norm1_end();
    /* ADDITION: 1 while the double at pdwt_sum_result_index() of the scratch below is (or, stream-ordered, will be) this instance's
     * CURRENT sum|c| -- norm1_begin() enqueued its reduction, or the one-pass threshold's value is still valid; 0 when norm1_begin()
     * failed or was never called: wt_batch.h then adds the shards' sums on the host instead of all-reducing a stale double. */
    int norm1_pending() const;
    /* ADDITION: device address of the instance's reduction scratch (allocated on first use; pdwt_sum_scratch_doubles() doubles on the
     * instance's device): after norm1_begin() the double at element pdwt_sum_result_index() is this instance's sum|c| -- what
     * wt_batch.h hands to the RCCL all-reduce across the devices of a batch.  0 on failure. */
    intptr_t norm1_scratch_int_ptr(void);
    /* ADDITION: 1 once set_filters_forward() / set_filters_inverse() has replaced the bank of `wname` (wt_batch.h: a batch whose
     * members do not all run the named bank any more is transformed image by image) */
    int custom_filters() const;

  private:
    /* per-instance filter bank (the reference keeps it in process-global constant memory, so two
     * live instances silently share the last one's taps -- SURVEY.md Appendix B-1).  Appended after
     * the public members so their offsets match the reference layout. */
    void* filters_;
    Wavelets& operator=(const Wavelets&); /* "do not use" in the reference (src/wt.h:49-51) */
};

#endif
