/*
 * wt_batch.h -- batch split of the PDWT hot path over the GPUs of one node, from ONE host process (header-only, on top of
 * wt.h).  The reference is single-GPU (TODO.txt:15); BASELINE.json's north star shards batched inputs over the 8 GPUs.
 *
 * Rows of a batched-1D array are independent signals (reference src/separable.cu:213) and whole 2-D images are
 * independent, so the batch shards with NO data-path collective: shard s owns a contiguous block of rows (or a whole
 * image) and a private `Wavelets` instance on device devices[s].  Every call below just issues the same method on every
 * shard in turn: launches are asynchronous, so the devices work concurrently; the methods that return a value to the host
 * (norm1, get_image) are the only synchronisation points.  norm1() = sum of the per-shard partial sums (each reduced
 * in double on its GPU) -- the single-process counterpart of the one-double all-reduce that pdwt_amd/batch.py does
 * over RCCL when the shards belong to different processes.
 */
#ifndef WT_BATCH_H
#define WT_BATCH_H

#include <stddef.h>

#include <vector>

#include "wt.h"
#include "pdwt_hip.h" /* plain C: the batched-2D entry points and the filter lookup (no device toolkit header) */

/* rows [first, first + count) of shard `shard` out of `nshards` (the first n_rows % nshards shards get one more) */
inline void w_shard_rows(int n_rows, int nshards, int shard, int* first, int* count)
{
    const int base = n_rows / nshards, rem = n_rows % nshards;
    *first = shard * base + (shard < rem ? shard : rem);
    *count = base + (shard < rem ? 1 : 0);
}

class WaveletsBatch {
  public:
    std::vector<Wavelets*> shard;   /* one instance per shard, on devices[s] */
    std::vector<int> first, count;  /* row block of each shard */
    int Nr, Nc;

    /* batched 1-D: `img` is an Nr x Nc host array of Nr signals; devices = device index of every shard (repeats allowed) */
    WaveletsBatch(DTYPE* img, int Nr_, int Nc_, const char* wname, int levels, const std::vector<int>& devices, int do_swt = 0)
        : Nr(Nr_), Nc(Nc_), used_rccl_(false), devs_(devices)
    {
        const int n = (int)devices.size();
        const int prev = w_get_device();
        for (int s = 0; s < n; s++) {
            int f, c;
            w_shard_rows(Nr, n, s, &f, &c);
            first.push_back(f);
            count.push_back(c);
            w_set_device(devices[s]);
            shard.push_back(c > 0 ? new Wavelets(img + (size_t)f * Nc, c, Nc, wname, levels, 1, 1, 0, do_swt, 1) : NULL);
        }
        if (prev >= 0) w_set_device(prev);
    }
    ~WaveletsBatch()
    {
        for (size_t s = 0; s < shard.size(); s++) delete shard[s];
    }
    bool ok() const
    {
        for (size_t s = 0; s < shard.size(); s++)
            if (shard[s] && shard[s]->state == W_CREATION_ERROR) return false;
        return !shard.empty();
    }
    void forward()
    {
        for (size_t s = 0; s < shard.size(); s++)
            if (shard[s]) shard[s]->forward();
    }
    void inverse()
    {
        for (size_t s = 0; s < shard.size(); s++)
            if (shard[s]) shard[s]->inverse();
    }
    void soft_threshold(DTYPE beta, int do_thresh_appcoeffs = 0, int normalize = 0)
    {
        for (size_t s = 0; s < shard.size(); s++)
            if (shard[s]) shard[s]->soft_threshold(beta, do_thresh_appcoeffs, normalize);
    }
    /* every shard's reduction is enqueued on its device before the first result is read (the devices reduce concurrently),
     * and the per-shard sums are added as doubles, before any rounding to DTYPE -- the value pdwt_amd/batch.py's all-reduce gives */
    double norm1()
    {
        for (size_t s = 0; s < shard.size(); s++)
            if (shard[s]) shard[s]->norm1_begin();
        /* one shard per device (the deployment this class exists for): the per-device doubles are all-reduced over RCCL (xGMI), grouped,
         * on the devices' streams -- every device then holds the batch norm, one of them hands it to the host.  Anything else (several
         * shards on one device, RCCL not loadable): the doubles are added here. */
        double sum = 0.0;
        if (rccl_norm1(&sum)) {
            for (size_t s = 0; s < shard.size(); s++)
                if (shard[s]) (void)shard[s]->norm1_end(); /* (drains each shard's own reduction; its per-shard value stays intact) */
            used_rccl_ = true;
            return sum;
        }
        used_rccl_ = false;
        double acc = 0.0;
        for (size_t s = 0; s < shard.size(); s++)
            if (shard[s]) acc += shard[s]->norm1_end();
        return acc;
    }
    bool last_norm1_used_rccl() const { return used_rccl_; }
    /* the reconstructed batch, shards stacked in order; returns the element count */
    size_t get_image(DTYPE* out)
    {
        size_t n = 0;
        for (size_t s = 0; s < shard.size(); s++)
            if (shard[s]) n += (size_t)shard[s]->get_image(out + (size_t)first[s] * Nc);
        return n;
    }

  private:
    bool used_rccl_;
    std::vector<int> devs_;
    bool rccl_norm1(double* out)
    {
        if (!pdwt_rccl_available() || shard.empty() || devs_.size() != shard.size()) return false;
        std::vector<int> dv;
        std::vector<const double*> in;
        std::vector<double*> o;
        const size_t ri = pdwt_sum_result_index(), si = pdwt_sum_spare_index();
        for (size_t s = 0; s < shard.size(); s++) {
            if (!shard[s]) return false; /* (an empty shard has no device buffer to contribute: host sum) */
            for (size_t t = 0; t < dv.size(); t++)
                if (dv[t] == devs_[s]) return false;
            /* a shard whose norm1_begin() failed (or found no scratch) holds a STALE double: host sum (norm1_end() falls back to norm1()) */
            if (!shard[s]->norm1_pending()) return false;
            double* sc = (double*)shard[s]->norm1_scratch_int_ptr();
            if (!sc) return false;
            dv.push_back(devs_[s]);
            in.push_back(sc + ri);
            o.push_back(sc + si); /* a spare double of the same buffer: the shard's own sum stays where norm1_end() reads it */
        }
        return pdwt_rccl_allreduce_sum_f64((int)dv.size(), dv.data(), in.data(), o.data(), out) == 0;
    }
    WaveletsBatch(const WaveletsBatch&);
    WaveletsBatch& operator=(const WaveletsBatch&);
};

/*
 * A batch of B equally sized 2-D images on ONE device, transformed together: every level of all images in one launch
 * (pdwt_batch2d_*, include/pdwt_hip.h) when the geometry allows -- small images are launch-bound, six launches per pair whatever
 * the size -- otherwise image after image.  Each image is an ordinary `Wavelets` instance (it owns image, bands and scratch:
 * img[b]->get_coeff(...), soft_threshold(...) etc. work as usual between forward() and inverse()); results are those of the
 * per-image transforms bit for bit.  Both builds (round 5: libpdwtd batches through the fused double-precision level kernels).
 */
#ifdef DOUBLEPRECISION
#define WB_FILTERS pdwt_filters_f64
#define WB_COMPUTE_FILTERS pdwt_compute_filters_separable_f64
#define WB_CREATE pdwt_batch2d_create_f64
#define WB_FORWARD pdwt_batch2d_forward_f64
#define WB_INVERSE pdwt_batch2d_inverse_f64
#define WB_DESTROY pdwt_batch2d_destroy_f64
#else
#define WB_FILTERS pdwt_filters_f32
#define WB_COMPUTE_FILTERS pdwt_compute_filters_separable_f32
#define WB_CREATE pdwt_batch2d_create_f32
#define WB_FORWARD pdwt_batch2d_forward_f32
#define WB_INVERSE pdwt_batch2d_inverse_f32
#define WB_DESTROY pdwt_batch2d_destroy
#endif
class WaveletsImages {
  public:
    std::vector<Wavelets*> img;
    int Nr, Nc;

    /* imgs: B contiguous Nr x Nc images (host, or device when memisonhost = 0); do_swt = 1: the undecimated transform (one launch per level
     * over all images in the float build when every level is inside the fused SWT level kernels; image after image otherwise) */
    WaveletsImages(DTYPE* imgs, int B, int Nr_, int Nc_, const char* wname, int levels, int memisonhost = 1, int do_swt = 0) : Nr(Nr_), Nc(Nc_), batch_(NULL)
    {
        for (int b = 0; b < B; b++) {
            img.push_back(new Wavelets(imgs + (size_t)b * Nr * Nc, Nr, Nc, wname, levels, memisonhost, 1, 0, do_swt, 2));
            /* the batched launches write the bands behind the instances' backs: no norm kept from a threshold pass, whatever
             * set_norm_cache() / PDWT_NORM_IN_THRESHOLD say later (taking a raw band pointer switches that shortcut off for good) */
            img.back()->set_norm_cache(0);
            if (img.back()->state != W_CREATION_ERROR) (void)img.back()->coeff_int_ptr(0);
        }
        if (ok()) {
            std::vector<DTYPE*> di, dt;
            std::vector<DTYPE**> dc;
            for (int b = 0; b < B; b++) {
                di.push_back(img[b]->d_image);
                dc.push_back(img[b]->d_coeffs);
                dt.push_back(img[b]->d_tmp);
            }
            const w_info w = img[0]->winfos;
            pdwt_info info = {w.ndims, w.Nr, w.Nc, w.nlevels, w.do_swt, w.hlen};
#ifdef DOUBLEPRECISION
            const bool kind_ok = !w.do_swt; /* (the double build batches the decimated transform and Haar) */
#else
            const bool kind_ok = true;
#endif
            if (kind_ok && WB_COMPUTE_FILTERS(wname, w.do_swt, &bank_) == w.hlen) batch_ = WB_CREATE(B, di.data(), dc.data(), dt.data(), info);
        }
    }
    ~WaveletsImages()
    {
        if (batch_) WB_DESTROY(batch_);
        for (size_t b = 0; b < img.size(); b++) delete img[b];
    }
    bool ok() const
    {
        for (size_t b = 0; b < img.size(); b++)
            if (img[b]->state == W_CREATION_ERROR) return false;
        return !img.empty();
    }
    bool batched() const { return batch_ != NULL; } /* one launch per level over all images */
    /* the one-launch form runs the bank of `wname` on every image: not for a batch in which set_filters_*() changed a member's bank */
    bool named_bank() const
    {
        for (size_t b = 0; b < img.size(); b++)
            if (img[b]->custom_filters()) return false;
        return true;
    }
    void forward()
    {
        if (batch_ && named_bank() && WB_FORWARD(batch_, &bank_) == 0) {
            for (size_t b = 0; b < img.size(); b++) img[b]->state = W_FORWARD;
            return;
        }
        for (size_t b = 0; b < img.size(); b++) img[b]->forward();
    }
    void inverse()
    {
        bool all_fwd = true;
        for (size_t b = 0; b < img.size(); b++) all_fwd = all_fwd && img[b]->state != W_INVERSE && img[b]->state != W_CREATION_ERROR;
        if (batch_ && all_fwd && named_bank() && WB_INVERSE(batch_, &bank_) == 0) {
            for (size_t b = 0; b < img.size(); b++) img[b]->state = W_INVERSE;
            return;
        }
        for (size_t b = 0; b < img.size(); b++) img[b]->inverse();
    }
    /* all images, stacked; returns the element count */
    size_t get_image(DTYPE* out)
    {
        size_t n = 0;
        for (size_t b = 0; b < img.size(); b++) n += (size_t)img[b]->get_image(out + b * (size_t)Nr * Nc);
        return n;
    }

  private:
    void* batch_;
    WB_FILTERS bank_;
    WaveletsImages(const WaveletsImages&);
    WaveletsImages& operator=(const WaveletsImages&);
};

#endif
