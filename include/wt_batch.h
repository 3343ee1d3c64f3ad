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
        : Nr(Nr_), Nc(Nc_)
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
        double acc = 0.0;
        for (size_t s = 0; s < shard.size(); s++)
            if (shard[s]) acc += shard[s]->norm1_end();
        return acc;
    }
    /* the reconstructed batch, shards stacked in order; returns the element count */
    size_t get_image(DTYPE* out)
    {
        size_t n = 0;
        for (size_t s = 0; s < shard.size(); s++)
            if (shard[s]) n += (size_t)shard[s]->get_image(out + (size_t)first[s] * Nc);
        return n;
    }

  private:
    WaveletsBatch(const WaveletsBatch&);
    WaveletsBatch& operator=(const WaveletsBatch&);
};

#endif
