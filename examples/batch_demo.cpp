// batch_demo.cpp -- the batch split from ONE host process (include/wt_batch.h): a batched-1D array is split into row
// blocks, one `Wavelets` per block on the listed devices (round-robin over the visible GPUs), and the sharded run is
// checked against the unsharded one: rows are independent signals, so the reconstruction must be bit-identical and
// the norms must agree to rounding.  Usage: batch_demo <Nr> <Nc> <wavelet> <levels> <nshards>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "wt_batch.h"

int main(int argc, char** argv)
{
    if (argc < 6) {
        printf("Usage: %s <Nr> <Nc> <wavelet> <levels> <nshards>\n", argv[0]);
        return 2;
    }
    const int Nr = atoi(argv[1]), Nc = atoi(argv[2]), levels = atoi(argv[4]), nshards = atoi(argv[5]);
    const int ndev = w_device_count();
    if (ndev < 1) {
        puts("ERROR: no HIP device");
        return 1;
    }
    std::vector<DTYPE> img((size_t)Nr * Nc);
    unsigned s = 12345u;
    for (size_t i = 0; i < img.size(); i++) {
        s = s * 1664525u + 1013904223u;
        img[i] = (DTYPE)((double)(s >> 8) / (double)(1u << 24) - 0.5);
    }
    std::vector<int> devices;
    for (int k = 0; k < nshards; k++) devices.push_back(k % ndev);

    WaveletsBatch B(img.data(), Nr, Nc, argv[3], levels, devices);
    Wavelets W(img.data(), Nr, Nc, argv[3], levels, 1, 1, 0, 0, 1);
    if (!B.ok() || W.state == W_CREATION_ERROR) return 1;
    B.forward();
    W.forward();
    const double nb = B.norm1(), nw = (double)W.norm1();
    const bool rccl = B.last_norm1_used_rccl();  // one shard per device and RCCL loadable: the per-device doubles were all-reduced over RCCL
    B.soft_threshold((DTYPE)0.05);
    W.soft_threshold((DTYPE)0.05);
    const double nbt = B.norm1(), nwt = (double)W.norm1();
    B.inverse();
    W.inverse();
    std::vector<DTYPE> rb(img.size()), rw(img.size());
    const size_t got = B.get_image(rb.data());
    W.get_image(rw.data());
    size_t diff = 0;
    for (size_t i = 0; i < img.size(); i++) diff += (rb[i] != rw[i]);
    const double tol = sizeof(DTYPE) == 4 ? 2e-6 : 1e-12;
    printf("shards %d on %d device(s): norm1 %.9e / %.9e, after threshold %.9e / %.9e, %zu differing samples of %zu; norm1 exchange: %s\n", nshards, ndev, nb, nw,
           nbt, nwt, diff, got, rccl ? "RCCL all-reduce" : "host sum");
    const bool ok = got == img.size() && diff == 0 && fabs(nb - nw) <= tol * nw && fabs(nbt - nwt) <= tol * nwt && nbt < nb;
    puts(ok ? "batch OK" : "batch MISMATCH");
    return ok ? 0 : 1;
}
