// demo.cpp -- drop-in acceptance program: written against include/wt.h exactly the way a user of the
// reference library writes against its src/wt.h (construct, print_informations, forward, get_coeff,
// norm1 / soft_threshold / norm1, set_image(zeros), inverse, get_image -- the call sequence of the
// reference's src/demo.cpp:81,187-217 and README.md:81-104).  Builds with a plain host compiler:
//   g++ -Iinclude examples/demo.cpp -Lpdwt_amd/lib -lpdwt -Wl,-rpath,pdwt_amd/lib -o demo
// Usage: demo <image.dat> <Nr> <Nc> <wavelet> <levels> <out.dat> [do_swt] [ndim]
// Writes: approximation band, thresholded reconstruction, and prints the two L1 norms.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "wt.h"

int main(int argc, char** argv)
{
    if (argc < 7) {
        printf("Usage: %s <image.dat> <Nr> <Nc> <wavelet> <levels> <out.dat> [do_swt] [ndim]\n", argv[0]);
        return 2;
    }
    const int Nr = atoi(argv[2]), Nc = atoi(argv[3]), levels = atoi(argv[5]);
    const int do_swt = argc > 7 ? atoi(argv[7]) : 0, ndim = argc > 8 ? atoi(argv[8]) : 2;
    std::vector<DTYPE> img((size_t)Nr * Nc);
    {
        std::vector<float> raw((size_t)Nr * Nc);
        FILE* f = fopen(argv[1], "rb");
        if (!f || fread(raw.data(), sizeof(float), raw.size(), f) != raw.size()) {
            printf("ERROR: cannot read %s\n", argv[1]);
            return 1;
        }
        fclose(f);
        for (size_t i = 0; i < raw.size(); i++) img[i] = (DTYPE)raw[i];
    }

    Wavelets W(img.data(), Nr, Nc, argv[4], levels, 1, 1, 0, do_swt, ndim);
    if (W.state == W_CREATION_ERROR) return 1;
    W.print_informations();
    W.forward();

    std::vector<DTYPE> band((size_t)Nr * Nc);
    const int nels = W.get_coeff(band.data(), 0);
    printf("approximation band: %d coefficients, first = %.9g\n", nels, (double)band[0]);

    printf("Before threshold : L1 = %.9e\n", (double)W.norm1());
    W.soft_threshold(90.0, 0, 0);
    printf("After threshold : L1 = %.9e\n", (double)W.norm1());

    std::vector<DTYPE> zeros((size_t)Nr * Nc, 0);
    W.set_image(zeros.data(), 0);
    W.inverse();
    std::vector<DTYPE> rec((size_t)Nr * Nc);
    if (W.get_image(rec.data()) != Nr * Nc) return 1;

    FILE* o = fopen(argv[6], "wb");
    if (!o) return 1;
    fwrite(&nels, sizeof(int), 1, o);
    fwrite(band.data(), sizeof(DTYPE), (size_t)nels, o);
    fwrite(rec.data(), sizeof(DTYPE), rec.size(), o);
    fclose(o);
    puts("demo OK");
    return 0;
}
