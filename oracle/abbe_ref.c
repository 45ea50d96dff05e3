/*
 * abbe_ref.c -- plain-C restatement of the Abbe hot loop.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product (lithographysimulator_amd) never links or calls it.
 *
 * What it restates (quarterwave0/LithographySimulator):
 *   imageformation.py:63     roll(pupil, (dy,dx))            -> A[i,j] = P[(i-dy) mod pn,(j-dx) mod pn] * M[i,j]
 *   imageformation.py:32-45  pad -> fftshift -> ifft2(norm='forward') -> ifftshift -> crop
 *                            == E[qy,qx] = sum_{iy,ix} A[iy,ix] exp(+2 pi i [(iy-c)(qy-c)+(ix-c)(qx-c)]/N), c = pn/2
 *   imageformation.py:67     image += |E|^2
 * in double precision with a direct separable DFT (O(pn^3) per source point), so it shares
 * no FFT code with either torch or the HIP kernels.  Pinned against tests/golden/g4 and g5
 * by tests/test_oracle_vs_golden.py.
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

static int wrap(int v, int n) { v %= n; return v < 0 ? v + n : v; }

/* field: out[pn*pn*2] (re,im) doubles.  pupil, maskFT: interleaved complex64. */
int oracle_field(const float *pupil, const float *maskFT, int pn, int N, int dy, int dx, double *out)
{
    if (pn <= 0 || N < pn || ((N - pn) & 1)) return -1;
    const int c = pn / 2;
    double *tw = (double *)malloc(sizeof(double) * 2 * (size_t)N);
    double *tmp = (double *)malloc(sizeof(double) * 2 * (size_t)pn * pn);   /* tmp[iy][qx] */
    if (!tw || !tmp) { free(tw); free(tmp); return -2; }
    for (int n = 0; n < N; ++n) {
        tw[2 * n] = cos(2.0 * M_PI * n / N);
        tw[2 * n + 1] = sin(2.0 * M_PI * n / N);
    }
    for (int iy = 0; iy < pn; ++iy) {
        const int py = wrap(iy - dy, pn);
        for (int qx = 0; qx < pn; ++qx) {
            double sr = 0.0, si = 0.0;
            for (int ix = 0; ix < pn; ++ix) {
                const int px = wrap(ix - dx, pn);
                const double pr = pupil[2 * ((size_t)py * pn + px)], pi = pupil[2 * ((size_t)py * pn + px) + 1];
                if (pr == 0.0 && pi == 0.0) continue;
                const double mr = maskFT[2 * ((size_t)iy * pn + ix)], mi = maskFT[2 * ((size_t)iy * pn + ix) + 1];
                const double ar = pr * mr - pi * mi, ai = pr * mi + pi * mr;
                const int t = wrap((ix - c) * (qx - c), N);
                sr += ar * tw[2 * t] - ai * tw[2 * t + 1];
                si += ar * tw[2 * t + 1] + ai * tw[2 * t];
            }
            tmp[2 * ((size_t)iy * pn + qx)] = sr;
            tmp[2 * ((size_t)iy * pn + qx) + 1] = si;
        }
    }
    for (int qy = 0; qy < pn; ++qy)
        for (int qx = 0; qx < pn; ++qx) {
            double sr = 0.0, si = 0.0;
            for (int iy = 0; iy < pn; ++iy) {
                const double ar = tmp[2 * ((size_t)iy * pn + qx)], ai = tmp[2 * ((size_t)iy * pn + qx) + 1];
                const int t = wrap((iy - c) * (qy - c), N);
                sr += ar * tw[2 * t] - ai * tw[2 * t + 1];
                si += ar * tw[2 * t + 1] + ai * tw[2 * t];
            }
            out[2 * ((size_t)qy * pn + qx)] = sr;
            out[2 * ((size_t)qy * pn + qx) + 1] = si;
        }
    free(tw); free(tmp);
    return 0;
}

/* image[pn*pn] += sum_s |E_s|^2 in double, source points in list order (imageformation.py:62-67). */
int oracle_abbe_accumulate(const float *pupil, const float *maskFT, const int32_t *shifts, int S,
                           int pn, int N, double *image)
{
    double *E = (double *)malloc(sizeof(double) * 2 * (size_t)pn * pn);
    if (!E) return -2;
    for (int s = 0; s < S; ++s) {
        int rc = oracle_field(pupil, maskFT, pn, N, shifts[2 * s], shifts[2 * s + 1], E);
        if (rc) { free(E); return rc; }
        for (size_t q = 0; q < (size_t)pn * pn; ++q) image[q] += E[2 * q] * E[2 * q] + E[2 * q + 1] * E[2 * q + 1];
    }
    free(E);
    return 0;
}
