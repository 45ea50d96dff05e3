/* c_abi_demo.c -- the reference's demo configuration (imageformation.py:99-119: 64 x 64 four-bar mask, quasar source
 * sigma 0.4-0.8, 10-term aberrated pupil, 25 nm pixels, 193 nm) driven through the C ABI of liblitho_abbe.so from PLAIN C:
 * no Python, no torch -- what a binding in any language (cgo, JNI, N-API, ctypes) would call, in the order abbeImage does.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c \
 *       -Llithographysimulator_amd/lib -llitho_abbe -L/opt/rocm/lib -lamdhip64 -lm -o build/c_abi_demo
 *   LD_LIBRARY_PATH=lithographysimulator_amd/lib:/opt/rocm/lib build/c_abi_demo [pn]
 *
 * Prints S, the image size and the image sum: 184 source points, 64 x 64, 2.2029254e13 for the demo (SURVEY.md section 3.1,
 * golden g5).  With an argument (e.g. 1000) it runs a synthetic pn x pn four-bar mask instead -- any even size.
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "litho_abbe.h"

#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)
#define LCK(x) do { int rc_ = (x); if (rc_ != LITHO_OK) { fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, litho_last_error()); return 3; } } while (0)

static uint16_t f32_to_f16(float f)               /* round to nearest even; enough for the demo's coefficients */
{
    union { float f; uint32_t u; } v = {f};
    const uint32_t s = (v.u >> 16) & 0x8000u;
    int32_t e = (int32_t)((v.u >> 23) & 0xFF) - 127 + 15;
    uint32_t m = v.u & 0x7FFFFFu;
    if (e <= 0) {
        if (e < -10) return (uint16_t)s;
        m |= 0x800000u;
        const int sh = 14 - e;
        uint32_t h = m >> sh;
        const uint32_t rem = m & ((1u << sh) - 1), half = 1u << (sh - 1);
        if (rem > half || (rem == half && (h & 1))) ++h;
        return (uint16_t)(s | h);
    }
    if (e >= 31) return (uint16_t)(s | 0x7C00u);
    uint32_t h = ((uint32_t)e << 10) | (m >> 13);
    const uint32_t rem = m & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) ++h;
    return (uint16_t)(s | h);
}

int main(int argc, char **argv)
{
    const int pn = argc > 1 ? atoi(argv[1]) : 64;
    const double wavelength = 193.0, NA = 0.7, pixel = 25.0;
    if (pn < 64 || (pn & 1)) { fprintf(stderr, "pn must be even and >= 64\n"); return 1; }
    printf("liblitho_abbe version %d for %s\n", litho_version(), litho_target_arch());

    /* Mask.calculateEpsilonN (mask.py:63-72) */
    double eps; int N, n_out;
    LCK(litho_epsilon_n(4.0 / pn, pixel, wavelength, &eps, &N));
    LCK(litho_postprocess_size(pn, eps, &n_out));
    int pe;
    LCK(litho_abbe_embedded_size(pn, N, &pe));
    size_t ws_bytes;
    LCK(litho_abbe_workspace_bytes(pn, N, &ws_bytes));

    /* the demo geometry (mask.py:24-27), scaled by pn / 64: four vertical bars */
    int16_t *geo_h = (int16_t *)calloc((size_t)pn * pn, sizeof(int16_t));
    const double k = pn / 64.0;
    for (int y = (int)(9 * k); y < (int)(55 * k); ++y)
        for (int b = 0; b < 4; ++b)
            for (int x = (int)((16 + 9 * b) * k); x < (int)((20 + 9 * b) * k); ++x) geo_h[(size_t)y * pn + x] = 1;

    void *geo, *spec, *bitmap, *shifts, *scratch, *pupil, *raw, *img, *ws;
    const size_t px = (size_t)pn * pn;
    HIPCK(hipMalloc(&geo, px * 2)); HIPCK(hipMalloc(&spec, px * 8)); HIPCK(hipMalloc(&bitmap, px * 8));
    HIPCK(hipMalloc(&shifts, px * 8)); HIPCK(hipMalloc(&scratch, (size_t)(pn + 1) * 4)); HIPCK(hipMalloc(&pupil, px * 8));
    HIPCK(hipMalloc(&raw, px * 4)); HIPCK(hipMalloc(&img, (size_t)n_out * n_out * 4)); HIPCK(hipMalloc(&ws, ws_bytes));
    HIPCK(hipMemcpy(geo, geo_h, px * 2, hipMemcpyHostToDevice));
    HIPCK(hipMemset(raw, 0, px * 4));                      /* the caller zeroes the accumulator */
    hipStream_t st;
    HIPCK(hipStreamCreate(&st));

    /* Mask.fraunhofer -> LightSource.generateQuasar -> argwhere -> Pupil.generatePupilFunction -> the Abbe loop -> post-process */
    LCK(litho_mask_spectrum((const int16_t *)geo, pn, eps, N, spec, ws, ws_bytes, st));
    LCK(litho_source_bitmap(1, 0.4, 0.8, pn, 0.0, 0.0, 4, -3.14159265358979323846 / 8, (int64_t *)bitmap, st));
    int64_t S = 0;
    LCK(litho_source_compact((const int64_t *)bitmap, pn, (int32_t *)shifts, (int64_t)px, (int32_t *)scratch, &S, st));
    const float ab[10] = {0, 0, 0.01f, 0, 100, 0.01f, 0, 0.01f, 0.01f, 0.01f};      /* imageformation.py:100 */
    uint16_t ab16[10];
    for (int i = 0; i < 10; ++i) ab16[i] = f32_to_f16(ab[i]);
    LCK(litho_pupil(ab16, 10, pn, NA, wavelength, 0, NULL, pupil, st));
    LCK(litho_abbe_accumulate(spec, pupil, 1, (const int32_t *)shifts, S, pn, N, (float *)raw, ws, ws_bytes, st));
    LCK(litho_postprocess((const float *)raw, 1, pn, eps, (float *)img, st));
    HIPCK(hipStreamSynchronize(st));

    float *img_h = (float *)malloc((size_t)n_out * n_out * 4);
    HIPCK(hipMemcpy(img_h, img, (size_t)n_out * n_out * 4, hipMemcpyDeviceToHost));
    double sum = 0, mx = 0;
    for (size_t i = 0; i < (size_t)n_out * n_out; ++i) { sum += img_h[i]; if (img_h[i] > mx) mx = img_h[i]; }
    int64_t plan[16];
    char kx[96], ky[96];
    litho_abbe_last_plan(plan);
    litho_abbe_last_kernels(kx, ky, sizeof kx);
    printf("pn %d  N %d  epsilon %.16g  runs at %d  S %lld  image %d x %d  sum %.7e  max %.7e  kernels %s / %s  coarse grid %lld\n",
           pn, N, eps, pe, (long long)S, n_out, n_out, sum, mx, kx, ky, (long long)plan[12]);
    return 0;
}
