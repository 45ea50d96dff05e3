// wave_fft.hpp -- length-4096 transform computed by ONE wavefront (64 lanes x 64 points).
//
//   X[m + 64 k2] = sum_l w64^(l k2) [ w4096^(l m) sum_j x[l + 64 j] w64^(j m) ]
//
// Lane l holds x[l + 64 j] in register slot j.  Pass A is a 64-point DFT over the slots, then
// the lane twiddle w4096^(l m), then a 64 x 64 transpose THROUGH LDS THAT ONLY THIS WAVE
// TOUCHES (no workgroup barrier: a wave's LDS operations complete in order), then pass B, a
// second 64-point DFT over the slots.  Lane m ends with X[m + 64 k2] in slot k2: the same
// "lane + 64 * slot" ownership as on input.  Compared with the 256-thread radix-16 engine this
// halves the LDS traffic and removes every s_barrier, at ~8 % more VALU work.
//
// The 64-point DFTs are in-place radix-2 decimation-in-frequency networks: natural order in,
// BIT-REVERSED slot order out (slot br6(k) holds bin k); all indices are compile-time, so
// zero inputs fold away and unused outputs are dead code.
#pragma once
#include <hip/hip_runtime.h>

#include "fft_core.hpp"

namespace litho {

static constexpr double W64C[33] = {
    1, 0.99518472667219693, 0.98078528040323043, 0.95694033573220882,
    0.92387953251128674, 0.88192126434835505, 0.83146961230254524, 0.77301045336273699,
    0.70710678118654757, 0.63439328416364549, 0.55557023301960229, 0.47139673682599781,
    0.38268343236508984, 0.29028467725446233, 0.19509032201612833, 0.09801714032956077,
    0.0, -0.098017140329560645, -0.19509032201612819, -0.29028467725446216,
    -0.38268343236508973, -0.4713967368259977, -0.55557023301960196, -0.63439328416364538,
    -0.70710678118654746, -0.77301045336273699, -0.83146961230254535, -0.88192126434835494,
    -0.92387953251128674, -0.95694033573220882, -0.98078528040323043, -0.99518472667219682,
    -1};

__host__ __device__ constexpr int br6(int v)
{
    return ((v & 1) << 5) | ((v & 2) << 3) | ((v & 4) << 1) | ((v & 8) >> 1) | ((v & 16) >> 3) | ((v & 32) >> 5);
}

// v * exp(+2 pi i M / 64), 0 <= M < 32
template <int M>
__device__ __forceinline__ float2 mul_root64(float2 v)
{
    if constexpr (M == 0) {
        return v;
    } else if constexpr (M == 16) {
        return make_float2(-v.y, v.x);
    } else if constexpr (M == 8) {
        constexpr float h = 0.70710678118654752f;
        return make_float2((v.x - v.y) * h, (v.x + v.y) * h);
    } else if constexpr (M == 24) {
        constexpr float h = 0.70710678118654752f;
        return make_float2(-(v.x + v.y) * h, (v.x - v.y) * h);
    } else {
        constexpr float wr = (float)W64C[M];
        constexpr float wi = (float)(M <= 16 ? W64C[16 - M] : W64C[M - 16]);     // sin(2 pi M/64) = cos(2 pi (16-M)/64)
        return make_float2(fmaf(v.x, wr, -v.y * wi), fmaf(v.x, wi, v.y * wr));
    }
}

// In-place 64-point DIF: x[br6(k)] <- sum_j x[j] exp(+2 pi i j k / 64)
__device__ __forceinline__ void dft64_dif(float2 (&x)[64])
{
    static_for<0, 6>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        constexpr int half = 32 >> s;                 // butterfly span
        constexpr int stride = 1 << s;                // twiddle step in 64ths
        static_for<0, 32>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            constexpr int grp = b / half, i = b % half;
            constexpr int lo = grp * 2 * half + i, hi = lo + half;
            const float2 a = x[lo], c = x[hi];
            x[lo] = cadd(a, c);
            x[hi] = mul_root64<(i * stride) % 32>(csub(a, c));
        });
    });
}

struct Wave4096 {
    static constexpr int LDS_FLOATS = 64 * 65;       // one padded 64 x 64 fp32 matrix per wave (16.6 KB)

    // Lane twiddles w4096^(lane*m), m = 8a + b, factored as w8[a]*w1[b]: 16 factors per lane, in registers
    // (an LDS-resident table was measured slower: 9.15 vs 8.06 us/point).
    struct LaneTwiddles {
        float2 row[16];         // [0..7] = w4096^(l b), [8..15] = w4096^(8 l a)
    };
    __device__ static __forceinline__ void load_lane_twiddles(LaneTwiddles& t, const float2* __restrict__ table, int lane)
    {
        static_for<0, 8>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            t.row[i] = table[lane * i];
            t.row[8 + i] = table[lane * 8 * i];
        });
    }

    // x: slot j = sample lane + 64 j (natural).  On return slot br6(k2) = bin lane + 64 k2.
    __device__ static __forceinline__ void run(float2 (&x)[64], const LaneTwiddles& tw, float* lds, int lane)
    {
        dft64_dif(x);                                  // slot br6(m) = y[m]
        // lane twiddle w4096^(lane*m), m = 8 a + b
        static_for<0, 64>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            constexpr int a = m >> 3, b = m & 7, sl = br6(m);
            if constexpr (b != 0) x[sl] = cmul(x[sl], tw.row[b]);
            if constexpr (a != 0) x[sl] = cmul(x[sl], tw.row[8 + a]);
        });
        // 64 x 64 transpose, real parts then imaginary parts, through this wave's private matrix:
        // element (row = writer lane, col = m) ; reader lane m takes column m.
        float* const wr = lds + lane * 65;
        float* const rd = lds + lane;
        static_for<0, 64>([&](auto m_) { constexpr int m = decltype(m_)::value; wr[m] = x[br6(m)].x; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float re[64];
        static_for<0, 64>([&](auto l_) { constexpr int l = decltype(l_)::value; re[l] = rd[l * 65]; });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_for<0, 64>([&](auto m_) { constexpr int m = decltype(m_)::value; wr[m] = x[br6(m)].y; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_for<0, 64>([&](auto l_) {
            constexpr int l = decltype(l_)::value;
            x[l] = make_float2(re[l], rd[l * 65]);
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        dft64_dif(x);                                  // slot br6(k2) = X[lane + 64 k2]
    }
};

}  // namespace litho
