// wave_fft.hpp -- S*S-point transforms computed inside ONE wavefront, S = 64 (one 4096-point line
// per wave) or S = 32 (two 1024-point lines per wave, one per half-wave).
//
//   X[m + S k2] = sum_l wS^(l k2) [ w_{S*S}^(l m) sum_j x[l + S j] wS^(j m) ]
//
// Lane l of a line holds x[l + S j] in register slot j.  Pass A is an S-point DFT over the slots,
// then the lane twiddle w_{S*S}^(l m), then an S x S transpose THROUGH LDS THAT ONLY THIS WAVE TOUCHES
// (no workgroup barrier: a wave's LDS operations complete in order), then pass B, a second S-point
// DFT over the slots.  Lane m ends with X[m + S k2] in slot k2: the same "lane + S * slot"
// ownership as on input.  Compared with the workgroup-wide radix-16 engine this halves the LDS
// traffic and removes every s_barrier, at ~8 % more VALU work.
//
// The S-point DFTs are in-place radix-2 decimation-in-frequency networks: natural order in,
// BIT-REVERSED slot order out (slot brev(k) holds bin k); all indices are compile-time, so zero
// inputs fold away and unused outputs are dead code.
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

// v * exp(+2 pi i M / 128), 0 <= M < 64 (compile-time constants; trivial cases folded)
template <int M>
__device__ __forceinline__ float2 mul_root128(float2 v)
{
    if constexpr (M % 2 == 0) {
        return mul_root64<M / 2>(v);
    } else {
        constexpr double ang = 6.283185307179586476925 * M / 128.0;
        const float wr = (float)__builtin_cos(ang), wi = (float)__builtin_sin(ang);      // folded at compile time
        return make_float2(fmaf(v.x, wr, -v.y * wi), fmaf(v.x, wi, v.y * wr));
    }
}

// In-place P-point radix-2 DIF network (P = 2^LP) on x[OFF .. OFF + P): x[OFF + brev_LP(k)] <- sum_j x[OFF + j] w_P^(j k)
template <int LP, int OFF, int TOTAL>
__device__ __forceinline__ void dif_network(float2 (&x)[TOTAL])
{
    constexpr int P = 1 << LP;
    static_for<0, LP>([&](auto s_) {
        constexpr int s = decltype(s_)::value;
        constexpr int half = (P / 2) >> s;            // butterfly span
        constexpr int stride = (64 / P) << s;         // twiddle step in 64ths
        static_for<0, P / 2>([&](auto b_) {
            constexpr int b = decltype(b_)::value;
            constexpr int grp = b / half, i = b % half;
            constexpr int lo = OFF + grp * 2 * half + i, hi = lo + half;
            const float2 a = x[lo], c = x[hi];
            x[lo] = cadd(a, c);
            x[hi] = mul_root64<(i * stride) % 32>(csub(a, c));
        });
    });
}
__host__ __device__ constexpr int brev_bits(int v, int bits)
{
    int r = 0;
    for (int b = 0; b < bits; ++b) r |= ((v >> b) & 1) << (bits - 1 - b);
    return r;
}

template <int LS>
struct WaveSq {
    static_assert(LS == 5 || LS == 6, "S = 32 or 64");
    static constexpr int S = 1 << LS;                 // lanes per line = slots per lane
    static constexpr int LINES = 64 / S;              // lines per wave
    // Row stride of the transpose matrix in floats.  S + 1 (odd: column reads and row writes conflict-free with 4-byte ops) -- or,
    // build-time experiment LITHO_TSTRIDE=68 for S = 64: a multiple of 4, so that a lane writes its row with 16-byte LDS stores
    // (ds_write_b128, 16 per phase instead of 32 ds_write2_b32; a 16-lane group covers all 64 banks) while the column reads stay
    // conflict-free (bank = (4 r + lane) mod 64 for a fixed row r).
#ifndef LITHO_TSTRIDE
#define LITHO_TSTRIDE 0
#endif
    static constexpr int TS = (LS == 6 && LITHO_TSTRIDE) ? LITHO_TSTRIDE : S + 1;
    static constexpr int LDS_FLOATS = 64 * TS;        // LINES padded S x S fp32 matrices per wave
    // one row of the matrix: S floats from consecutive compile-time slots, 4 at a time when the stride allows aligned 16-byte stores
    template <typename Get>
    __device__ static __forceinline__ void write_row(float* wr, Get&& get)
    {
        if constexpr (TS % 4 == 0) {
            static_for<0, S / 4>([&](auto k_) {
                constexpr int k = decltype(k_)::value;
                *reinterpret_cast<float4*>(wr + 4 * k) = make_float4(get(std::integral_constant<int, 4 * k>{}), get(std::integral_constant<int, 4 * k + 1>{}),
                                                                      get(std::integral_constant<int, 4 * k + 2>{}), get(std::integral_constant<int, 4 * k + 3>{}));
            });
        } else {
            static_for<0, S>([&](auto m_) { wr[decltype(m_)::value] = get(m_); });
        }
    }
    static constexpr int NTW = 8 + S / 8;             // lane twiddle factors

    __host__ __device__ static constexpr int brev(int v)
    {
        int r = 0;
        for (int b = 0; b < LS; ++b) r |= ((v >> b) & 1) << (LS - 1 - b);
        return r;
    }

    // In-place S-point DIF: x[brev(k)] <- sum_j x[j] exp(+2 pi i j k / S)
    __device__ static __forceinline__ void dft_dif(float2 (&x)[S])
    {
        static_for<0, LS>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            constexpr int half = (S / 2) >> s;            // butterfly span
            constexpr int stride = (64 / S) << s;         // twiddle step in 64ths
            static_for<0, S / 2>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                constexpr int grp = b / half, i = b % half;
                constexpr int lo = grp * 2 * half + i, hi = lo + half;
                const float2 a = x[lo], c = x[hi];
                x[lo] = cadd(a, c);
                x[hi] = mul_root64<(i * stride) % 32>(csub(a, c));
            });
        });
    }

    // dft_dif with a hook after every butterfly stage (a caller that spreads memory instructions over the transform)
    template <typename Hook>
    __device__ static __forceinline__ void dft_dif_hook(float2 (&x)[S], Hook&& after_stage)
    {
        static_for<0, LS>([&](auto s_) {
            constexpr int s = decltype(s_)::value;
            constexpr int half = (S / 2) >> s;
            constexpr int stride = (64 / S) << s;
            static_for<0, S / 2>([&](auto b_) {
                constexpr int b = decltype(b_)::value;
                constexpr int grp = b / half, i = b % half;
                constexpr int lo = grp * 2 * half + i, hi = lo + half;
                const float2 a = x[lo], c = x[hi];
                x[lo] = cadd(a, c);
                x[hi] = mul_root64<(i * stride) % 32>(csub(a, c));
            });
            after_stage(s_);
        });
    }

    // Lane twiddles w_{S*S}^(l*m), m = 8a + b, factored as w8[a]*w1[b], in registers (an LDS-resident table
    // was measured slower: 9.15 vs 8.06 us/point at 2048^2).
    struct LaneTwiddles {
        float2 row[NTW];        // [0..7] = w^(l b), [8..] = w^(8 l a)
    };
    // table[n] = exp(2 pi i n / Ntab) with Ntab = S*S*step, so entry step*e is w_{S*S}^e.  l = lane within its line.
    __device__ static __forceinline__ void load_lane_twiddles(LaneTwiddles& t, const float2* __restrict__ table, int l, int step)
    {
        static_for<0, 8>([&](auto i_) { constexpr int i = decltype(i_)::value; t.row[i] = table[step * l * i]; });
        static_for<0, S / 8>([&](auto i_) { constexpr int i = decltype(i_)::value; t.row[8 + i] = table[step * l * 8 * i]; });
    }

    // Lane twiddles in a workgroup-shared LDS table instead of registers: entry (i, lane) at tab[i * 64 + lane]
    // (NTW x 64 float2 = 8 KB for S = 64).  They are then live only during the twiddle stage: the full-output kernels
    // (64 accumulators per lane) need those 2 NTW registers.
    static constexpr int TW_LDS_FLOAT2 = NTW * 64;
    __device__ static __forceinline__ void fill_lane_twiddle_table(float2* tab, const float2* __restrict__ table,
                                                                   int step, int tid, int nthreads)
    {
        for (int i = tid; i < TW_LDS_FLOAT2; i += nthreads) {
            const int e = i >> 6, l = i & (S - 1);
            tab[i] = table[e < 8 ? step * l * e : step * l * 8 * (e - 8)];
        }
    }
    // run() with the lane twiddles read from such a table (the caller synchronises the workgroup after filling it)
    template <bool TABLE_FIRST = false>
    __device__ static __forceinline__ void run_lds_tw(float2 (&x)[S], const float2* tab, float* lds, int lane)
    {
        if constexpr (TABLE_FIRST) {                   // (a caller whose inputs are still in flight from memory: the reads overlap the wait)
            LaneTwiddles tw;
            asm volatile("" ::: "memory");
            static_for<1, NTW>([&](auto i_) { constexpr int i = decltype(i_)::value; if constexpr (i != 8) tw.row[i] = tab[i * 64 + lane]; });
            run(x, tw, lds, lane);
            return;
        }
        // Pass A first, the table reads after it: 30 twiddle registers on top of the 128 of x and the caller's 64
        // accumulators during pass A are what made the callers spill (k_ypass_coop: 18 accumulators per line, to HBM).
        dft_dif(x);                                    // slot brev(m) = y[m]
        LaneTwiddles tw;
        asm volatile("" ::: "memory");                 // the table reads stay here: inside the caller's loop, behind pass A
        static_for<1, NTW>([&](auto i_) { constexpr int i = decltype(i_)::value; if constexpr (i != 8) tw.row[i] = tab[i * 64 + lane]; });
        run_tail(x, tw, lds, lane);
    }

    // x: slot j = sample l + S j (natural).  On return slot brev(k2) = bin l + S k2.
    // lds: this wave's LDS_FLOATS floats; lane = 0..63.
    __device__ static __forceinline__ void run(float2 (&x)[S], const LaneTwiddles& tw, float* lds, int lane)
    {
        dft_dif(x);                                    // slot brev(m) = y[m]
        run_tail(x, tw, lds, lane);
    }
    // the lane twiddles w_{S*S}^(l m) on the outputs of pass A (slot brev(m) = y[m])
    __device__ static __forceinline__ void lane_twiddle_mul(float2 (&x)[S], const LaneTwiddles& tw)
    {
        static_for<0, S>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            constexpr int a = m >> 3, b = m & 7, sl = brev(m);
            if constexpr (b != 0) x[sl] = cmul(x[sl], tw.row[b]);
            if constexpr (a != 0) x[sl] = cmul(x[sl], tw.row[8 + a]);
        });
    }
    // everything after pass A
    __device__ static __forceinline__ void run_tail(float2 (&x)[S], const LaneTwiddles& tw, float* lds, int lane)
    {
        lane_twiddle_mul(x, tw);
        transpose(x, lds, lane);
        dft_dif(x);                                    // slot brev(k2) = X[l + S k2]
    }
    // slot brev(m) of lane l  ->  slot l of lane m (natural slot order on return)
    __device__ static __forceinline__ void transpose(float2 (&x)[S], float* lds, int lane)
    {
#ifdef LITHO_TRANSPOSE_OPAQUE              // build-time experiment: the 16 read bases re-derived per call instead of hoisted out of the caller's loop
        asm volatile("" : "+v"(lane));
#endif
        const int l = lane & (S - 1), line = lane >> LS;
        // S x S transpose, real parts then imaginary parts, through this line's private matrix:
        // element (row = writer lane, col = m); reader lane m takes column m.
        float* const mat = lds + line * (S * TS);
        float* const wr = mat + l * TS;
        float* const rd = mat + l;
        write_row(wr, [&](auto m_) { return x[brev(decltype(m_)::value)].x; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float re[S];
        static_for<0, S>([&](auto r_) { constexpr int r = decltype(r_)::value; re[r] = rd[r * TS]; });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        write_row(wr, [&](auto m_) { return x[brev(decltype(m_)::value)].y; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_for<0, S>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            x[r] = make_float2(re[r], rd[r * TS]);
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    // ------------------------------------------------------------------------------------------------
    // NL = 2, 4, 8 or 16 lines of N = S*S/NL points per wave, rectangular split N = S lanes x H slots (H = S / NL;
    // N = 2048, 1024, 512, 256 for S = 64):
    //   X[m + H k2] = sum_l wS^(l k2) [ w_N^(l m) sum_j x[l + S j] w_H^(j m) ],   l < S, j, m < H, k2 < S.
    // Slots [line*H, line*H + H) hold line `line` (sample l + S j in slot line*H + j).  Pass A = NL independent
    // H-point DIFs over the slots; the S x S transpose is the square kernel's: column c = line*H + m goes to lane c,
    // which then runs the full S-point pass B over l for ITS (line, m).  On return lane c = (line, m) holds
    // X_line[m + H k2] in slot brev(k2).  tw = load_lane_twiddles(table_N, lane, 1) (entries >= 8 + H/8 are unused).
    // Same register and LDS footprint as run(), and all 64 lanes stay busy on short lines.
    // ------------------------------------------------------------------------------------------------
    template <int NL>
    __device__ static __forceinline__ void run_rect(float2 (&x)[S], const LaneTwiddles& tw, float* lds, int lane)
    {
        static_assert(LS == 6 && (NL == 2 || NL == 4 || NL == 8 || NL == 16), "rectangular multi-line transform: S = 64");
        constexpr int H = S / NL, LH = LS - (NL == 2 ? 1 : NL == 4 ? 2 : NL == 8 ? 3 : 4);
        static_for<0, NL>([&](auto q_) { dif_network<LH, decltype(q_)::value * H, S>(x); });   // slot line*H + brev_LH(m) = Y_line[l, m]
        auto slot_of = [](int c) constexpr { return (c / H) * H + brev_bits(c % H, LH); };
        static_for<0, S>([&](auto c_) {
            constexpr int c = decltype(c_)::value;
            constexpr int m = c % H, a = m >> 3, b = m & 7, sl = (c / H) * H + brev_bits(m, LH);
            if constexpr (b != 0) x[sl] = cmul(x[sl], tw.row[b]);
            if constexpr (a != 0) x[sl] = cmul(x[sl], tw.row[8 + a]);
        });
#ifdef LITHO_TRANSPOSE_OPAQUE
        asm volatile("" : "+v"(lane));
#endif
        float* const wr = lds + lane * TS;
        float* const rd = lds + lane;
        write_row(wr, [&](auto c_) { return x[slot_of(decltype(c_)::value)].x; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float re[S];
        static_for<0, S>([&](auto r_) { constexpr int r = decltype(r_)::value; re[r] = rd[r * TS]; });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        write_row(wr, [&](auto c_) { return x[slot_of(decltype(c_)::value)].y; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_for<0, S>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            x[r] = make_float2(re[r], rd[r * TS]);
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        dft_dif(x);                                    // slot brev(k2) = X_line[m + H k2]
    }

    // ------------------------------------------------------------------------------------------------
    // 2*S*S-point lines (N = 8192 for S = 64) by a PAIR of waves.  With n = l + S J (J < 2S), u = M + 2S k2:
    //   X[M + 2S k2] = sum_l wS^(l k2) [ w_N^(l M) sum_J x[l + S J] w_2S^(J M) ].
    // Wave `par` (0/1) holds the samples with J = 2 j + par in slot j (n = l + S par + 2S j), runs the S-point
    // pass A on them, A_par[l,m] (m < S), and applies ITS lane twiddle w_N^((l + S par) m): B_0 = w_N^(l m) A_0,
    // B_1 = w_N^(l m) w_2S^m A_1.  Then
    //   Z[l, m]     = B_0 + B_1                    (M = m,     handled by wave 0)
    //   Z[l, m + S] = w_2S^l (B_0 - B_1)           (M = m + S, handled by wave 1)
    // so the two waves exchange B through LDS AT the transpose that pass B needs anyway: each wave writes its own
    // S x S matrix and reads column `lane` of BOTH.  The factor w_2S^l is a compile-time constant per slot after
    // the transpose.  On return slot brev(k2) = X[lane + S par + 2S k2].  Four workgroup barriers per line (the
    // partner's data must be complete before the reads, and consumed before the next overwrite); every wave of the
    // workgroup must call this the same number of times.  `tw` = load_lane_twiddles(table_N, l + S par, 1).
    // ------------------------------------------------------------------------------------------------
    __device__ static __forceinline__ void run_pair(float2 (&x)[S], const LaneTwiddles& tw, float* mat_own,
                                                    const float* mat_other, int lane, int par)
    {
        static_assert(LS == 6, "pair transform: S = 64");
        dft_dif(x);                                    // slot brev(m) = A_par[l, m]
        static_for<0, S>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            constexpr int a = m >> 3, b = m & 7, sl = brev(m);
            if constexpr (b != 0) x[sl] = cmul(x[sl], tw.row[b]);
            if constexpr (a != 0) x[sl] = cmul(x[sl], tw.row[8 + a]);
        });
        float* const wr = mat_own + lane * TS;
        const float* const rd0 = (par ? mat_other : mat_own) + lane;      // B_0: wave 0's matrix
        const float* const rd1 = (par ? mat_own : mat_other) + lane;      // B_1: wave 1's matrix
        const float sgn = par ? -1.f : 1.f;
        write_row(wr, [&](auto m_) { return x[brev(decltype(m_)::value)].x; });
        lds_barrier();
        float re[S];
        static_for<0, S>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            re[r] = fmaf(sgn, rd1[r * TS], rd0[r * TS]);
        });
        lds_barrier();
        write_row(wr, [&](auto m_) { return x[brev(decltype(m_)::value)].y; });
        lds_barrier();
        static_for<0, S>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            x[r] = make_float2(re[r], fmaf(sgn, rd1[r * TS], rd0[r * TS]));
        });
        lds_barrier();
        if (par) static_for<1, S>([&](auto r_) { constexpr int r = decltype(r_)::value; x[r] = mul_root128<r>(x[r]); });
        dft_dif(x);                                    // slot brev(k2) = X[lane + S par + 2S k2]
    }
};

using Wave4096 = WaveSq<6>;

}  // namespace litho
