// lab_kernels.hpp -- EXPERIMENTAL y-pass kernels, measured against the product kernel in scripts/ubench/ypass_lab.hip and
// REJECTED (results: profiles/r03_ypass_lab.txt).  Not part of liblitho_abbe.so.  Both are bit-/fp32-exact against
// k_ypass_rect on the same T; both are slower:
//   * k_ypass_line: one 2048-point column per wave (32 slots, <= 168 / 128 VGPRs, 3 / 4 waves per SIMD, last radix-2 stage
//     across the half-waves with v_permlane32_swap_b32).  More resident waves do NOT buy VALU throughput on gfx950: the
//     same instruction count issues at 1.58 ns per instruction per SIMD at 3-4 waves against 1.26 ns at 2.
//   * k_ypass_rect_pf: k_ypass_rect with the next item's samples prefetched by LDS-DMA into the idle transpose matrix
//     (no register cost).  The DMA issue cost and its LDS traffic outweigh the hidden latency (64 vs 57 us per launch).
#pragma once
#include "../../lithographysimulator_amd/csrc/wave_kernels.hpp"

namespace litho {

// WaveSq<6>::run_rect with a hook that runs once the wave's LDS matrix is free again (before pass B)
template <int NL, typename Hook>
__device__ __forceinline__ void run_rect_hook(float2 (&x)[64], const WaveSq<6>::LaneTwiddles& tw, float* lds, int lane, Hook&& after_transpose)
{
    using W = WaveSq<6>;
    constexpr int S = 64, LS = 6;
    constexpr int H = S / NL, LH = LS - (NL == 2 ? 1 : NL == 4 ? 2 : NL == 8 ? 3 : 4);
    static_for<0, NL>([&](auto q_) { dif_network<LH, decltype(q_)::value * H, S>(x); });
    auto slot_of = [](int c) constexpr { return (c / H) * H + brev_bits(c % H, LH); };
    static_for<0, S>([&](auto c_) {
        constexpr int c = decltype(c_)::value;
        constexpr int m = c % H, a = m >> 3, b = m & 7, sl = (c / H) * H + brev_bits(m, LH);
        if constexpr (b != 0) x[sl] = cmul(x[sl], tw.row[b]);
        if constexpr (a != 0) x[sl] = cmul(x[sl], tw.row[8 + a]);
    });
    float* const wr = lds + lane * (S + 1);
    float* const rd = lds + lane;
    static_for<0, S>([&](auto c_) { constexpr int c = decltype(c_)::value; wr[c] = x[slot_of(c)].x; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float re[S];
    static_for<0, S>([&](auto r_) { constexpr int r = decltype(r_)::value; re[r] = rd[r * (S + 1)]; });
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    static_for<0, S>([&](auto c_) { constexpr int c = decltype(c_)::value; wr[c] = x[slot_of(c)].y; });
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    static_for<0, S>([&](auto r_) {
        constexpr int r = decltype(r_)::value;
        x[r] = make_float2(re[r], rd[r * (S + 1)]);
    });
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    after_transpose();
    W::dft_dif(x);
}

}  // namespace litho

namespace litho {

// ----------------------------------------------------------------------------------
// k_ypass_rect with the NEXT line group's samples prefetched by LDS-DMA (buffer_load_dwordx4 ... lds) into the wave's own
// transpose matrix, which is idle from the end of the transposes until the next line group's.  The multi-column
// kernel has no registers left for a prefetch (256 VGPRs at two waves per SIMD) and was measured to spend a fifth of
// its time waiting for its 17 loads (scripts/ubench/ypass_lab.hip: 57 us per 12-item launch, 45 us without the loads);
// staged through LDS the loads of item s + G are in flight during pass B and the accumulation of item s and cost no
// register.  One DMA instruction moves the 64 lanes' 16-byte pieces (one live slot of the column pair) to 1 KB of LDS:
// (2 JL + 1) NL/2 KB per line group, which must fit the wave's LDS region (PF_BYTES).
// ----------------------------------------------------------------------------------
// (dma_b128_to_lds: since round 5 in the product header, csrc/wave_kernels.hpp -- k_ypass_coop_dma uses it)

template <int LOG2N, bool FULL>
struct RectPrefetch {
    static constexpr int S = 64, N = 1 << LOG2N, NL = (S * S) / N, H = S / NL;
    static constexpr int JL = FULL ? H / 4 : H / 8;
    static constexpr int NLIVE = 2 * JL + 1;
    static constexpr int STAGE_BYTES = NLIVE * (NL / 2) * 1024;
    static constexpr int MATRIX_BYTES = WaveSq<6>::LDS_FLOATS * 4;
    static constexpr int WAVE_BYTES = STAGE_BYTES > MATRIX_BYTES ? STAGE_BYTES : MATRIX_BYTES;
    static constexpr bool OK = WAVE_BYTES <= 20 * 1024;       // two workgroups of four waves per CU: 160 KB / 8
};

template <int LOG2N, int TC, bool FULL = false>
__global__ __launch_bounds__(256, 2) void k_ypass_rect_pf(
    const float2* __restrict__ Tbuf, float* __restrict__ slab, const float2* __restrict__ twtab,
    PassGeom g, int nb, int G, int gstride)
{
    using W = WaveSq<6>;
    using PF = RectPrefetch<LOG2N, FULL>;
    static_assert(PF::OK && (TC == 2 || TC == 4 || TC == 8), "prefetch stage must fit the wave's LDS region");
    constexpr int S = 64, N = 1 << LOG2N, NL = PF::NL, H = PF::H, JL = PF::JL;
    constexpr int NACC = FULL ? S : S / 2;
    auto kept_k2 = [](int i) constexpr { return FULL ? i : (i < S / 4 ? i : S / 2 + i); };
    static_assert(NL <= TC, "the wave's columns sit in one T tile");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    unsigned char* const wbase = smem_raw + (size_t)wv * PF::WAVE_BYTES;
    float* lds = reinterpret_cast<float*>(wbase);
    typename W::LaneTwiddles tw;
    W::load_lane_twiddles(tw, twtab, lane, 1);

    const int qx0 = wave_first_column<TC, 4 * NL>(blockIdx.x) + NL * wv;
    const int tile = qx0 / TC, col = qx0 & (TC - 1);
    const int plane = blockIdx.y / G, grp = blockIdx.y - plane * G;
    Tbuf += (size_t)plane * nb * g.t_point;
    slab += (size_t)plane * gstride * g.nt * 4 * g.pn;
    const bool active = tile * TC < g.pn;
    float acc[NACC];
    static_for<0, NACC>([&](auto i) { acc[i] = 0.f; });

    constexpr int RB = 8 * TC;
    const unsigned tile_bytes = active ? (unsigned)g.rows * RB : 0u;
    const unsigned vb = (unsigned)(lane - g.ky0) * RB + (unsigned)col * 8u;
    auto slot_off = [&](int j) { return vb + (unsigned)(j <= JL ? RB * S * j : RB * S * j - RB * N); };
    auto stage_index = [](int j) constexpr { return j <= JL ? j : j - (H - JL) + JL + 1; };    // live slot -> 0 .. NLIVE-1

    auto prefetch = [&](int s) {
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point + (size_t)(active ? tile : 0) * g.rows * TC, tile_bytes);
        // The slot offsets (4096 j bytes apart: one too many for the 12-bit immediate) are formed right here from ONE
        // register: hoisted out of the loop they would be 17 live registers in a kernel that has none to spare.
        unsigned vbx = vb;
        asm volatile("" : "+v"(vbx));
        static_for<0, H>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j <= JL || j >= H - JL) {
                static_for<0, NL / 2>([&](auto q_) {
                    constexpr int q = decltype(q_)::value;
                    constexpr int slot = stage_index(j) * (NL / 2) + q;
                    constexpr unsigned rel = (unsigned)(j <= JL ? RB * S * j : RB * S * j - RB * N) + 16u * q;
                    dma_b128_to_lds(rT, wbase + slot * 1024, vbx + rel);
                });
            }
        });
    };

    if (grp < nb) prefetch(grp);
    for (int s = grp; s < nb; s += G) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the staged samples of item s have landed in LDS
        float2 x[S];
        static_for<0, H>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            static_for<0, NL / 2>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                if constexpr (j <= JL || j >= H - JL) {
                    constexpr int slot = stage_index(j) * (NL / 2) + q;
                    const u32x4v v = *reinterpret_cast<const u32x4v*>(wbase + slot * 1024 + lane * 16);
                    x[(2 * q) * H + j] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
                    x[(2 * q + 1) * H + j] = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
                } else {
                    x[(2 * q) * H + j] = make_float2(0.f, 0.f);
                    x[(2 * q + 1) * H + j] = make_float2(0.f, 0.f);
                }
            });
        });
        const int snext = s + G;
        run_rect_hook<NL>(x, tw, lds, lane, [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the transposes have read the matrix: it may be overwritten
            if (snext < nb) prefetch(snext);
        });
        static_for<0, NACC>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            constexpr int k2 = kept_k2(i);
            const float2 v = x[W::brev(k2)];
            acc[i] = fmaf(v.x, v.x, fmaf(v.y, v.y, acc[i]));
        });
    }

    const int qx = qx0 + lane / H, m = lane & (H - 1);
    if (!active || qx >= g.pn) return;
    float* srow = slab + ((size_t)grp * g.nt * 4 + qx) * g.pn;
    static_for<0, NACC>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int k2 = kept_k2(i);
        constexpr int ubase = k2 < S / 2 ? H * k2 : H * k2 - N;
        srow[ubase + m + g.c] += acc[i];
    });
}


}  // namespace litho

// line_kernels.hpp -- "one line per wavefront" transforms for N = 64 * H with H = 32 slots per lane (N = 2048), and the
// y-pass built on them.  Compared with the multi-column kernel k_ypass_rect (two 2048-point columns per wave, 64 slots
// per lane, 256 VGPRs, two waves per SIMD) a wave here owns ONE column: 32 slots, 32 accumulators, <= 128 VGPRs, so
// four waves per SIMD hide each other's global-load latency and LDS transposes -- the multi-column kernel was shown to
// be bound by exactly those waits, not by VALU issue (scripts/ubench/ypass_lab.hip: 57 us per 12-item launch, 45 us
// without its global loads, 38 us without loads and LDS traffic).
//
//   X[m + 32 k2] = sum_l w64^(l k2) [ w_N^(l m) sum_j x[l + 64 j] w32^(j m) ],      l < 64, j, m < 32, k2 < 64.
//
// Pass A: 32-point DIF over the slots.  Lane twiddle.  64 x 32 transpose through wave-private LDS so that lane
// c = 32 p + m holds Z[2 l' + p, m], l' < 32 (decimation in time over l).  Pass B: 32-point DIF over l',
// U_p[k'] = sum_l' w32^(l' k') Z[2 l' + p, m].  Last radix-2 stage ACROSS the two half-waves,
//   X[m + 32 k' + 1024 q] = U_0[k'] + (-1)^q w64^k' U_1[k'],
// with v_permlane32_swap_b32 (gfx950): swapping slot pair (i, i + 16) between the halves leaves lanes 0..31 with
// U_0 and w64^k' U_1 of k' = brev5(i), lanes 32..63 with those of k' = brev5(i) + 1, and the butterfly is in-register.

namespace litho {

typedef unsigned int u32x2s __attribute__((ext_vector_type(2)));

// swap the upper half-wave of `a` with the lower half-wave of `b`
__device__ __forceinline__ void permlane32_swap(float& a, float& b)
{
    const u32x2s r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r.x);
    b = __uint_as_float(r.y);
}

struct WaveLine2048 {
    static constexpr int H = 32, LH = 5, N = 2048;
    static constexpr int LDS_FLOATS = 64 * (H + 1);           // one padded 64 x 32 fp32 matrix per wave
    static constexpr int NTW = 8 + H / 8;                      // lane twiddle factors: w^(l b), b < 8; w^(8 l a), a < 4

    struct LaneTwiddles {
        float2 row[NTW];
    };
    __device__ static __forceinline__ void load_lane_twiddles(LaneTwiddles& t, const float2* __restrict__ table, int l)
    {
        static_for<0, 8>([&](auto i_) { constexpr int i = decltype(i_)::value; t.row[i] = table[l * i]; });
        static_for<0, H / 8>([&](auto i_) { constexpr int i = decltype(i_)::value; t.row[8 + i] = table[l * 8 * i]; });
    }
    // the same factors in a workgroup-shared LDS table: entry (i, lane) at tab[i * 64 + lane]
    static constexpr int TW_LDS_FLOAT2 = NTW * 64;
    __device__ static __forceinline__ void fill_lane_twiddle_table(float2* tab, const float2* __restrict__ table, int tid,
                                                                   int nthreads)
    {
        for (int i = tid; i < TW_LDS_FLOAT2; i += nthreads) {
            const int e = i >> 6, l = i & 63;
            tab[i] = table[e < 8 ? l * e : l * 8 * (e - 8)];
        }
    }

    // bin held by slot s of lane (p = lane >> 5, m = lane & 31) on return from run(): u = m + 32 k, natural order mod N
    __host__ __device__ static constexpr int out_k(int s, int p)
    {
        return s < 16 ? brev_bits(s, LH) + p : brev_bits(s - 16, LH) + p + 32;       // s >= 16: the bin 1024 above
    }

    // x: slot j = sample l + 64 j (natural).  On return slot s of lane (p, m) = X[m + 32 out_k(s, p)].
    template <bool TW_IN_LDS>
    __device__ static __forceinline__ void run(float2 (&x)[H], const LaneTwiddles& twr, const float2* twl, float* lds, int lane)
    {
        dif_network<LH, 0, H>(x);                              // slot brev(m) = Y[l, m]
        if constexpr (TW_IN_LDS) {
            LaneTwiddles t;
            asm volatile("" ::: "memory");                     // keep the table reads inside the caller's loop
            static_for<1, NTW>([&](auto i_) { constexpr int i = decltype(i_)::value; if constexpr (i != 8) t.row[i] = twl[i * 64 + lane]; });
            static_for<1, H>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                constexpr int a = m >> 3, b = m & 7, sl = brev_bits(m, LH);
                if constexpr (b != 0) x[sl] = cmul(x[sl], t.row[b]);
                if constexpr (a != 0) x[sl] = cmul(x[sl], t.row[8 + a]);
            });
        } else {
            static_for<1, H>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                constexpr int a = m >> 3, b = m & 7, sl = brev_bits(m, LH);
                if constexpr (b != 0) x[sl] = cmul(x[sl], twr.row[b]);
                if constexpr (a != 0) x[sl] = cmul(x[sl], twr.row[8 + a]);
            });
        }
        // 64 x 32 transpose, real parts then imaginary parts: writer lane l, column m; reader lane 32 p + m takes the
        // rows l = 2 l' + p of column m.
        const int p = lane >> 5, m = lane & 31;
#ifndef LITHO_DIAG_NOLDSWRITE
        float* const wr = lds + lane * (H + 1);
        const float* const rd = lds + p * (H + 1) + m;
        static_for<0, H>([&](auto m_) { constexpr int mm = decltype(m_)::value; wr[mm] = x[brev_bits(mm, LH)].x; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        float re[H];
        static_for<0, H>([&](auto r_) { constexpr int r = decltype(r_)::value; re[r] = rd[r * 2 * (H + 1)]; });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_for<0, H>([&](auto m_) { constexpr int mm = decltype(m_)::value; wr[mm] = x[brev_bits(mm, LH)].y; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_for<0, H>([&](auto r_) {
            constexpr int r = decltype(r_)::value;
            x[r] = make_float2(re[r], rd[r * 2 * (H + 1)]);
        });
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#endif
        dif_network<LH, 0, H>(x);                              // slot brev(k') = U_p[k']
        if (p) {                                               // upper half-wave: w64^k' U_1[k']
            static_for<1, H>([&](auto k_) {
                constexpr int k = decltype(k_)::value;
                constexpr int sl = brev_bits(k, LH);
                x[sl] = mul_root64<k>(x[sl]);
            });
        }
        static_for<0, H / 2>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            permlane32_swap(x[i].x, x[i + 16].x);
            permlane32_swap(x[i].y, x[i + 16].y);
            const float2 a = x[i], b = x[i + 16];
            x[i] = cadd(a, b);
            x[i + 16] = csub(a, b);
        });
    }
};

// ----------------------------------------------------------------------------------
// y-pass, one column per wave, N = 2048 with pn = N (the coarse-grid transform: every bin kept, rows |k| <= pn/4 live),
// T in 2-column tiles ([tile][row][2]: a row of a tile is 16 bytes, so the 64 lanes of a load instruction -- 64
// consecutive rows -- read one contiguous kilobyte and four consecutive rows form one 64-byte granule for the x-pass
// to write).  Workgroup = 4 waves = 4 adjacent columns.  WPS = waves per SIMD the kernel is built for (3 or 4).
// ----------------------------------------------------------------------------------
template <int LOG2N, int WPS>
__global__ __launch_bounds__(256, WPS) void k_ypass_line(const float2* __restrict__ Tbuf, float* __restrict__ slab,
                                                         const float2* __restrict__ twtab, PassGeom g, int nb, int G,
                                                         int gstride)
{
    static_assert(LOG2N == 11, "one-line-per-wave y-pass: N = 2048");
    using W = WaveLine2048;
    constexpr int H = W::H, N = W::N, TC = 2;
    constexpr int JL = H / 4;                                 // live slots: j <= JL (k = n) and j >= H - JL (k = n - N)
    constexpr bool TW_IN_LDS = WPS >= 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds = smem + wv * W::LDS_FLOATS;
    float2* twl = reinterpret_cast<float2*>(smem + 4 * W::LDS_FLOATS);
    typename W::LaneTwiddles tw;
    if constexpr (TW_IN_LDS) {
        W::fill_lane_twiddle_table(twl, twtab, threadIdx.x, 256);
        __syncthreads();
    } else {
        W::load_lane_twiddles(tw, twtab, lane);
    }

    const int qx = blockIdx.x * 4 + wv;
    const int tile = qx / TC, col = qx & (TC - 1);
    const int plane = blockIdx.y / G, grp = blockIdx.y - plane * G;
    Tbuf += (size_t)plane * nb * g.t_point;
    slab += (size_t)plane * gstride * g.nt * 4 * g.pn;
    const bool active = qx < g.pn;
    float acc[H];
    static_for<0, H>([&](auto i) { acc[i] = 0.f; });

    constexpr int RB = 8 * TC;                                // bytes per T row inside a tile
    const unsigned tile_bytes = active ? (unsigned)g.rows * RB : 0u;
    const unsigned vb = (unsigned)(lane - g.ky0) * RB + (unsigned)col * 8u;
    auto slot_off = [&](int j) { return vb + (unsigned)(j <= JL ? RB * 64 * j : RB * 64 * j - RB * N); };

    for (int s = grp; s < nb; s += G) {
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point + (size_t)(active ? tile : 0) * g.rows * TC, tile_bytes);
        float2 x[H];
        static_for<0, H>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
#ifdef LITHO_DIAG_YNOLOAD
            if constexpr (j <= JL || j >= H - JL) x[j] = make_float2(__uint_as_float((vb + s) | 0x3f000000u), (float)j);
#else
            if constexpr (j <= JL || j >= H - JL) x[j] = buf_load_c64(rT, slot_off(j));
#endif
            else x[j] = make_float2(0.f, 0.f);
        });
        W::template run<TW_IN_LDS>(x, tw, twl, lds, lane);
        static_for<0, H>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            acc[i] = fmaf(x[i].x, x[i].x, fmaf(x[i].y, x[i].y, acc[i]));
        });
    }

    if (!active) return;
    const int p = lane >> 5, m = lane & 31;
    float* srow = slab + ((size_t)grp * g.nt * 4 + qx) * g.pn;
    static_for<0, H>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        // bin u = m + 32 (k + p) (+ 1024): centred position = u + c for u < N/2, u - N + c above
        constexpr int k0 = W::out_k(i, 0);
        const int u = m + 32 * (k0 + p);
        srow[u < N / 2 ? u + g.c : u - N + g.c] += acc[i];
    });
}

template <int LOG2N, int WPS>
hipError_t launch_ypass_line(const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int planes, int G,
                             int gstride, hipStream_t st)
{
    if constexpr (LOG2N == 11) {
        if (g.tcl != 1 || g.N != g.pn || g.N != (1 << LOG2N)) return hipErrorNotSupported;
        using W = WaveLine2048;
        static LdsOnce once;
        constexpr size_t lds = 4 * W::LDS_FLOATS * sizeof(float) + (WPS >= 4 ? (size_t)W::TW_LDS_FLOAT2 * sizeof(float2) : 0);
        auto kern = k_ypass_line<LOG2N, WPS>;
        hipError_t e = set_lds(once, kern, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((g.pn + 3) / 4, planes * G), dim3(256), lds, st, T, slab, tw, g, nb, G, gstride);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

}  // namespace litho
