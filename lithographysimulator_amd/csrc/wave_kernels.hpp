// wave_kernels.hpp -- the wave-level y-pass kernels (one wavefront, a pair of wavefronts, or several columns per
// wavefront), the multi-row x-pass built on the same engine, and their launchers.  Included ONLY by instw_*.hip
// (N = 256 .. 8192): these translation units can take their own scheduling flags (Makefile WAVEFLAGS_<log2 N>), so the
// kernels must not be instantiated anywhere else.  Today every one of them is built with the default strategy:
// max-ilp was 2 % faster for the half-output k_ypass_wave<12> but makes its full-output variant spill 73 registers,
// and is 3-10 % slower for k_ypass_rect<11,.,true> and k_ypass_pair (A/B numbers: Makefile, profiles/r02_tuning_sweeps.txt).
// (Makefile INSTFLAGS_13 = max-ilp applies to inst_13.o, the radix-16 unit with the N = 8192 split x-pass, not to these.)
#pragma once
#include "engine_kernels.hpp"

namespace litho {

// ----------------------------------------------------------------------------------
// y-pass, wave-per-line variant (wave_fft.hpp) for N = S*S*D, pn = N/2, pupil inside the unit disk:
//   N = 1024: S = 32, D = 1     N = 2048: S = 32, D = 2     N = 4096: S = 64, D = 1     N = 8192: S = 64, D = 2
// A "unit" is one S*S-point sub-transform, computed by S lanes on their own: no workgroup barriers, half
// the LDS traffic of the radix-16 engine.  D = 1: one unit per column.  D = 2: decimation in frequency,
//   out[2v + p] = sum_n' ([x[n'] + (-1)^p x[n' + N/2]] w_N^(n' p)) w_{N/2}^(n' v),
// i.e. two independent sub-transforms (even / odd bins) per column; the pruned input makes the first
// butterfly trivial (at most one of x[n'], x[n'+N/2] is non-zero) and the two units never exchange data.
// Every consumer of a 4-column T tile sits in the same workgroup.
// ----------------------------------------------------------------------------------
template <int LOG2N>
struct WaveShape {
    static_assert(LOG2N >= 10 && LOG2N <= 12, "wave-per-line y-pass: N = 1024 .. 4096 (N = 8192: k_ypass_pair)");
    static constexpr int LS = LOG2N / 2;                     // 5, 5, 6, 6
    static constexpr int D = 1 << (LOG2N - 2 * LS);          // 1, 2, 1, 2
    using W = WaveSq<LS>;
    static constexpr int S = W::S;
    // (one 512-thread workgroup per 8-column tile, so that every consumer of a 64-byte row granule sits on one CU,
    // was measured slower at 2048^2: y-pass 10.0 vs 9.5 us/point)
    static constexpr int THREADS = 256;
    static constexpr int UNITS = THREADS / S;                // sub-transforms per workgroup
    static constexpr int COLS = UNITS / D;                   // columns per workgroup (4 or 8)
    static constexpr int TILES = COLS / 4;                   // T tiles per workgroup
    static constexpr int JLIVE = S * D / 8;                  // live slots: j in [0, JLIVE] and [S - JLIVE, S)
    static constexpr size_t LDS_BYTES = (size_t)(THREADS / 64) * W::LDS_FLOATS * sizeof(float);
#ifndef LITHO_WAVE32_MINWAVES
#define LITHO_WAVE32_MINWAVES 4
#endif
    static constexpr int MINWAVES = LS == 5 ? LITHO_WAVE32_MINWAVES : 2;   // S = 32 needs few registers: 4 workgroups per CU
};

// Block index -> first column of the workgroup.  A workgroup covers COLS columns of T tiles that are TC columns wide;
// when a tile is shared by WPT = TC / COLS workgroups these are blocks b, b + 8, ... (same XCD, back to back), so the
// tile's granules are served by one L2.  The grid must have wave_grid_x<TC, COLS>(pn) blocks.
template <int TC, int COLS>
__device__ __forceinline__ int wave_first_column(int b)
{
    constexpr int WPT = TC > COLS ? TC / COLS : 1;
    if constexpr (WPT == 1) return b * COLS;
    else return (((b / (8 * WPT)) * 8 + (b & 7)) * TC) + ((b >> 3) % WPT) * COLS;
}
template <int TC, int COLS>
static inline int wave_grid_x(int pn)
{
    constexpr int WPT = TC > COLS ? TC / COLS : 1;
    if constexpr (WPT == 1) return (pn + COLS - 1) / COLS;
    else return WPT * (((pn + TC - 1) / TC + 7) / 8 * 8);
}

typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
// cache-policy bits of k_ypass_rect's T loads (build-time experiment hook: sc0 = 1, nt = 2, sc1 = 16)
#ifndef LITHO_YLOAD_AUX
#define LITHO_YLOAD_AUX 0
#endif

// FULL = false: N = 2 pn (half of the bins are kept, a quarter of the inputs live).  FULL = true: N = pn -- the
// "coarse grid" transform (every bin kept, half of the inputs live; D = 1 only), see k_ypass_rect.
// The full-output kernel drags 33 64-byte row granules per lane and line through the L1 for 8 useful bytes each, and
// that traffic -- not the arithmetic -- is its limit: a workgroup barrier per line keeps the four waves of a workgroup
// (four adjacent columns of one tile) in step, so that the L1 serves three of their four requests for a granule
// (4096^2, y-pass us per source point: 32.5 free-running, 28.2 with the barrier; one 512-thread workgroup per tile
// with the barrier: 34.5).  The same barrier costs the half-output and the multi-column kernels 5-10 %.
template <int LOG2N, int TC, bool FULL = false>
__global__ __launch_bounds__(WaveShape<LOG2N>::THREADS, WaveShape<LOG2N>::MINWAVES) void k_ypass_wave(
    const float2* __restrict__ Tbuf, float* __restrict__ slab, const float2* __restrict__ twtab,
    PassGeom g, int nb, int G, int gstride)
{
    static_assert(TC == 4 || TC == 8, "T tiles are 4 or 8 columns wide");
    using WS = WaveShape<LOG2N>;
    using W = typename WS::W;
    static_assert(!FULL || WS::D == 1, "full-output variant: one unit per column");
    constexpr int S = WS::S, D = WS::D, JLIVE = FULL ? 2 * WS::JLIVE : WS::JLIVE, N = 1 << LOG2N;
    constexpr int NACC = FULL ? S : S / 2;                   // kept bins per lane
    auto kept_k2 = [](int i) constexpr { return FULL ? i : (i < S / 4 ? i : S / 2 + i); };
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l = lane & (S - 1);                            // lane inside its unit
    const int unit = wv * W::LINES + (lane >> WS::LS);
    const int p = unit % D;                                  // residue of the output bins this unit makes
    const int colg = unit / D;                               // column inside the workgroup's column group
    float* lds = smem + wv * W::LDS_FLOATS;
    typename W::LaneTwiddles tw;
    float2* twlds = reinterpret_cast<float2*>(smem_raw + WS::LDS_BYTES);      // FULL: lane twiddles in LDS (registers go to acc)
    if constexpr (FULL) {
        W::fill_lane_twiddle_table(twlds, twtab, D, threadIdx.x, WS::THREADS);
        __syncthreads();
    } else {
        W::load_lane_twiddles(tw, twtab, l, D);              // w_{S*S}^e is entry D*e of the w_N table
    }
    const float2 tl = twtab[l * p];                          // w_N^(l p)  (1 for p = 0)

    // The tile index is wave-uniform (a wave's units cover at most one tile); readfirstlane makes that
    // provable, otherwise every buffer access through the tile's descriptor becomes a waterfall loop.
    const int qx = wave_first_column<TC, WS::COLS>(blockIdx.x) + colg;
    const int tile = __builtin_amdgcn_readfirstlane(qx / TC);
    const int col = qx & (TC - 1);
    const int plane = blockIdx.y / G, grp = blockIdx.y - plane * G;      // see k_ypass_acc
    Tbuf += (size_t)plane * nb * g.t_point;
    slab += (size_t)plane * gstride * g.nt * 4 * g.pn;
    const bool active = tile * TC < g.pn;
    float acc[NACC];
    static_for<0, NACC>([&](auto i) { acc[i] = 0.f; });

    // Live input slots j (sub-transform sample n' = l + S j): k = n' for j <= JLIVE, k = n' - S*S for the upper
    // ones (sample n' + N - S*S of the full line); T row a = k - ky0.  The descriptor is windowed on this tile's
    // rows ([tile][row][TC] layout, RB = 8 TC bytes per row): the range check is the validity test.
    constexpr int RB = 8 * TC;
    const unsigned tile_bytes = active ? (unsigned)g.rows * RB : 0u;
    const unsigned vb = (unsigned)(l - g.ky0) * RB + (unsigned)col * 8u;
    auto slot_off = [&](int j) { return vb + (unsigned)(j <= JLIVE ? RB * S * j : RB * S * j - RB * S * S); };

    for (int s = grp; s < nb; s += G) {
#ifndef LITHO_WAVE_FULL_NOSYNC
        if constexpr (FULL) __builtin_amdgcn_s_barrier();      // the waves of a tile in step: one L1 fill per granule
#endif
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point + (size_t)(active ? tile : 0) * g.rows * TC, tile_bytes);
        float2 x[S];
#ifndef LITHO_WAVE_FULL_NOPAIR
        if constexpr (FULL && TC == 8) {
            // Pair loading: waves 2i and 2i+1 own adjacent columns (one 16-byte half of a row granule).  The even wave
            // loads BOTH columns of the lower live slots with 16-byte loads, the odd wave both columns of the upper
            // ones; each keeps its own column and hands the other to its partner through the partner's transpose
            // matrix, which is idle until pass A is done (the barrier at the top of the loop makes sure of that).
            // Half the load instructions, and every granule is requested once per pair instead of twice.
            float2* const mine = reinterpret_cast<float2*>(lds);
            float2* const theirs = reinterpret_cast<float2*>(smem + (wv ^ 1) * W::LDS_FLOATS);
            const unsigned vbp = vb - (unsigned)(col & 1) * 8u;          // the pair's even column
            auto pair_off = [&](int j) { return vbp + (unsigned)(j <= JLIVE ? RB * S * j : RB * S * j - RB * S * S); };
            static_for<JLIVE + 1, S - JLIVE>([&](auto j_) { x[decltype(j_)::value] = make_float2(0.f, 0.f); });
            const bool even_wave = (__builtin_amdgcn_readfirstlane(wv) & 1) == 0;      // wave-uniform: a scalar branch
            if (even_wave) {
                static_for<0, JLIVE + 1>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rT, pair_off(j), 0, 0);
                    x[j] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
                    theirs[j * 64 + lane] = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
                });
            } else {
                static_for<S - JLIVE, S>([&](auto j_) {
                    constexpr int j = decltype(j_)::value;
                    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rT, pair_off(j), 0, 0);
                    x[j] = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));
                    theirs[(j - (S - JLIVE)) * 64 + lane] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
                });
            }
            __syncthreads();
            if (even_wave) {
                static_for<S - JLIVE, S>([&](auto j_) { constexpr int j = decltype(j_)::value; x[j] = mine[(j - (S - JLIVE)) * 64 + lane]; });
            } else {
                static_for<0, JLIVE + 1>([&](auto j_) { constexpr int j = decltype(j_)::value; x[j] = mine[j * 64 + lane]; });
            }
            asm volatile("" ::: "memory");             // the transposes below reuse `mine`: keep these reads in front of them
        } else
#endif
        static_for<0, S>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j <= JLIVE || j >= S - JLIVE) {
                float2 v = buf_load_c64(rT, slot_off(j));
                if constexpr (D > 1) {
                    // w_N^(n' p) = w_N^(l p) * w_N^(S j p), and (-1)^p for the upper half (q = 1)
                    if (p) {
                        constexpr int e = (S * j) % N;
                        constexpr double ang = 6.283185307179586476925 * e / N;
                        constexpr float sgn = (j >= S - JLIVE) ? -1.f : 1.f;
                        const float2 cj = make_float2(sgn * (float)__builtin_cos(ang), sgn * (float)__builtin_sin(ang));
                        v = cmul(cmul(v, cj), tl);
                    }
                }
                x[j] = v;
            } else {
                x[j] = make_float2(0.f, 0.f);
            }
        });
        if constexpr (FULL) W::template run_lds_tw<true>(x, twlds, lds, lane);
        else W::run(x, tw, lds, lane);
        // kept bins of the sub-transform: v in [-S*S/4, S*S/4)  ->  k2 in [0, S/4) and [3S/4, S)
        static_for<0, NACC>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            constexpr int k2 = kept_k2(i);
            const float2 v = x[W::brev(k2)];
            acc[i] = fmaf(v.x, v.x, fmaf(v.y, v.y, acc[i]));
        });
    }

    if (!active || qx >= g.pn) return;
    float* srow = slab + ((size_t)grp * g.nt * 4 + qx) * g.pn;
    static_for<0, NACC>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int k2 = kept_k2(i);
        const int n = l + S * k2;
        const int v = n < S * S / 2 ? n : n - S * S;        // bin of the sub-transform
        srow[D * v + p + g.c] += acc[i];                    // bin u = D v + p of the full line
    });
}

// ----------------------------------------------------------------------------------
// y-pass for the 4096-point coarse-grid transform (N = pn = 4096: BASELINE config 4) over 16-COLUMN T tiles.  At 4096^2 a
// T item is 67 MB: T streams through HBM, and there the x-pass depends on how a line of T is stored -- 8-column tiles
// (two rows per 128-byte line) 21.6 us per item, 16-column tiles (a store instruction covers whole lines) 14.0 -- but a wave
// that owns one column uses 8 bytes of every 128-byte row (k_ypass_wave over 16-column tiles: 21.2 -> 28.8 us per item).
// Here the four waves of a workgroup own four adjacent columns = 32 bytes of every tile row and load them TOGETHER: the
// line's 33 live slots x 64 rows are 66 units of 32 rows; wave w takes the units w, w + 4, ... -- two lanes per row, 16 bytes
// per lane, 32 rows per load instruction, 17 loads per wave -- and every lane deals its two columns into the staging areas
// of the waves that own them (the transpose matrices, idle at that point).  After a barrier every wave reads its 33 live
// samples back and runs the transform of k_ypass_wave<12, ., true>, with the lane-twiddle table read AFTER pass A.  Two
// workgroup barriers per line; 76 KB of LDS, two workgroups per CU.  Measured steps (us per item): 64 rows x 32 bytes per
// instruction with hoisted (spilled) offsets 38.9, literal offsets 33.3, two lanes per row 29.5, no accumulator spills
// (1.3 GB of scratch writes per launch before) 24.4.  An eight-wave variant (64 bytes of the row, one 143 KB workgroup per
// CU) left the load latency exposed: 44.  profiles/r03_slab4096_probe.txt.
// ----------------------------------------------------------------------------------
template <int NW>
struct CoopShape {
    static_assert(NW == 4, "four waves share four columns (32 bytes) of a tile row");
    static constexpr int S = 64, N = 4096, JLIVE = 16, NLIVE = 2 * JLIVE + 1;       // live slots: j <= 16 and j >= 48
    static constexpr int STAGE_FLOATS = NLIVE * 64 * 2 + 16;                        // staging (33 x 64 float2); 64-byte skew between areas
    static constexpr int WAVE_FLOATS = STAGE_FLOATS >= WaveSq<6>::LDS_FLOATS ? STAGE_FLOATS : WaveSq<6>::LDS_FLOATS + 16;   // ... or the transpose matrix, if larger
    static_assert(WAVE_FLOATS >= WaveSq<6>::LDS_FLOATS, "the staging area holds the transpose matrix");
    static constexpr size_t LDS_BYTES = NW * (size_t)WAVE_FLOATS * sizeof(float) + (size_t)WaveSq<6>::TW_LDS_FLOAT2 * sizeof(float2);
    static constexpr int PARTS = 16 / NW;                                           // workgroups per 16-column tile
    static inline int grid_x(int pn) { return PARTS * (((pn + 15) / 16 + 7) / 8 * 8); }
};

#ifdef LITHO_DIAG_COOP_NOLOAD
#define COOP_LOAD(r, off) u32x4v{(off), 0x3f800000u, (off) >> 3, 0x3f000000u}
#else
#define COOP_LOAD(r, off) __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0)
#endif
template <int LOG2N, int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_ypass_coop(
    const float2* __restrict__ Tbuf, float* __restrict__ slab, const float2* __restrict__ twtab,
    PassGeom g, int nb, int G, int gstride)
{
    static_assert(LOG2N == 12, "cooperative-loading y-pass over 16-column tiles: N = 4096");
    using W = WaveSq<6>;
    using CS = CoopShape<NW>;
    constexpr int S = CS::S, N = CS::N, JLIVE = CS::JLIVE, NLIVE = CS::NLIVE, TC = 16, RB = 8 * TC;
    constexpr int PART_BYTES = 8 * NW;                         // bytes of a tile row this workgroup owns (NW columns)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & (NW - 1);      // column inside the workgroup's part
    float* lds = smem + wv * CS::WAVE_FLOATS;
    float2* twlds = reinterpret_cast<float2*>(smem + NW * CS::WAVE_FLOATS);
    W::fill_lane_twiddle_table(twlds, twtab, 1, threadIdx.x, 64 * NW);
    __syncthreads();

    // blocks b, b + 8, ... (same XCD, back to back) take the PARTS parts of one 16-column tile
    const int b = blockIdx.x;
    const int tile = (b / (8 * CS::PARTS)) * 8 + (b & 7), part = (b >> 3) % CS::PARTS;
    const int qx = tile * TC + part * NW + wv;
    const int plane = blockIdx.y / G, grp = blockIdx.y - plane * G;
    Tbuf += (size_t)plane * nb * g.t_point;
    slab += (size_t)plane * gstride * g.nt * 4 * g.pn;
    const bool active = tile * TC < g.pn;
    float acc[S];
    static_for<0, S>([&](auto i) { acc[i] = 0.f; });

    const unsigned tile_bytes = active ? (unsigned)g.rows * RB : 0u;
    // two lanes per row: lane 2r + e takes 16 bytes = columns 2e, 2e + 1 of the workgroup's 32 bytes of row r (32 rows per load)
    // (the lane's offsets are re-derived inside the loop, from one opaque register: hoisted, they are spilled and every
    // line then opens with scratch reloads in front of its loads)
    const int i0 = wv >> 1, h = wv & 1;                        // wave-uniform: this wave's slots are i0, i0 + 2, ..., rows 32 h ...
    const unsigned sb = (unsigned)(-g.ky0 * RB + part * PART_BYTES + i0 * RB * S + h * 32 * RB);
    const unsigned last_oob = i0 ? 0x80000000u : 0u;           // slot 33 does not exist: out of range, the load returns zeros
    auto live_index = [](int j) constexpr { return j <= JLIVE ? j : j - (S - NLIVE); };

    for (int s = grp; s < nb; s += G) {
        __syncthreads();                                       // every wave is done with the previous line's matrices
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point + (size_t)(active ? tile : 0) * g.rows * TC, tile_bytes);
        // Units (slot, half of its 64 rows) wv, wv + 4, ...: the half is h = wv & 1, the slots i0 = wv >> 1, i0 + 2, ...
        // The row offset is re-derived from one register per line: hoisted out of the loop, the 17 load offsets of a
        // wave spill, and every load then waits for a scratch reload (vmcnt(0)) before it is issued.
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int lrow = ln >> 1, le = ln & 1;
        const unsigned vbx = (unsigned)lrow * RB + (unsigned)le * 16u + sb;
        constexpr int NM = (NLIVE + 1) / 2;                    // 17 slots for i0 = 0, 16 for i0 = 1
        u32x4v v[NM];
        static_for<0, NM>([&](auto m_) {
            constexpr int m = decltype(m_)::value;             // slot i = i0 + 2 m: live index, j = i (i <= 16) or i + 31
            constexpr unsigned lo = (unsigned)(RB * S * 2 * m), hi = (unsigned)(RB * S * (2 * m + S - NLIVE) - RB * N);
            if constexpr (2 * m + 1 <= JLIVE) v[m] = COOP_LOAD(rT, vbx + lo);
            else if constexpr (2 * m + 1 >= NLIVE) v[m] = COOP_LOAD(rT, vbx + (hi + last_oob));
            else if constexpr (2 * m > JLIVE) v[m] = COOP_LOAD(rT, vbx + hi);
            else v[m] = COOP_LOAD(rT, vbx + (i0 ? hi : lo));
        });
        // column 2e's area, this lane's row of this wave's first slot
        float2* const dealw = reinterpret_cast<float2*>(smem + 2 * le * CS::WAVE_FLOATS) + (lrow + i0 * 64 + 32 * h);
        static_for<0, NM>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            if (2 * m + 1 < NLIVE || i0 == 0) {
                const u32x4v t = v[m];
                dealw[m * 128] = make_float2(__uint_as_float(t.x), __uint_as_float(t.y));
                dealw[CS::WAVE_FLOATS / 2 + m * 128] = make_float2(__uint_as_float(t.z), __uint_as_float(t.w));
            }
        });
        __syncthreads();
        float2 x[S];
        const float2* mine = reinterpret_cast<const float2*>(lds);
        static_for<0, S>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j <= JLIVE || j >= S - JLIVE) x[j] = mine[live_index(j) * 64 + lane];
            else x[j] = make_float2(0.f, 0.f);
        });
#ifndef LITHO_COOP_NOCLOBBER
        asm volatile("" ::: "memory");                         // the transposes below reuse `mine`: keep these reads in front of them
#endif
        W::run_lds_tw(x, twlds, lds, lane);
        static_for<0, S>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            const float2 v = x[W::brev(i)];
            acc[i] = fmaf(v.x, v.x, fmaf(v.y, v.y, acc[i]));
        });
    }
    if (!active || qx >= g.pn) return;
    float* srow = slab + ((size_t)grp * g.nt * 4 + qx) * g.pn;
    static_for<0, S>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        const int n = lane + S * i;
        const int v = n < S * S / 2 ? n : n - S * S;
        srow[v + g.c] += acc[i];
    });
}

// ----------------------------------------------------------------------------------
// k_ypass_coop with the NEXT line's samples in flight while the current line is transformed (round 5).  k_ypass_coop is
// latency-bound: per line a wave issues its 17 loads, waits, deals, waits at a barrier, and only then has arithmetic to do, and
// at two waves per SIMD (64 samples + 64 accumulators per lane) the partner of a waiting wave issues at the lone-wave rate
// (profiles/r04_pmc_cfg4_summary.txt: VALU busy 33 %, 0.35 of the HBM peak).  There are no registers for a prefetch (68 more)
// and no LDS for a second staging area -- but the staging area IS the transpose matrix, and that is idle from the end of a
// line's transposes to the start of the next line's: 60 % of the line (pass B + accumulation).  So the loads of line s + G are
// issued right after the transposes of line s as LDS-DMA (buffer_load_dwordx4 ... lds: no register, no deal), each wave INTO
// ITS OWN region (its matrix: free as soon as its own transposes are done -- no barrier needed before the issue), and every
// wave reads its column out of all four regions after the barrier at the top of the next line.  One DMA instruction = 32 rows
// x 32 bytes = 1 KB of LDS in lane order; the lane -> (row, 16-byte half) assignment is swizzled (half ^= bit 3 of the row) so
// that the 64 8-byte reads of a column spread over all banks.  Two workgroup barriers per line, as before:
//   B1 (top): every wave's DMA has landed (each waits vmcnt(0) first)            -> read x from the regions
//   B2 (after pass A + lane twiddles): every wave has read its x                 -> the matrices may overwrite the regions
// ----------------------------------------------------------------------------------
template <int NW>
struct CoopDmaShape {
    static_assert(NW == 4 || NW == 8, "four waves share four columns (32 bytes) of a tile row -- or, as an experiment, eight share 64 bytes");
    static constexpr int S = 64, N = 4096, JLIVE = 16, NLIVE = 2 * JLIVE + 1;
    static constexpr int QB = NW / 2;                                               // 16-byte pieces of a tile row the workgroup owns
    static constexpr int RPU = 64 / QB;                                             // rows per 1 KB unit (32 x 32 bytes, or 16 x 64 bytes)
    static constexpr int UNITS = (NLIVE + 1) / 2;                                   // 1 KB units per wave and line
    static constexpr int WAVE_BYTES = UNITS * 1024;                                 // 17 KB >= the 64 x 65 fp32 matrix
    static_assert(WAVE_BYTES >= WaveSq<6>::LDS_FLOATS * 4 && WAVE_BYTES % 256 == 0, "a region holds the transpose matrix; regions bank-aligned");
    static constexpr size_t LDS_BYTES = NW * (size_t)WAVE_BYTES + (size_t)WaveSq<6>::TW_LDS_FLOAT2 * sizeof(float2);
    static constexpr int PARTS = 16 / NW;
    static inline int grid_x(int pn) { return PARTS * (((pn + 15) / 16 + 7) / 8 * 8); }
};

// one LDS-DMA instruction: lane i's 16 bytes at byte offset `voff` of the buffer -> LDS bytes [16 i, 16 i + 16) of `dst`
// (dst is wave-uniform: it travels in M0); out-of-range lanes deliver zeros
#ifndef LITHO_COOP_DMA_AUX                // cache-policy bits (build-time experiment hook: sc0 = 1, nt = 2, sc1 = 16)
#define LITHO_COOP_DMA_AUX 0
#endif
__device__ __forceinline__ void dma_b128_to_lds(__amdgpu_buffer_rsrc_t r, unsigned char* dst, unsigned voff)
{
#if defined(__HIP_DEVICE_COMPILE__)       // (the host pass has no LDS address space to cast to)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)dst, 16, voff, 0, 0, LITHO_COOP_DMA_AUX);
#endif
}

template <int LOG2N, int NW>
__global__ __launch_bounds__(64 * NW, 2) void k_ypass_coop_dma(
    const float2* __restrict__ Tbuf, float* __restrict__ slab, const float2* __restrict__ twtab,
    PassGeom g, int nb, int G, int gstride)
{
    static_assert(LOG2N == 12, "cooperative-loading y-pass over 16-column tiles: N = 4096");
    using W = WaveSq<6>;
    using CS = CoopDmaShape<NW>;
    constexpr int S = CS::S, N = CS::N, JLIVE = CS::JLIVE, NLIVE = CS::NLIVE, TC = 16, RB = 8 * TC;
    constexpr int PART_BYTES = 8 * NW;                         // bytes of a tile row this workgroup owns (NW columns)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) & (NW - 1);      // column inside the workgroup's part
    unsigned char* const region = smem_raw + wv * CS::WAVE_BYTES;                    // this wave's DMA target = its transpose matrix
    float* lds = reinterpret_cast<float*>(region);
    float2* twlds = reinterpret_cast<float2*>(smem_raw + NW * CS::WAVE_BYTES);
    W::fill_lane_twiddle_table(twlds, twtab, 1, threadIdx.x, 64 * NW);
    __syncthreads();

    // blocks b, b + 8, ... (same XCD, back to back) take the PARTS parts of one 16-column tile
    const int b = blockIdx.x;
#if !defined(LITHO_COOP_MAP) || LITHO_COOP_MAP == 0
    const int tile = (b / (8 * CS::PARTS)) * 8 + (b & 7), part = (b >> 3) % CS::PARTS;
#elif LITHO_COOP_MAP == 1
    // experiment: the two workgroups that SHARE A CU (XCD-local workgroups j and j + 32, if the dispatcher fills the 32 CUs of an
    // XCD round-robin, one slot at a time) take adjacent parts of one tile, so that their L1 serves each 128-byte line twice
    const int xcd = b & 7, j = b >> 3, rnd = j >> 6, jj = j & 63, q = (jj & 31) + 32 * rnd;
    const int tile = (q >> 1) * 8 + xcd, part = 2 * (q & 1) + (jj >> 5);
#else
    // experiment: the same for consecutive XCD-local workgroups (j, j + 1)
    const int xcd = b & 7, j = b >> 3, k = j >> 1;
    const int tile = (k >> 1) * 8 + xcd, part = 2 * (k & 1) + (j & 1);
#endif
    const int qx = tile * TC + part * NW + wv;
    const int plane = blockIdx.y / G, grp = blockIdx.y - plane * G;
    Tbuf += (size_t)plane * nb * g.t_point;
    slab += (size_t)plane * gstride * g.nt * 4 * g.pn;
    const bool active = tile * TC < g.pn;
    float acc[S];
    static_for<0, S>([&](auto i) { acc[i] = 0.f; });

    const unsigned tile_bytes = active ? (unsigned)g.rows * RB : 0u;
    // Units (live slot, half -- NW = 8: quarter -- of its 64 rows): wave wv takes the rows RPU h .. of the slots i0, i0 + 2, ...
    constexpr int QB = CS::QB, RPU = CS::RPU;
    const int i0 = wv / QB, h = wv % QB;                       // wave-uniform
    const unsigned sb = (unsigned)(-g.ky0 * RB + part * PART_BYTES + i0 * RB * S + h * RPU * RB);
    constexpr int NM = CS::UNITS;                              // 17 units for i0 = 0, 16 for i0 = 1

    // units [M0, M1) of line s
    auto prefetch_part = [&](int s, auto m0_, auto m1_) {
#ifdef LITHO_DIAG_COOP_NOLOAD                                    // timing diagnostic (wrong results): no global traffic at all
        return;
#endif
        constexpr int M0 = decltype(m0_)::value, M1 = decltype(m1_)::value;
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point + (size_t)(active ? tile : 0) * g.rows * TC, tile_bytes);
        // DMA lane p -> row p >> 1 of the unit, 16-byte half (p & 1) ^ (bit 3 of the row): LDS granule p of the unit.
        // Re-derived per line from one opaque register (hoisted, the offsets would be live across the transform).
        int ln = lane;
        asm volatile("" : "+v"(ln));
#ifdef LITHO_COOP_DMA_NOSWZ
        const int lrow = ln / QB, le = ln % QB;
#else
        // (QB = 2: piece ^= bit 3 of the row; QB = 4: piece ^= bits 2-3 of the row -- granule QB row + piece then covers every
        // 16-byte position of a 256-byte bank row equally often for a fixed column)
        const int lrow = ln / QB, le = (ln ^ (lrow / (16 / QB))) % QB;
#endif
#if defined(LITHO_DIAG_COOP_DMA_MODE) && LITHO_DIAG_COOP_DMA_MODE == 1     // timing diagnostics (wrong results): one row per instruction,
        const unsigned vbx = (unsigned)le * 16u + sb + 0 * lrow;
#elif defined(LITHO_DIAG_COOP_DMA_MODE) && LITHO_DIAG_COOP_DMA_MODE == 2   // every lane out of range (no memory traffic, zeros to LDS)
        const unsigned vbx = 0xF0000000u + (unsigned)lrow * RB + (unsigned)le * 16u;
#else
        const unsigned vbx = (unsigned)lrow * RB + (unsigned)le * 16u + sb;
#endif
        static_for<M0, (M1 < NM ? M1 : NM)>([&](auto m_) {
            constexpr int m = decltype(m_)::value;             // slot i = i0 + 2 m: live index, j = i (i <= 16) or i + 31
            constexpr unsigned lo = (unsigned)(RB * S * 2 * m), hi = (unsigned)(RB * S * (2 * m + S - NLIVE) - RB * N);
            unsigned char* const dst = region + m * 1024;
            if constexpr (2 * m + 1 <= JLIVE) dma_b128_to_lds(rT, dst, vbx + lo);
            else if constexpr (2 * m + 1 >= NLIVE) { if (i0 == 0) dma_b128_to_lds(rT, dst, vbx + hi); }      // slot 33 does not exist
            else if constexpr (2 * m > JLIVE) dma_b128_to_lds(rT, dst, vbx + hi);
            else dma_b128_to_lds(rT, dst, vbx + (i0 ? hi : lo));
        });
    };
    auto prefetch = [&](int s) { prefetch_part(s, std::integral_constant<int, 0>{}, std::integral_constant<int, NM>{}); };

    // reader: lane = row of the slot (half hh = lane >> 5, row r = lane & 31 of its unit); column wv = 16-byte half e, float2 c
    // of the half.  Live slot li sits in the region of wave 2 (li & 1) + hh, unit li >> 1, granule 2 r + (e ^ bit 3 of r).
    const int rr = lane % RPU, hh = lane / RPU;
#ifdef LITHO_COOP_DMA_NOSWZ
    const unsigned rd_lane = (unsigned)((QB * rr + (wv >> 1)) * 16 + (wv & 1) * 8);
#else
    const unsigned rd_lane = (unsigned)((QB * rr + (((wv >> 1) ^ (rr / (16 / QB))) % QB)) * 16 + (wv & 1) * 8);
#endif
    const unsigned char* const rd_even = smem_raw + hh * CS::WAVE_BYTES + rd_lane;
    const unsigned char* const rd_odd = rd_even + QB * CS::WAVE_BYTES;
    auto live_index = [](int j) constexpr { return j <= JLIVE ? j : j - (S - NLIVE); };

    if (grp < nb) prefetch(grp);
    for (int s = grp; s < nb; s += G) {
#ifndef LITHO_DIAG_COOP_NOWAIT                                   // (timing diagnostic, wrong results: the loads are issued, nobody waits)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's units of line s have landed in its region
#endif
        lds_barrier();                                         // B1: and everybody else's in theirs
        float2 x[S];
        static_for<0, S>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j <= JLIVE || j >= S - JLIVE) {
                constexpr int li = live_index(j);
                x[j] = *reinterpret_cast<const float2*>((li & 1 ? rd_odd : rd_even) + (li >> 1) * 1024);
            } else {
                x[j] = make_float2(0.f, 0.f);
            }
        });
        W::dft_dif(x);                                         // pass A: slot brev(m) = y[m]
        typename W::LaneTwiddles tw;
        asm volatile("" ::: "memory");                         // the table reads stay behind pass A (30 registers)
        static_for<1, W::NTW>([&](auto i_) { constexpr int i = decltype(i_)::value; if constexpr (i != 8) tw.row[i] = twlds[i * 64 + lane]; });
        W::lane_twiddle_mul(x, tw);
        lds_barrier();                                         // B2: every wave has taken its samples out of the regions
        int lt = lane;                                         // opaque: the transposes' 16 read bases (ds_read2 reaches 1 KB) are then
        asm volatile("" : "+v"(lt));                           // re-derived per line instead of living -- spilled -- across the loop
        W::transpose(x, lds, lt);
        const int snext = s + G;
#ifdef LITHO_COOP_DMA_SPREAD                                     // the 17 DMA instructions spread over the six stages of pass B
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        W::dft_dif_hook(x, [&](auto st_) {
            constexpr int st = decltype(st_)::value;
            __builtin_amdgcn_sched_barrier(0);
            if (snext < nb) prefetch_part(snext, std::integral_constant<int, 3 * st>{}, std::integral_constant<int, 3 * st + 3>{});
            __builtin_amdgcn_sched_barrier(0);
        });
#else
        if (snext < nb) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the transposes have read the matrix: the DMA may overwrite it
            prefetch(snext);
        }
        W::dft_dif(x);                                         // pass B
#endif
        static_for<0, S>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            const float2 v = x[W::brev(i)];
            acc[i] = fmaf(v.x, v.x, fmaf(v.y, v.y, acc[i]));
        });
    }
    if (!active || qx >= g.pn) return;
    float* srow = slab + ((size_t)grp * g.nt * 4 + qx) * g.pn;
    static_for<0, S>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        const int n = lane + S * i;
        const int v = n < S * S / 2 ? n : n - S * S;
        srow[v + g.c] += acc[i];
    });
}

// ----------------------------------------------------------------------------------
// y-pass for N = 8192, pn = 4096 (BASELINE config 4), pupil inside the unit disk: a PAIR of waves per column
// (WaveSq<6>::run_pair).  Each wave loads 17 live slots (like the 4096-point kernel), runs a pruned-input pass A,
// exchanges with its partner at the transpose, and accumulates the 32 kept bins of its half of the line.
// Workgroup = 4 waves = 2 columns; the two workgroups that share a 4-column T tile are blocks b and b + 8 (same
// XCD, back to back) so the tile's 32-byte granules are served by one L2.
// ----------------------------------------------------------------------------------
template <int LOG2N, int TC>
__global__ __launch_bounds__(256, 2) void k_ypass_pair(
    const float2* __restrict__ Tbuf, float* __restrict__ slab, const float2* __restrict__ twtab,
    PassGeom g, int nb, int G, int gstride)
{
    static_assert(LOG2N == 13, "pair-of-waves y-pass: N = 8192");
    using W = WaveSq<6>;
    constexpr int S = 64, N = 1 << LOG2N;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int par = wv & 1, pair = wv >> 1;
    float* mat_own = smem + wv * W::LDS_FLOATS;
    const float* mat_other = smem + (wv ^ 1) * W::LDS_FLOATS;
    typename W::LaneTwiddles tw;
    W::load_lane_twiddles(tw, twtab, lane + S * par, 1);

    static_assert(TC == 4 || TC == 8, "T tiles are 4 or 8 columns wide");
    const int qx = wave_first_column<TC, 2>(blockIdx.x) + pair;
    const int tile = qx / TC, col = qx & (TC - 1);
    const int plane = blockIdx.y / G, grp = blockIdx.y - plane * G;      // see k_ypass_acc
    Tbuf += (size_t)plane * nb * g.t_point;
    slab += (size_t)plane * gstride * g.nt * 4 * g.pn;
    const bool active = tile * TC < g.pn;
    float acc[S / 2];
    static_for<0, S / 2>([&](auto i) { acc[i] = 0.f; });

    // slot j <-> sample n = lane + 64 par + 128 j: k = n for j <= 8, k = n - N for j >= 56; T row a = k - ky0.
    constexpr int RB = 8 * TC;                                            // bytes per T row inside a tile
    const unsigned tile_bytes = active ? (unsigned)g.rows * RB : 0u;
    const unsigned vb = (unsigned)(lane + S * par - g.ky0) * RB + (unsigned)col * 8u;
    auto slot_off = [&](int j) { return vb + (unsigned)(j <= 8 ? RB * 2 * S * j : RB * 2 * S * j - RB * N); };

    for (int s = grp; s < nb; s += G) {
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point + (size_t)(active ? tile : 0) * g.rows * TC, tile_bytes);
        float2 x[S];
        static_for<0, S>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j <= 8 || j >= S - 8) x[j] = buf_load_c64(rT, slot_off(j));
            else x[j] = make_float2(0.f, 0.f);
        });
        W::run_pair(x, tw, mat_own, mat_other, lane, par);
        static_for<0, S / 2>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            constexpr int k2 = i < S / 4 ? i : S / 2 + i;
            const float2 v = x[W::brev(k2)];
            acc[i] = fmaf(v.x, v.x, fmaf(v.y, v.y, acc[i]));
        });
    }

    if (!active || qx >= g.pn) return;
    float* srow = slab + ((size_t)grp * g.nt * 4 + qx) * g.pn;
    static_for<0, S / 2>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int k2 = i < S / 4 ? i : S / 2 + i;
        constexpr int ubase = k2 < S / 2 ? 2 * S * k2 : 2 * S * k2 - N;      // bin u = lane + 64 par + 128 k2 (mod N, centred)
        srow[ubase + lane + S * par + g.c] += acc[i];
    });
}

// ----------------------------------------------------------------------------------
// y-pass for N = 512, 1024, 2048 with pn = N/2 (BASELINE configs 1 and 2), pupil inside the unit disk: NL = 8, 4, 2
// ADJACENT columns per wave (WaveSq<6>::run_rect: 64 lanes x 64/NL slots per line).  A row of the NL columns is
// 8 NL contiguous bytes of a T tile: NL/2 16-byte loads.  Workgroup = 4 waves = 4 NL columns.
// ----------------------------------------------------------------------------------
// FULL = false: N = 2 pn.  FULL = true: N = pn, the COARSE-GRID transform -- the Abbe sum is accumulated on the
// pn x pn grid q = 2 v that spans the whole period (E_s(2v) = sum_k A_s[k] w_pn^(k v)): every bin is kept
// (64 accumulators per lane) and |k| <= pn/4 = N/4 of the inputs are live; abbe_engine.hip reconstructs the fine
// image from it (band-limited interpolation + the exact Nyquist-line correction).
// GW = groups per workgroup.  GW = 1: blockIdx.y = plane * G + group, every group flushes into its own slab.  GW = 2:
// a 512-thread workgroup holds BOTH groups of its column block (waves 0-3: group 2 gb, waves 4-7: group 2 gb + 1;
// blockIdx.y = plane * G/2 + gb); at the end the odd group hands its accumulators to the even one through its
// transpose matrix (64 x 64 floats fit), which adds them and flushes ONE slab (index gb): the read-modify-write of
// the slabs -- 67 MB per 12-item launch at 2048^2 against 201 MB of T -- is halved, at the same residency (8 waves
// per CU).
template <int LOG2N, int TC, bool FULL = false, int GW = 1>
__global__ __launch_bounds__(256 * GW, 2) void k_ypass_rect(
    const float2* __restrict__ Tbuf, float* __restrict__ slab, const float2* __restrict__ twtab,
    PassGeom g, int nb, int G, int gstride)
{
    static_assert(GW == 1 || GW == 2, "one or two groups per workgroup");
    static_assert(LOG2N >= 8 && LOG2N <= 11, "multi-column-per-wave y-pass: N = 256 (full output only), 512, 1024, 2048");
    static_assert(TC == 2 || TC == 4 || TC == 8, "T tiles are 2, 4 or 8 columns wide");
    static_assert(LOG2N >= 9 || FULL, "N = 256: the coarse-grid transform of 256^2 images");
    using W = WaveSq<6>;
    constexpr int S = 64, N = 1 << LOG2N, NL = (S * S) / N, H = S / NL;
    constexpr int JL = FULL ? H / 4 : H / 8;                 // live rows: |k| <= pn/4 = 64 JL
    constexpr int NACC = FULL ? S : S / 2;                   // kept bins per lane
    auto kept_k2 = [](int i) constexpr { return FULL ? i : (i < S / 4 ? i : S / 2 + i); };
    static_assert(NL <= TC || NL == 2 * TC, "the wave's columns sit in one T tile, or in two adjacent ones (N = 256)");
    constexpr int NT = NL <= TC ? 1 : 2;                     // tiles the wave's columns span
    constexpr int QT = NL / 2 / NT;                          // 16-byte column pairs per tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cw = wv & 3, gl = wv >> 2;                                 // column wave, group inside the workgroup
    float* lds = smem + wv * W::LDS_FLOATS;
    typename W::LaneTwiddles tw;
    W::load_lane_twiddles(tw, twtab, lane, 1);

    const int qx0 = wave_first_column<TC, 4 * NL>(blockIdx.x) + NL * cw;  // multiple of NL: the wave's columns sit in one tile
    const int tile = qx0 / TC, col = qx0 & (TC - 1);
    const int GB = G / GW;                                               // group blocks (= slabs written) per plane
    const int plane = blockIdx.y / GB, gb = blockIdx.y - plane * GB;     // see k_ypass_acc
    const int grp = gb * GW + gl;
    Tbuf += (size_t)plane * nb * g.t_point;
    slab += (size_t)plane * gstride * g.nt * 4 * g.pn;
    const bool active = tile * TC < g.pn;
    float acc[NACC];
    static_for<0, NACC>([&](auto i) { acc[i] = 0.f; });

    // slot j (of every line) <-> sample n = lane + 64 j: k = n for j <= JL, k = n - N for j >= H - JL; T row a = k - ky0
    constexpr int RB = 8 * TC;
    const unsigned tile_bytes = active ? (unsigned)g.rows * RB : 0u;
    const unsigned vb = (unsigned)(lane - g.ky0) * RB + (unsigned)col * 8u;
    auto slot_off = [&](int j) { return vb + (unsigned)(j <= JL ? RB * S * j : RB * S * j - RB * N); };

    for (int s = grp; s < nb; s += G) {
        __amdgpu_buffer_rsrc_t rTs[NT];
        static_for<0, NT>([&](auto t_) {
            constexpr int t = decltype(t_)::value;
            rTs[t] = make_rsrc(Tbuf + (size_t)s * g.t_point + (size_t)(active ? tile + t : 0) * g.rows * TC, tile_bytes);
        });
        float2 x[S];
        static_for<0, H>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            static_for<0, NL / 2>([&](auto q_) {
                constexpr int q = decltype(q_)::value;
                if constexpr (j <= JL || j >= H - JL) {
                    const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rTs[q / QT], slot_off(j) + 16u * (q % QT), 0, LITHO_YLOAD_AUX);
                    x[(2 * q) * H + j] = make_float2(__uint_as_float(v.x), __uint_as_float(v.y));          // column qx0 + 2q
                    x[(2 * q + 1) * H + j] = make_float2(__uint_as_float(v.z), __uint_as_float(v.w));      // column qx0 + 2q + 1
                } else {
                    x[(2 * q) * H + j] = make_float2(0.f, 0.f);
                    x[(2 * q + 1) * H + j] = make_float2(0.f, 0.f);
                }
            });
        });
        W::template run_rect<NL>(x, tw, lds, lane);
        static_for<0, NACC>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            constexpr int k2 = kept_k2(i);
            const float2 v = x[W::brev(k2)];
            acc[i] = fmaf(v.x, v.x, fmaf(v.y, v.y, acc[i]));
        });
    }

    if constexpr (GW == 2) {
        // the odd group's accumulators -> its own (now idle) transpose matrix, [i][lane]; the even group adds them
        if (gl == 1) static_for<0, NACC>([&](auto i_) { constexpr int i = decltype(i_)::value; lds[i * 64 + lane] = acc[i]; });
        __syncthreads();
        if (gl == 1) return;
        const float* other = smem + (wv + 4) * W::LDS_FLOATS;
        static_for<0, NACC>([&](auto i_) { constexpr int i = decltype(i_)::value; acc[i] += other[i * 64 + lane]; });
    }
    const int qx = qx0 + lane / H, m = lane & (H - 1);
    if (!active || qx >= g.pn) return;
    float* srow = slab + ((size_t)gb * g.nt * 4 + qx) * g.pn;
    static_for<0, NACC>([&](auto i_) {
        constexpr int i = decltype(i_)::value;
        constexpr int k2 = kept_k2(i);
        constexpr int ubase = k2 < S / 2 ? H * k2 : H * k2 - N;             // bin u = m + H k2 (mod N, centred)
#if defined(LITHO_DIAG_YFLUSH_STORE)                                          // timing diagnostics (wrong results): a write-only flush,
        srow[ubase + m + g.c] = acc[i];
#elif defined(LITHO_DIAG_YFLUSH_NONE)                                         // and none at all (accumulators that never leave the registers)
        if (acc[i] == 12345.678f) srow[ubase + m + g.c] = acc[i];
#else
        srow[ubase + m + g.c] += acc[i];
#endif
    });
}


// ----------------------------------------------------------------------------------
// x-pass on the wave-level engine for N = 512, 1024, 2048 with pn = N/2, pupil inside the unit disk, no wrapping
// shift, 8-column T tiles: NL = 8, 4, 2 ADJACENT ROWS of the pupil box per wave (WaveSq<6>::run_rect).  Work item of
// a wave = (group of NL rows, source point); it walks a chunk of source points for its row group.  Per item: the
// live pupil and mask-spectrum samples of the NL rows are loaded through windowed descriptors (validity = hardware
// range check) and multiplied, run_rect transforms the NL rows at once, and every store instruction then holds NL
// rows x 64/NL consecutive columns: rows 2i, 2i+1 of an 8-column tile are adjacent 64-byte runs, so each
// instruction writes WHOLE 128-byte lines (measured 4.2-4.3 TB/s against 3.4 for the 64-byte granules of the
// one-row-per-workgroup x-pass, scripts/ubench/write_bw.hip L0/L1).  No workgroup barriers.
// ----------------------------------------------------------------------------------
template <int LOG2N, bool FULL = false>
__global__ __launch_bounds__(256, 2) void k_xpass_rect(
    const float2* __restrict__ P, const float2* __restrict__ M, const int* __restrict__ shifts,
    float2* __restrict__ Tbuf, const float2* __restrict__ twtab, PassGeom g, int nb, int chunk)
{
    static_assert(LOG2N >= 9 && LOG2N <= 11, "multi-row-per-wave x-pass: N = 512, 1024, 2048");
    using W = WaveSq<6>;
    // FULL: N = pn (coarse-grid transforms): half of the samples live, every bin kept
    constexpr int S = 64, N = 1 << LOG2N, NL = (S * S) / N, H = S / NL, JL = FULL ? H / 4 : H / 8, TC = 8;
    constexpr int NOUT = FULL ? S : S / 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* lds = smem + wv * W::LDS_FLOATS;
    typename W::LaneTwiddles tw;
    W::load_lane_twiddles(tw, twtab, lane, 1);

    const int a0 = (blockIdx.x * 4 + wv) * NL;                  // first box row of this wave's group (multiple of NL)
    const int s_begin = blockIdx.y * chunk;
    const int s_end = min(nb, s_begin + chunk);
    if (a0 >= g.rows) return;                                   // whole wave idle (no barriers in this kernel)

    // input: slot (line, j) <-> row a0 + line, sample n = lane + 64 j: k = n for j <= JL, k = n - N for j >= H - JL.
    // Descriptors are windowed on the valid columns [kx0, kx1) of each row (a left-of-window slot wraps to a huge
    // unsigned offset), so every slot address is one register + a compile-time constant.
    const unsigned win_bytes = (unsigned)(g.kx1 - g.kx0) * 8u;
    const unsigned vb = (unsigned)(lane - g.kx0) * 8u;
    auto slot_off = [&](int j) { return vb + (unsigned)(j <= JL ? 512 * j : 512 * j - 8 * N); };
    const int r0 = g.ky0 + g.c + a0;                            // row of P (and, shifted, of M) for line 0

    // output: lane c = (line = c / H, m = c % H) holds X_line[m + H k2]; column q = u + c with u = m + H k2 (k2 < 16)
    // or m + H k2 - N (k2 >= 48).  H k2 + c is a multiple of 8, so tile = (m >> 3) + (H k2 + c) / 8, column in tile = m & 7.
    const int oline = lane / H, om = lane & (H - 1);
    const bool ovalid = a0 + oline < g.rows;
    const unsigned rowstride = (unsigned)g.rows * (TC * 8u);    // bytes between consecutive tiles
    const unsigned obase = ovalid ? (unsigned)(om >> 3) * rowstride + (unsigned)(a0 + oline) * (TC * 8u) + (unsigned)(om & 7) * 8u
                                  : BUF_OOB;

    for (int s = s_begin; s < s_end; ++s) {
        const int dy = shifts[2 * s], dx = shifts[2 * s + 1];
        float2 x[S];
        static_for<0, NL>([&](auto q_) {
            constexpr int line = decltype(q_)::value;
            const bool rvalid = a0 + line < g.rows;
            const __amdgpu_buffer_rsrc_t rP =
                make_rsrc(P + (size_t)(r0 + (rvalid ? line : 0)) * g.pn + g.c + g.kx0, rvalid ? win_bytes : 0u);
            const __amdgpu_buffer_rsrc_t rM =
                make_rsrc(M + (size_t)(r0 + (rvalid ? line : 0) + dy) * g.pn + dx + g.c + g.kx0, rvalid ? win_bytes : 0u);
            static_for<0, H>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                if constexpr (j <= JL || j >= H - JL) x[line * H + j] = cmul(buf_load_c64(rP, slot_off(j)), buf_load_c64(rM, slot_off(j)));
                else x[line * H + j] = make_float2(0.f, 0.f);
            });
        });
        W::template run_rect<NL>(x, tw, lds, lane);
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point, (size_t)g.t_point * sizeof(float2));
        static_for<0, NOUT>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            constexpr int k2 = FULL ? i : (i < S / 4 ? i : S / 2 + i);
            constexpr int ub = k2 < S / 2 ? H * k2 : H * k2 - N;            // u - m
            const unsigned tile_off = (unsigned)((ub + g.c) >> 3) * rowstride;
            buf_store_c64(rT, obase == BUF_OOB ? BUF_OOB : obase + tile_off, x[W::brev(k2)]);
        });
    }
}

template <int LOG2N>
hipError_t launch_xpass_rect(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                             const PassGeom& g, int nb, int chunk, hipStream_t st)
{
    if constexpr (LOG2N >= 9 && LOG2N <= 11) {
        if (g.tcl != 3) return hipErrorNotSupported;
        static LdsOnce once;
        constexpr size_t lds = 4 * WaveSq<6>::LDS_FLOATS * sizeof(float);
        constexpr int NL = 4096 >> LOG2N;
        const int groups = (g.rows + NL - 1) / NL;
        const dim3 grid((groups + 3) / 4, (nb + chunk - 1) / chunk);
        if (g.N == g.pn) {
            static LdsOnce once_full;
            auto kern = k_xpass_rect<LOG2N, true>;
            hipError_t e = set_lds(once_full, kern, lds);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, P, M, shifts, T, tw, g, nb, chunk);
            note_kernel(0, "k_xpass_rect<%d, true>", LOG2N);
            return hipGetLastError();
        }
        auto kern = k_xpass_rect<LOG2N>;
        hipError_t e = set_lds(once, kern, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, P, M, shifts, T, tw, g, nb, chunk);
        note_kernel(0, "k_xpass_rect<%d, false>", LOG2N);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int LOG2N, int TC>
static hipError_t launch_ypass_wave_tc(const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int planes, int G,
                     int gstride, hipStream_t st)
{
    constexpr size_t lds4 = 4 * WaveSq<6>::LDS_FLOATS * sizeof(float);       // four waves, one 64 x 65 matrix each
    if (g.N == g.pn) {                                  // coarse-grid transforms (N = pn): every bin kept
        if constexpr (LOG2N == 12) {
            static LdsOnce once;
            constexpr size_t ldsf = lds4 + (size_t)WaveSq<6>::TW_LDS_FLOAT2 * sizeof(float2);     // + the lane-twiddle table
            auto kern = k_ypass_wave<LOG2N, TC, true>;
            hipError_t e = set_lds(once, kern, ldsf);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(wave_grid_x<TC, 4>(g.pn), planes * G), dim3(256), ldsf, st, T, slab, tw, g, nb, G,
                               gstride);
            note_kernel(1, "k_ypass_wave<%d, %d, true>", LOG2N, TC);
            return hipGetLastError();
        } else if constexpr (LOG2N >= 8 && LOG2N <= 11) {
            constexpr int NL = 4096 >> LOG2N;
            if constexpr (NL <= TC || NL == 2 * TC) {
                if (g.gcombine && G % 2 == 0) {             // both groups of a column block in one 512-thread workgroup
                    static LdsOnce once2;
                    auto kern2 = k_ypass_rect<LOG2N, TC, true, 2>;
                    hipError_t e2 = set_lds(once2, kern2, 2 * lds4);
                    if (e2 != hipSuccess) return e2;
                    hipLaunchKernelGGL(kern2, dim3(wave_grid_x<TC, 4 * NL>(g.pn), planes * (G / 2)), dim3(512), 2 * lds4, st,
                                       T, slab, tw, g, nb, G, gstride);
                    note_kernel(1, "k_ypass_rect<%d, %d, true, 2>", LOG2N, TC);
                    return hipGetLastError();
                }
                static LdsOnce once;
                auto kern = k_ypass_rect<LOG2N, TC, true>;
                hipError_t e = set_lds(once, kern, lds4);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, dim3(wave_grid_x<TC, 4 * NL>(g.pn), planes * G), dim3(256), lds4, st, T, slab, tw,
                                   g, nb, G, gstride);
                note_kernel(1, "k_ypass_rect<%d, %d, true, 1>", LOG2N, TC);
                return hipGetLastError();
            }
        }
        return hipErrorNotSupported;
    }
    if constexpr (LOG2N == 13) {
        static LdsOnce once;
        auto kern = k_ypass_pair<LOG2N, TC>;
        hipError_t e = set_lds(once, kern, lds4);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(wave_grid_x<TC, 2>(g.pn), planes * G), dim3(256), lds4, st, T, slab, tw, g, nb, G,
                           gstride);
        note_kernel(1, "k_ypass_pair<%d, %d>", LOG2N, TC);
            return hipGetLastError();
    } else if constexpr (LOG2N >= 9 && LOG2N <= 11) {
        constexpr int NL = 4096 >> LOG2N;                       // columns per wave of k_ypass_rect
        if constexpr (NL <= TC) {
            if (!g.rect_off) {
                if (g.gcombine && G % 2 == 0) {
                    static LdsOnce once2;
                    auto kern2 = k_ypass_rect<LOG2N, TC, false, 2>;
                    hipError_t e2 = set_lds(once2, kern2, 2 * lds4);
                    if (e2 != hipSuccess) return e2;
                    hipLaunchKernelGGL(kern2, dim3(wave_grid_x<TC, 4 * NL>(g.pn), planes * (G / 2)), dim3(512), 2 * lds4, st,
                                       T, slab, tw, g, nb, G, gstride);
                    note_kernel(1, "k_ypass_rect<%d, %d, false, 2>", LOG2N, TC);
                    return hipGetLastError();
                }
                static LdsOnce once;
                auto kern = k_ypass_rect<LOG2N, TC>;
                hipError_t e = set_lds(once, kern, lds4);
                if (e != hipSuccess) return e;
                hipLaunchKernelGGL(kern, dim3(wave_grid_x<TC, 4 * NL>(g.pn), planes * G), dim3(256), lds4, st, T, slab, tw,
                                   g, nb, G, gstride);
                note_kernel(1, "k_ypass_rect<%d, %d, false, 1>", LOG2N, TC);
                return hipGetLastError();
            }
        }
        if constexpr (LOG2N >= 10) {
            static LdsOnce once;
            using WS = WaveShape<LOG2N>;
            auto kern = k_ypass_wave<LOG2N, TC>;
            hipError_t e = set_lds(once, kern, WS::LDS_BYTES);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(wave_grid_x<TC, WS::COLS>(g.pn), planes * G), dim3(WS::THREADS),
                               WS::LDS_BYTES, st, T, slab, tw, g, nb, G, gstride);
            note_kernel(1, "k_ypass_wave<%d, %d, false>", LOG2N, TC);
            return hipGetLastError();
        } else {
            return hipErrorNotSupported;
        }
    } else if constexpr (LOG2N == 12) {
        static LdsOnce once;
        using WS = WaveShape<LOG2N>;
        auto kern = k_ypass_wave<LOG2N, TC>;
        hipError_t e = set_lds(once, kern, WS::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(wave_grid_x<TC, WS::COLS>(g.pn), planes * G), dim3(WS::THREADS), WS::LDS_BYTES,
                           st, T, slab, tw, g, nb, G, gstride);
        note_kernel(1, "k_ypass_wave<%d, %d, false>", LOG2N, TC);
        return hipGetLastError();
    } else {
        return hipErrorNotSupported;
    }
}

template <int LOG2N>
hipError_t launch_ypass_wave(const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int planes,
                             int G, int gstride, hipStream_t st)
{
    if (g.tcl == 2) return launch_ypass_wave_tc<LOG2N, 4>(T, slab, tw, g, nb, planes, G, gstride, st);
    if (g.tcl == 3) return launch_ypass_wave_tc<LOG2N, 8>(T, slab, tw, g, nb, planes, G, gstride, st);
    if constexpr (LOG2N == 12) {
        if (g.tcl == 4 && g.N == g.pn) {                       // 16-column tiles: the cooperative-loading kernel
#ifndef LITHO_COOP_WAVES
#define LITHO_COOP_WAVES 4
#endif
            constexpr int NW = LITHO_COOP_WAVES;
            if (g.coop_dma) {
#ifndef LITHO_COOP_DMA_WAVES                     // build-time experiment: 8 = eight waves own 64 bytes of the row (one 144 KB workgroup per CU)
#define LITHO_COOP_DMA_WAVES 4
#endif
                constexpr int ND = LITHO_COOP_DMA_WAVES;
                static LdsOnce once_dma;
                auto kd = k_ypass_coop_dma<LOG2N, ND>;
                hipError_t ed = set_lds(once_dma, kd, CoopDmaShape<ND>::LDS_BYTES);
                if (ed != hipSuccess) return ed;
                hipLaunchKernelGGL(kd, dim3(CoopDmaShape<ND>::grid_x(g.pn), planes * G), dim3(64 * ND), CoopDmaShape<ND>::LDS_BYTES, st, T,
                                   slab, tw, g, nb, G, gstride);
                note_kernel(1, "k_ypass_coop_dma<%d, %d>", LOG2N, ND);
                return hipGetLastError();
            }
            static LdsOnce once;
            auto kern = k_ypass_coop<LOG2N, NW>;
            hipError_t e = set_lds(once, kern, CoopShape<NW>::LDS_BYTES);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(CoopShape<NW>::grid_x(g.pn), planes * G), dim3(64 * NW), CoopShape<NW>::LDS_BYTES, st, T,
                               slab, tw, g, nb, G, gstride);
            note_kernel(1, "k_ypass_coop<%d, %d>", LOG2N, NW);
            return hipGetLastError();
        }
    }
    return hipErrorNotSupported;
}

#define LITHO_DEFINE_WAVE_OPS(L2)                                                                            \
    template hipError_t launch_ypass_wave<L2>(const float2*, float*, const float2*, const PassGeom&, int, int, int, \
                                              int, hipStream_t);                                             \
    template hipError_t launch_xpass_rect<L2>(const float2*, const float2*, const int*, float2*, const float2*,  \
                                              const PassGeom&, int, int, hipStream_t);

}  // namespace litho
