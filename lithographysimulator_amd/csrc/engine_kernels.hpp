// engine_kernels.hpp -- the two pass kernels of the Abbe engine and their launchers.
//
// Both passes run the per-line FFT of fft_core.hpp; thread t of a line holds the 16
// samples n = t + T*e ("slots" e = 0..15).  Which slots can be non-zero on input and which
// are kept on output depends only on the window sizes relative to N, so for the shapes that
// matter (pn a power of two, pupil support inside the unit-radius disk, N/pn = 1, 2 or 4) the
// slot sets are compile-time template parameters: empty slots are literal zeros that the
// compiler folds through the first butterfly stage (file is built with -fno-signed-zeros),
// discarded outputs are dead code, and the per-slot arrays only have the live slots.
#pragma once
#include <hip/hip_runtime.h>

#include "engine_common.hpp"
#include "abbe_plan.hpp"
#include "fft_core.hpp"
#include "wave_fft.hpp"

namespace litho {

template <int LOG2N>
struct Launch {
    using F = LineFFT<LOG2N, +1>;
    static constexpr int L = (F::T >= 64) ? 1 : 64 / F::T;       // lines per workgroup
    static constexpr int THREADS = F::T * L;
#ifndef LITHO_NBUF
#define LITHO_NBUF 1
#endif
    static constexpr int NBUF = (LOG2N <= 12) ? LITHO_NBUF : 1;
    static constexpr size_t LDS_EXCH = (size_t)L * NBUF * F::LDS_LINE;          // float2 slots of the exchange buffers
    static constexpr size_t LDS_BYTES = sizeof(float2) * (LDS_EXCH + F::LDS_TW);   // + the twiddle tables
    // launch_bounds second argument = waves per SIMD we want resident: LITHO_WG_PER_CU workgroups per CU
    // up to N = 4096 (256 threads each), one above.  Measured at 2048^2: one LDS buffer (extra barrier) with 3
    // workgroups per CU beats two buffers with 2 by 7 %.
#ifndef LITHO_WG_PER_CU
#define LITHO_WG_PER_CU 3
#endif
    // N = 8192 (512 threads, 78 KB LDS): two workgroups per CU, otherwise every barrier idles the CU
    // (measured at 4096^2: y-pass 96 -> 58 us/point).  N = 16384 needs 147 KB LDS: one.
    static constexpr int WG_PER_CU = LOG2N <= 12 ? LITHO_WG_PER_CU : (LOG2N == 13 ? 2 : 1);
    static constexpr int WAVES = (THREADS / 256 > 0 ? THREADS / 256 : 1) * WG_PER_CU;
    // (Software prefetch of the next line's inputs was measured SLOWER on gfx950 -- the extra live
    // registers spill, 35.9 vs 28.7 us/point at 2048^2 -- and is not implemented.)
};

// (PassGeom and the slot sets natural_in_mask / out_mask: abbe_plan.hpp, shared with the host-only planner)

// ----------------------------------------------------------------------------------
// buffer addressing: 32-bit offsets, and the hardware range check is the zero-padding
// predicate (an offset >= num_records loads 0 / drops the store).
// ----------------------------------------------------------------------------------
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
static constexpr unsigned BUF_OOB = 0xFFFF0000u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float2 buf_load_c64(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
    return make_float2(__uint_as_float(v.x), __uint_as_float(v.y));
}
// Cache policy of the T stores (buffer aux bits: sc0 = 1, nt = 2, sc1 = 16).  With sc1 the L2 writes T through instead of
// keeping dirty lines nobody on its XCD will ask for again (the y-pass reads T through other L2s).  Same-box A/B, sc1 against
// none: config 3 (2048^2) 309.0 vs 322.3 ms per 36,000 points (x-pass 49.9 vs 54.8 us per launch, y-pass unchanged), config 4
// 208.0 vs 211.3; config 2 (1024^2) 195.2 vs 193.4 and config 1 equal -- so: from 2048-point rows up.  sc0 alone changes
// nothing, nt costs 18 %.  profiles/r03_xstore_policy.txt.  LITHO_XSTORE_AUX overrides the bits of the large sizes.
#ifndef LITHO_XSTORE_AUX
#define LITHO_XSTORE_AUX 16
#endif
template <int LOG2N> inline constexpr int t_store_aux = LOG2N >= 11 ? LITHO_XSTORE_AUX : 0;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int AUX = 0>
__device__ __forceinline__ void buf_store_c64x2(__amdgpu_buffer_rsrc_t r, unsigned off, float2 a, float2 b) {
    u32x4 w;
    w.x = __float_as_uint(a.x); w.y = __float_as_uint(a.y);
    w.z = __float_as_uint(b.x); w.w = __float_as_uint(b.y);
    __builtin_amdgcn_raw_buffer_store_b128(w, r, off, 0, AUX);
}
template <int AUX = 0>
__device__ __forceinline__ void buf_store_c64(__amdgpu_buffer_rsrc_t r, unsigned off, float2 v) {
    u32x2 w;
    w.x = __float_as_uint(v.x);
    w.y = __float_as_uint(v.y);
    __builtin_amdgcn_raw_buffer_store_b64(w, r, off, 0, AUX);
}

// float2 offset of (row a, column q) inside one source point's T block
__device__ __forceinline__ unsigned t_offset(const PassGeom& g, unsigned a, unsigned q) {
    return ((((q >> g.tcl) * g.rows + a) << g.tcl) + (q & ((1u << g.tcl) - 1u)));
}

typedef unsigned int uint2v __attribute__((ext_vector_type(2)));

// an empty asm that "uses and redefines" a register pair: pins where the compiler must have waited for a load
__device__ __forceinline__ void touch_vgpr(float2& v) { asm volatile("" : "+v"(v.x), "+v"(v.y)); }
__device__ __forceinline__ void diag_keep(float2 v, unsigned o) { asm volatile("" ::"v"(v.x), "v"(v.y), "v"(o)); }

// ----------------------------------------------------------------------------------
// x-pass (no wrapping shift): A = P[box] * M[box + shift], rows of the support box ->
// T[s][tile][row][4].  One workgroup = one row of the box for a CHUNK of source points: the
// pupil row and the twiddles stay in registers, only the mask-spectrum window moves.
// ----------------------------------------------------------------------------------
// RP = 2 (row pairs, opt-in: LITHO_ABBE_ROWPAIRS): the workgroup takes rows 2a and 2a + 1, one after the other for every
// source point, and stores them blended so that every store instruction writes whole 128-byte lines of 8-column tiles
// (two rows per line).  Measured at 4096^2 (T streams through HBM), us per item: one row per workgroup 21.6; pairs with
// the two rows' stores merely interleaved 19.8, half-wave by half-wave (v_permlane32_swap) 20.6, blended 19.6 -- against
// 14.0 for one row per workgroup on 16-column tiles, which is what the planner picks there.  Costs a second pupil row and
// a held output row in registers: two workgroups per CU instead of three.
// (Those figures are with plain stores; since the T stores carry sc1 -- no write-allocate fetch -- one row per workgroup on
// 8-column tiles takes 19.6 us per item itself and the pairs 19.8: the knob stays as a parity-tested variant, nothing more.)
template <int LOG2N, int RL, bool PRUNED, int NP, int RP = 1>
__global__ __launch_bounds__(Launch<LOG2N>::THREADS, (RP == 2 ? Launch<LOG2N>::WAVES / Launch<LOG2N>::WG_PER_CU * 2 : Launch<LOG2N>::WAVES)) void k_xpass_abbe(
    const float2* __restrict__ P, const float2* __restrict__ M, const int* __restrict__ shifts,
    float2* __restrict__ Tbuf, const float2* __restrict__ twtab, PassGeom g, int nb, int chunk)
{
    // NP = planes of a through-focus stack handled by this workgroup (P points at the first of them,
    // planes are pn*pn apart).  The mask-spectrum window of a source point is gathered ONCE and multiplied by
    // the NP pupil rows held in registers; T item (plane p, point s) = p * nb + s.
    using F = LineFFT<LOG2N, +1>;
    using LC = Launch<LOG2N>;
    constexpr unsigned IN = PRUNED ? natural_in_mask(RL) : 0xFFFFu;
    constexpr unsigned OUT = out_mask(RL);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    const int lt = threadIdx.x % F::T, lg = threadIdx.x / F::T;
    float2* lds = smem + (size_t)lg * LC::NBUF * F::LDS_LINE;

    typename F::Twiddles tw;
    F::load_twiddles(tw, twtab, lt, smem + LC::LDS_EXCH, threadIdx.x, LC::THREADS);

    // XCD-aware row mapping.  A 128-byte line of T holds 4 consecutive rows x 4 columns, and the
    // dispatcher deals consecutive workgroups round-robin over the 8 XCDs (private L2s): with the
    // identity mapping the four quarters of every line are written from four different L2s and each
    // one evicts a partial line (measured: 3.7 of 8.5 us/point).  Blocks b, b+8, b+16, b+24 run on the
    // same XCD back to back, so they get rows 4j..4j+3 and the line is completed inside one L2.
    static_assert(RP == 1 || (RP == 2 && NP == 1 && LC::L == 1), "row pairs: one plane, one line per workgroup");
    int a;
    if constexpr (LC::L == 1) {
        const int b = blockIdx.x, xcd = b & 7, i = b >> 3;
        a = ((i >> 2) * 32 + xcd * 4 + (i & 3)) * RP;
    } else {
        a = blockIdx.x * LC::L + lg;
    }
    const bool active = a < g.rows;
    const bool active1 = a + 1 < g.rows;                      // RP = 2: the second row of the pair exists
    const int r = g.ky0 + g.c + a;                            // row of P inside its support box
    const size_t plane_bytes = (size_t)g.pn * g.pn * sizeof(float2);
    const __amdgpu_buffer_rsrc_t rM = make_rsrc(M, plane_bytes);

    unsigned koff[16];           // (c + k) of slot e, or BUF_OOB when outside the window
    float2 pv[NP * RP][16];      // RP = 2: pv[1] is the pupil row r + 1
    static_for<0, 16>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
        if constexpr ((IN >> e) & 1u) {
            int k;
            const bool ok = centred_index(lt + F::T * e, F::N, g.kx0, g.kx1, k) && active;
            koff[e] = ok ? (unsigned)(g.c + k) : BUF_OOB;
        }
    });
    static_for<0, NP * RP>([&](auto p_) {
        constexpr int p = decltype(p_)::value;
        const __amdgpu_buffer_rsrc_t rP = make_rsrc(P + (size_t)(RP == 1 ? p : 0) * g.pn * g.pn, plane_bytes);
        static_for<0, 16>([&](auto e_) {
            constexpr int e = decltype(e_)::value;
            if constexpr ((IN >> e) & 1u)
                pv[p][e] = buf_load_c64(rP, koff[e] != BUF_OOB ? ((unsigned)(r + (RP == 1 ? 0 : p)) * g.pn + koff[e]) * 8u : BUF_OOB);
        });
    });
    unsigned toff[16];           // byte offset of output bin m inside one T item
    static_for<0, 16>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        if constexpr ((OUT >> m) & 1u) {
            int u;
            const bool ok = centred_index(lt + F::T * m, F::N, -g.c, g.pn - g.c, u) && active;
            const unsigned q = (unsigned)(u + g.c);
            toff[m] = ok ? t_offset(g, a, q) * 8u : BUF_OOB;
        }
    });

    const int s_begin = blockIdx.y * chunk;
    const int s_end = min(nb, s_begin + chunk);
    auto load_window = [&](int s, int rr, float2 (&mv)[16]) {
        const int dy = shifts[2 * s], dx = shifts[2 * s + 1];
        const unsigned mrow = (unsigned)(r + rr + dy) * g.pn + dx;     // same window of M moved by the shift
        static_for<0, 16>([&](auto e_) {
            constexpr int e = decltype(e_)::value;
            if constexpr ((IN >> e) & 1u)
#ifdef LITHO_DIAG_XNOLOAD
                mv[e] = make_float2((float)((mrow + koff[e]) & 1023u), 1.0f);
#else
                mv[e] = buf_load_c64(rM, koff[e] != BUF_OOB ? (mrow + koff[e]) * 8u : BUF_OOB);
#endif
        });
    };

    int flip = 0;
    float2 mnext[16];
    if (s_begin < s_end) load_window(s_begin, 0, mnext);
    // Consume the first window HERE, so that the compiler waits for it in the preheader.  Otherwise the loop header
    // inherits "loads may be pending" from the entry edge and opens every iteration with s_waitcnt vmcnt(0) -- which,
    // with vmcnt shared between loads and stores on gfx950, also waits for the previous iteration's eight T stores
    // to be acknowledged (measured: 2 of 6.5 us per source point at 2048^2).
    static_for<0, 16>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
        if constexpr ((IN >> e) & 1u) touch_vgpr(mnext[e]);
    });
    // RP = 2: row a + 1 sits one tile row further (the last pair of an odd box has no second row: out of range).
    // The two rows are stored BLENDED: with B' = row a + 1 rotated by 8 lanes inside every 16-lane row (one DPP move),
    // C = (lane & 8 ? B' : A) holds, in every 16 consecutive lanes, row a and row a + 1 of ONE 8-column tile -- a whole
    // 128-byte line (8-column tiles: rows 2i, 2i + 1 of a tile are one line) -- and D = (lane & 8 ? A : B') the same
    // for the odd tiles: every store instruction writes whole lines.  Offsets: C = toff + (hi ? row1 - tile : 0),
    // D = toff + (hi ? 0 : row1 + tile), tile = the byte distance of neighbouring tiles.
    const unsigned row1 = active1 ? (8u << g.tcl) : 0x80000000u;
    float2 held[RP == 2 ? 16 : 1];
    const bool hi8 = (threadIdx.x & 8) != 0;
    unsigned offc = 0, offd = 0;
    if constexpr (RP == 2) {
        const unsigned tileb = (unsigned)g.rows * (8u << g.tcl);
        offc = hi8 ? row1 - tileb : 0u;
        offd = hi8 ? 0u : row1 + tileb;
    }
    for (int s = s_begin; s < s_end; ++s) {
        static_for<0, RP>([&](auto rr_) {
            constexpr int rr = decltype(rr_)::value;
            float2 mv[16];
            static_for<0, 16>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                if constexpr ((IN >> e) & 1u) mv[e] = mnext[e];
            });
#ifndef LITHO_XPASS_NO_PREFETCH
            // the next window is in flight while this one's NP transforms run (the LDS-only
            // barriers of the FFT do not drain vmcnt); the last iteration re-reads its own window
            if constexpr (rr + 1 < RP) load_window(s, rr + 1, mnext);
            else load_window(s + 1 < s_end ? s + 1 : s, 0, mnext);
#else
            if constexpr (rr + 1 < RP) load_window(s, rr + 1, mnext);
            else if (s + 1 < s_end) load_window(s + 1, 0, mnext);
#endif
            static_for<0, NP>([&](auto p_) {
                constexpr int p = decltype(p_)::value;
                float2 x[16];
                static_for<0, 16>([&](auto e_) {
                    constexpr int e = decltype(e_)::value;
                    if constexpr ((IN >> e) & 1u) x[e] = cmul(pv[RP == 1 ? p : rr][e], mv[e]);
                    else x[e] = make_float2(0.f, 0.f);
                });
                F::template run<LC::NBUF>(x, tw, lds, lt, flip);
                const __amdgpu_buffer_rsrc_t rT =
                    make_rsrc(Tbuf + ((size_t)p * nb + s) * g.t_point, (size_t)g.t_point * sizeof(float2));
                if constexpr (RP == 2 && rr == 0) {
                    // keep row 2a's line: it is stored together with row 2a + 1's, half line next to half line
                    static_for<0, 16>([&](auto m_) { held[decltype(m_)::value] = x[decltype(m_)::value]; });
                } else {
                    static_for<0, 16>([&](auto m_) {
                        constexpr int m = decltype(m_)::value;
#ifdef LITHO_DIAG_XNOSTORE
                        if constexpr ((OUT >> m) & 1u) diag_keep(x[m], toff[m]);
#elif defined(LITHO_DIAG_XSTORE_L2)
                        if constexpr ((OUT >> m) & 1u) buf_store_c64(rT, toff[m] == BUF_OOB ? BUF_OOB : (toff[m] & 0xFFFFFu), x[m]);
#else
                        if constexpr ((OUT >> m) & 1u) {
                            if constexpr (RP == 2) {
                                // (N = pn: every column exists; a padding workgroup's BUF_OOB + offd would wrap into range)
                                if (active) {
                                    const float bx = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x[m].x), 0x128, 0xF, 0xF, false));   // row_ror:8
                                    const float by = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(x[m].y), 0x128, 0xF, 0xF, false));
                                    buf_store_c64<t_store_aux<LOG2N>>(rT, toff[m] + offc, hi8 ? make_float2(bx, by) : held[m]);
                                    buf_store_c64<t_store_aux<LOG2N>>(rT, toff[m] + offd, hi8 ? held[m] : make_float2(bx, by));
                                }
                            } else {
                                buf_store_c64<t_store_aux<LOG2N>>(rT, toff[m], x[m]);
                            }
                        }
#endif
                    });
                }
            });
        });
    }
}

// ----------------------------------------------------------------------------------
// x-pass for N = 2 pn at the sizes where the N-point engine is the weak one (N = 8192: 512-thread workgroups,
// 128-register cap).  Decimation in frequency turns every row into TWO N/2-point transforms:
//     out[2v + p] = sum_k (in[k] w_N^(k p)) w_{N/2}^(k v),      p = 0, 1,   v in [-pn/4, pn/4),
// (the pn/2 + 1 live inputs do not alias in N/2 points), both run by the better-tuned N/2-point engine with the
// slot sets IN = natural_in_mask(0), OUT = out_mask(1).  The factor w_N^(k p) is folded into a second register copy
// of the pupil row, the mask-spectrum window is gathered once for both, and because the SAME thread ends up with
// the adjacent columns 2v and 2v + 1, T is written with 16-byte stores in its usual layout: the y-pass does not
// know the difference.
// ----------------------------------------------------------------------------------
template <int LOG2N>
__global__ __launch_bounds__(Launch<LOG2N - 1>::THREADS, Launch<LOG2N - 1>::WAVES) void k_xpass_split(
    const float2* __restrict__ P, const float2* __restrict__ M, const int* __restrict__ shifts,
    float2* __restrict__ Tbuf, const float2* __restrict__ twtab, PassGeom g, int nb, int chunk)
{
    using F = LineFFT<LOG2N - 1, +1>;
    using LC = Launch<LOG2N - 1>;
    static_assert(LC::L == 1, "split x-pass: one row per workgroup");
    constexpr unsigned IN = natural_in_mask(0), OUT = out_mask(1);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    const int lt = threadIdx.x;
    float2* lds = smem;

    typename F::Twiddles tw;
    F::load_twiddles(tw, twtab, lt, smem + LC::LDS_EXCH, threadIdx.x, LC::THREADS, 2);   // twtab = the N-point table

    const int b = blockIdx.x, xcd = b & 7, i = b >> 3;           // XCD-aware row mapping, see k_xpass_abbe
    const int a = (i >> 2) * 32 + xcd * 4 + (i & 3);
    const bool active = a < g.rows;
    const int r = g.ky0 + g.c + a;
    const size_t plane_bytes = (size_t)g.pn * g.pn * sizeof(float2);
    const __amdgpu_buffer_rsrc_t rM = make_rsrc(M, plane_bytes);
    const __amdgpu_buffer_rsrc_t rP = make_rsrc(P, plane_bytes);

    unsigned koff[16];
    float2 pv0[16], pv1[16];
    static_for<0, 16>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
        if constexpr ((IN >> e) & 1u) {
            int k;
            const bool ok = centred_index(lt + F::T * e, F::N, g.kx0, g.kx1, k) && active;
            koff[e] = ok ? (unsigned)(g.c + k) : BUF_OOB;
            pv0[e] = buf_load_c64(rP, ok ? ((unsigned)r * g.pn + koff[e]) * 8u : BUF_OOB);
            pv1[e] = cmul(pv0[e], twtab[(k + 2 * F::N) & (2 * F::N - 1)]);       // w_N^k, k may be negative
        }
    });
    unsigned toff[16];           // byte offset of the column PAIR (2v, 2v + 1) inside one T item
    static_for<0, 16>([&](auto m_) {
        constexpr int m = decltype(m_)::value;
        if constexpr ((OUT >> m) & 1u) {
            int v;
            const bool ok = centred_index(lt + F::T * m, F::N, -(g.c >> 1), g.c >> 1, v) && active;
            toff[m] = ok ? t_offset(g, a, (unsigned)(2 * v + g.c)) * 8u : BUF_OOB;
        }
    });

    const int s_begin = blockIdx.y * chunk;
    const int s_end = min(nb, s_begin + chunk);
    int flip = 0;
    for (int s = s_begin; s < s_end; ++s) {
        const int dy = shifts[2 * s], dx = shifts[2 * s + 1];
        const unsigned mrow = (unsigned)(r + dy) * g.pn + dx;
        float2 mv[16];
        static_for<0, 16>([&](auto e_) {
            constexpr int e = decltype(e_)::value;
            if constexpr ((IN >> e) & 1u) mv[e] = buf_load_c64(rM, koff[e] != BUF_OOB ? (mrow + koff[e]) * 8u : BUF_OOB);
        });
        float2 x[16], even[16];
        static_for<0, 16>([&](auto e_) {
            constexpr int e = decltype(e_)::value;
            if constexpr ((IN >> e) & 1u) x[e] = cmul(pv0[e], mv[e]);
            else x[e] = make_float2(0.f, 0.f);
        });
        F::template run<LC::NBUF>(x, tw, lds, lt, flip);
        static_for<0, 16>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            if constexpr ((OUT >> m) & 1u) even[m] = x[m];
        });
        static_for<0, 16>([&](auto e_) {
            constexpr int e = decltype(e_)::value;
            if constexpr ((IN >> e) & 1u) x[e] = cmul(pv1[e], mv[e]);
            else x[e] = make_float2(0.f, 0.f);
        });
        F::template run<LC::NBUF>(x, tw, lds, lt, flip);
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point, (size_t)g.t_point * sizeof(float2));
        static_for<0, 16>([&](auto m_) {
            constexpr int m = decltype(m_)::value;
            if constexpr ((OUT >> m) & 1u) buf_store_c64x2<t_store_aux<LOG2N>>(rT, toff[m], even[m], x[m]);
        });
    }
}

// ----------------------------------------------------------------------------------
// generic x-pass (any loader, runtime predication): rows -> T[s][tile][row][4]
// ----------------------------------------------------------------------------------
template <int LOG2N, int SIGN, typename Loader>
__global__ __launch_bounds__(Launch<LOG2N>::THREADS, Launch<LOG2N>::WAVES) void k_xpass(
    Loader ld, float2* __restrict__ Tbuf, const float2* __restrict__ twtab, PassGeom g)
{
    using F = LineFFT<LOG2N, SIGN>;
    using LC = Launch<LOG2N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    const int lt = threadIdx.x % F::T, lg = threadIdx.x / F::T;
    float2* lds = smem + (size_t)lg * LC::NBUF * F::LDS_LINE;

    typename F::Twiddles tw;
    F::load_twiddles(tw, twtab, lt, smem + LC::LDS_EXCH, threadIdx.x, LC::THREADS);

    const int s = blockIdx.y;
    const int a = blockIdx.x * LC::L + lg;
    const bool active = a < g.rows;
    ld.begin_line(s, active ? a : 0, g);

    float2 x[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int k;
        const bool ok = centred_index(lt + F::T * e, F::N, g.kx0, g.kx1, k) && active;
        x[e] = ok ? ld.load(k, g) : make_float2(0.f, 0.f);
    }
    int flip = 0;
    F::template run<LC::NBUF>(x, tw, lds, lt, flip);

    float2* tpt = Tbuf + (size_t)s * g.t_point;
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        int u;
        if (centred_index(lt + F::T * m, F::N, -g.c, g.pn - g.c, u) && active) {
            tpt[t_offset(g, a, (unsigned)(u + g.c))] = x[m];
        }
    }
}

// Loader: Abbe product P*M for source point s (imageformation.py:34 and :63), with the roll
// kept on P (modular gather) when a shifted support would wrap around the grid.
struct AbbeLoader {
    const float2* P;
    const float2* M;
    const int* shifts;       // (dy,dx) pairs of this batch
    const float2* prow;
    const float2* mrow;
    int dx, pad_;
    __device__ __forceinline__ void begin_line(int s, int a, const PassGeom& g) {
        const int dy = shifts[2 * s];
        dx = shifts[2 * s + 1];
        if (!g.general) {
            const int r = g.ky0 + g.c + a;                  // row of P (support box row)
            prow = P + (size_t)r * g.pn;
            mrow = M + (size_t)(r + dy) * g.pn + dx;        // same window of M, moved by the shift
        } else {
            int r = (a - dy) % g.pn;                        // torch.roll: A[i] = P[(i - d) mod pn]
            if (r < 0) r += g.pn;
            prow = P + (size_t)r * g.pn;
            mrow = M + (size_t)a * g.pn;
        }
    }
    __device__ __forceinline__ float2 load(int k, const PassGeom& g) const {
        const int col = k + g.c;
        if (!g.general) return cmul(prow[col], mrow[col]);
        int pc = (col - dx) % g.pn;
        if (pc < 0) pc += g.pn;
        return cmul(prow[pc], mrow[col]);
    }
};

// Loader: a complex pn x pn array used as it is (the coefficient array of the coarse-grid reconstruction).
struct FieldLoader {
    const float2* A;         // [pn,pn]
    const float2* row;
    __device__ __forceinline__ void begin_line(int, int a, const PassGeom& g) { row = A + (size_t)(g.ky0 + g.c + a) * g.pn; }
    __device__ __forceinline__ float2 load(int k, const PassGeom& g) const { return row[k + g.c]; }
};

// Loader: a real image (the bilinearly scaled mask, mask.py:76-81).  Line a / sample k of the
// padded N x N frame map to img[a + off][k - kx0 + off]; the zero padding (or, when the
// scaled mask is larger than N, the crop) is expressed by the window and `off` alone.
struct RealImageLoader {
    const float* img;        // [n,n]
    int n, off;
    const float* row;
    __device__ __forceinline__ void begin_line(int, int a, const PassGeom&) { row = img + (size_t)(a + off) * n + off; }
    __device__ __forceinline__ float2 load(int k, const PassGeom& g) const {
        return make_float2(row[k - g.kx0], 0.f);
    }
};

// ----------------------------------------------------------------------------------
// y-pass with |E|^2 accumulation in registers over the batch
// ----------------------------------------------------------------------------------
template <int LOG2N, int RL, bool PRUNED>
__global__ __launch_bounds__(Launch<LOG2N>::THREADS, Launch<LOG2N>::WAVES) void k_ypass_acc(
    const float2* __restrict__ Tbuf, float* __restrict__ slab, const float2* __restrict__ twtab,
    PassGeom g, int nb, int G, int gstride)
{
    // blockIdx.y = plane * G + sub: group `sub` of plane `plane` takes the points sub, sub + G, ... of the
    // batch (T items plane * nb + s) and owns slab plane * gstride + sub.
    using F = LineFFT<LOG2N, +1>;
    using LC = Launch<LOG2N>;
    constexpr unsigned IN = PRUNED ? natural_in_mask(RL) : 0xFFFFu;
    constexpr unsigned OUT = out_mask(RL);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    const int lt = threadIdx.x % F::T, lg = threadIdx.x / F::T;
    float2* lds = smem + (size_t)lg * LC::NBUF * F::LDS_LINE;

    typename F::Twiddles tw;
    F::load_twiddles(tw, twtab, lt, smem + LC::LDS_EXCH, threadIdx.x, LC::THREADS);

    const int tile = blockIdx.x * LC::L + lg;
    const bool active = tile < g.nt;
    const int plane = blockIdx.y / G, grp = blockIdx.y - plane * G;
    Tbuf += (size_t)plane * nb * g.t_point;
    slab += (size_t)plane * gstride * g.nt * 4 * g.pn;

    float acc[4][16];
    static_for<0, 4>([&](auto c_) {
        static_for<0, 16>([&](auto m_) {
            if constexpr ((OUT >> decltype(m_)::value) & 1u) acc[decltype(c_)::value][decltype(m_)::value] = 0.f;
        });
    });

    // per-thread input map: slot e <-> byte offset of T row a_e inside this tile (or out of range)
    unsigned voff[16];
    static_for<0, 16>([&](auto e_) {
        constexpr int e = decltype(e_)::value;
        if constexpr ((IN >> e) & 1u) {
            int k;
            const bool ok = centred_index(lt + F::T * e, F::N, g.ky0, g.ky1, k) && active;
            voff[e] = ok ? t_offset(g, (unsigned)(k - g.ky0), (unsigned)tile * 4u) * 8u : BUF_OOB;
        }
    });
    float2 nx[16];
    auto load_column = [&](int s, int cidx) {
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point, (size_t)g.t_point * sizeof(float2));
        static_for<0, 16>([&](auto e_) {
            constexpr int e = decltype(e_)::value;
#ifdef LITHO_DIAG_YNOLOAD
            if constexpr ((IN >> e) & 1u) nx[e] = make_float2(1.0f + 0.001f * e, 0.5f);
#else
            if constexpr ((IN >> e) & 1u) nx[e] = buf_load_c64(rT, voff[e] + cidx * 8u);
#endif
        });
    };

    int flip = 0;
    for (int s = grp; s < nb; s += G) {
        static_for<0, 4>([&](auto c_) {
            constexpr int cidx = decltype(c_)::value;
            load_column(s, cidx);
            float2 x[16];
            static_for<0, 16>([&](auto e_) {
                constexpr int e = decltype(e_)::value;
                if constexpr ((IN >> e) & 1u) x[e] = nx[e];
                else x[e] = make_float2(0.f, 0.f);
            });
            F::template run<LC::NBUF>(x, tw, lds, lt, flip);
            static_for<0, 16>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                if constexpr ((OUT >> m) & 1u) acc[cidx][m] = fmaf(x[m].x, x[m].x, fmaf(x[m].y, x[m].y, acc[cidx][m]));
            });
        });
    }

    if (!active) return;
    // flush into this group's private slab, laid out [G][qx][qy] (qy contiguous -> coalesced)
    static_for<0, 4>([&](auto c_) {
        constexpr int cidx = decltype(c_)::value;
        const int qx = tile * 4 + cidx;
        if (qx < g.pn) {
            float* srow = slab + ((size_t)grp * g.nt * 4 + qx) * g.pn;
            static_for<0, 16>([&](auto m_) {
                constexpr int m = decltype(m_)::value;
                if constexpr ((OUT >> m) & 1u) {
                    int u;
                    if (centred_index(lt + F::T * m, F::N, -g.c, g.pn - g.c, u)) srow[u + g.c] += acc[cidx][m];
                }
            });
        }
    });
}

// ----------------------------------------------------------------------------------
// x-pass, wave-per-line variant for N = 4096, pn = 2048, pupil inside the unit disk, no wrapping
// shift.  Work item = (group of 4 consecutive box rows, source point); the 4 waves of a workgroup
// take the 4 rows, i.e. exactly the rows that share each 128-byte line of T, so every line is
// completed inside one CU.  Each workgroup walks a contiguous range of items (row-group major), so
// the grid is sized to the machine and the pupil row is re-read only when the row group changes.
// MEASURED SLOWER than the radix-16 x-pass (10.9 vs 6.5 us/point at 2048^2: 32 scattered 8-byte stores per
// lane), so it is opt-in (LITHO_ABBE_W64X=1) and kept only as a parity-tested alternative.
// ----------------------------------------------------------------------------------
template <int LOG2N>
__global__ __launch_bounds__(256, 2) void k_xpass_w64(
    const float2* __restrict__ P, const float2* __restrict__ M, const int* __restrict__ shifts,
    float2* __restrict__ Tbuf, const float2* __restrict__ twtab, PassGeom g, int nb, int items_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = reinterpret_cast<float*>(smem_raw);
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably wave-uniform: descriptors built from it stay scalar
    float* lds = smem + wv * Wave4096::LDS_FLOATS;
    Wave4096::LaneTwiddles tw;
    Wave4096::load_lane_twiddles(tw, twtab, lane, 1);

    const int ngroups = (g.rows + 3) >> 2;
    const int items = ngroups * nb;
    const int i0 = blockIdx.x * items_per_wg;
    const int i1 = min(items, i0 + items_per_wg);

    // Live input slots j (sample n = lane + 64 j): k = lane + 64 j (j <= 8) or lane + 64 j - 4096 (j >= 56).
    // The buffer descriptors are WINDOWED on the valid columns [kx0, kx1) of the current row, so the
    // hardware range check is the validity test (a slot left of the window wraps to a huge unsigned
    // offset) and every slot address is one base register + a compile-time constant.
    const unsigned win_bytes = (unsigned)(g.kx1 - g.kx0) * 8u;
    const unsigned vb = (unsigned)(lane - g.kx0) * 8u;
    auto slot_off = [&](int j) { return vb + (unsigned)(j <= 8 ? 512 * j : 512 * j - 32768); };

    float2 pv[64];
    int cur_group = -1, a = 0, r = 0;
    bool active = false;
    for (int it = i0; it < i1; ++it) {
        const int grp = it / nb, s = it - grp * nb;
        if (grp != cur_group) {                                  // wave-uniform: new row group, reload the pupil row
            cur_group = grp;
            a = grp * 4 + wv;
            active = a < g.rows;
            r = g.ky0 + g.c + (active ? a : 0);
            const __amdgpu_buffer_rsrc_t rP = make_rsrc(P + (size_t)r * g.pn + g.c + g.kx0, active ? win_bytes : 0u);
            static_for<0, 64>([&](auto j_) {
                constexpr int j = decltype(j_)::value;
                if constexpr (j <= 8 || j >= 56) pv[j] = buf_load_c64(rP, slot_off(j));
            });
        }
        if (!active) continue;                                   // whole wave idle: no workgroup barriers anywhere
        const int dy = shifts[2 * s], dx = shifts[2 * s + 1];
        const __amdgpu_buffer_rsrc_t rM = make_rsrc(M + (size_t)(r + dy) * g.pn + dx + g.c + g.kx0, win_bytes);
        float2 x[64];
        static_for<0, 64>([&](auto j_) {
            constexpr int j = decltype(j_)::value;
            if constexpr (j <= 8 || j >= 56) x[j] = cmul(pv[j], buf_load_c64(rM, slot_off(j)));
            else x[j] = make_float2(0.f, 0.f);
        });
        Wave4096::run(x, tw, lds, lane);
        // kept bins u in [-1024, 1024): k2 in 0..15 -> q = lane + 64 k2 + 1024 ; k2 in 48..63 -> q = lane + 64 (k2 - 48)
        const __amdgpu_buffer_rsrc_t rT =
            make_rsrc(Tbuf + (size_t)s * g.t_point, (size_t)g.t_point * sizeof(float2));
        const unsigned vbase = (((unsigned)(lane >> 2) * g.rows + a) * 4u + (lane & 3u)) * 8u;
        const unsigned kstride = 16u * g.rows * 32u;             // bytes between q and q + 64
        static_for<0, 32>([&](auto i_) {
            constexpr int i = decltype(i_)::value;
            constexpr int k2 = i < 16 ? i : 32 + i;
            constexpr int kk = i < 16 ? i + 16 : i - 16;         // (q - lane) / 64
#ifdef LITHO_DIAG_XNOSTORE
            diag_keep(x[Wave4096::brev(k2)], vbase + kk * kstride);
#else
            buf_store_c64(rT, vbase + kk * kstride, x[Wave4096::brev(k2)]);
#endif
        });
    }
}

// y-pass that writes the complex field instead (calculateFFTAerial, mask spectrum).
template <int LOG2N, int SIGN>
__global__ __launch_bounds__(Launch<LOG2N>::THREADS, Launch<LOG2N>::WAVES) void k_ypass_field(
    const float2* __restrict__ Tbuf, float2* __restrict__ field, const float2* __restrict__ twtab, PassGeom g)
{
    using F = LineFFT<LOG2N, SIGN>;
    using LC = Launch<LOG2N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    const int lt = threadIdx.x % F::T, lg = threadIdx.x / F::T;
    float2* lds = smem + (size_t)lg * LC::NBUF * F::LDS_LINE;
    typename F::Twiddles tw;
    F::load_twiddles(tw, twtab, lt, smem + LC::LDS_EXCH, threadIdx.x, LC::THREADS);
    const int tile = blockIdx.x * LC::L + lg;
    const bool active = tile < g.nt;
    int flip = 0;
    for (int cidx = 0; cidx < 4; ++cidx) {
        float2 x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int k;
            const bool ok = centred_index(lt + F::T * e, F::N, g.ky0, g.ky1, k) && active;
            x[e] = ok ? Tbuf[t_offset(g, (unsigned)(k - g.ky0), (unsigned)tile * 4u + cidx)] : make_float2(0.f, 0.f);
        }
        F::template run<LC::NBUF>(x, tw, lds, lt, flip);
        const int qx = tile * 4 + cidx;
        if (!active || qx >= g.pn) continue;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            int u;
            if (centred_index(lt + F::T * m, F::N, -g.c, g.pn - g.c, u)) field[(size_t)(u + g.c) * g.pn + qx] = x[m];
        }
    }
}

// y-pass that ADDS scale * Re(field) to a real image (coarse-grid reconstruction: the interpolated intensity).
template <int LOG2N>
__global__ __launch_bounds__(Launch<LOG2N>::THREADS, Launch<LOG2N>::WAVES) void k_ypass_addreal(
    const float2* __restrict__ Tbuf, float* __restrict__ img, float scale, const float2* __restrict__ twtab, PassGeom g)
{
    using F = LineFFT<LOG2N, +1>;
    using LC = Launch<LOG2N>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float2* smem = reinterpret_cast<float2*>(smem_raw);
    const int lt = threadIdx.x % F::T, lg = threadIdx.x / F::T;
    float2* lds = smem + (size_t)lg * LC::NBUF * F::LDS_LINE;
    typename F::Twiddles tw;
    F::load_twiddles(tw, twtab, lt, smem + LC::LDS_EXCH, threadIdx.x, LC::THREADS);
    const int tile = blockIdx.x * LC::L + lg;
    const bool active = tile < g.nt;
    int flip = 0;
    for (int cidx = 0; cidx < 4; ++cidx) {
        float2 x[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            int k;
            const bool ok = centred_index(lt + F::T * e, F::N, g.ky0, g.ky1, k) && active;
            x[e] = ok ? Tbuf[t_offset(g, (unsigned)(k - g.ky0), (unsigned)tile * 4u + cidx)] : make_float2(0.f, 0.f);
        }
        F::template run<LC::NBUF>(x, tw, lds, lt, flip);
        const int qx = tile * 4 + cidx;
        if (!active || qx >= g.pn) continue;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            int u;
            if (centred_index(lt + F::T * m, F::N, -g.c, g.pn - g.c, u)) img[(size_t)(u + g.c) * g.pn + qx] += scale * x[m].x;
        }
    }
}

// ----------------------------------------------------------------------------------
// per-FFT-size launch table (one translation unit per size, see inst_*.hip)
// ----------------------------------------------------------------------------------
struct SizeOps {
    // variant: -1 = generic (any even pn, runtime predication); 0/1/2 = pruned, RL = log2(N/pn).
    // np = planes fused into the launch (1, 2 or 4; the generic variant takes 1 only): P points at the first
    // plane, T item (p, s) = p * nb + s.
    hipError_t (*xpass_abbe)(int variant, int np, const float2* P, const float2* M, const int* shifts, float2* T,
                             const float2* tw, const PassGeom& g, int nb, int chunk, hipStream_t st);
    hipError_t (*xpass_general)(const AbbeLoader& ld, float2* T, const float2* tw, const PassGeom& g, int nb,
                                hipStream_t st);
    // N = 2 pn as two N/2-point transforms per row (k_xpass_split; N = 8192 only): hipErrorNotSupported otherwise
    hipError_t (*xpass_split)(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                              const PassGeom& g, int nb, int chunk, hipStream_t st);
    // several adjacent box rows per wave, whole-line T stores (k_xpass_rect; N = 512..2048, 8-column tiles)
    hipError_t (*xpass_rect)(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                             const PassGeom& g, int nb, int chunk, hipStream_t st);
    hipError_t (*xpass_real_fwd)(const RealImageLoader& ld, float2* T, const float2* tw, const PassGeom& g,
                                 hipStream_t st);
    // y-pass over `planes` planes x G groups per plane (grid.y = planes * G); slab of (plane, group) =
    // plane * gstride + group
    hipError_t (*ypass_acc)(int variant, const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb,
                            int planes, int G, int gstride, hipStream_t st);
    hipError_t (*ypass_field)(int sign, const float2* T, float2* field, const float2* tw, const PassGeom& g,
                              hipStream_t st);
    // coarse-grid reconstruction: rows of a complex coefficient array -> T (sign +), then img += scale * Re(field)
    hipError_t (*xpass_field_inv)(const FieldLoader& ld, float2* T, const float2* tw, const PassGeom& g, hipStream_t st);
    hipError_t (*ypass_addreal)(const float2* T, float* img, float scale, const float2* tw, const PassGeom& g,
                                hipStream_t st);
    // wave-per-line passes (y: N = 1024..8192 with pn = N/2, pruned only; x: N = 4096); hipErrorNotSupported otherwise
    hipError_t (*xpass_w64)(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                            const PassGeom& g, int nb, hipStream_t st);
    hipError_t (*ypass_w64)(const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int planes,
                            int G, int gstride, hipStream_t st);
};
const SizeOps* size_ops(int log2n);      // nullptr outside 4..14

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is needed once per kernel and device, not once per launch
// (a config-3 image is ~23,000 launches).  One LdsOnce lives in each launcher; bit d = done on device d.
struct LdsOnce {
    unsigned done = 0;
};
template <typename K>
static hipError_t set_lds(LdsOnce& once, K kern, size_t bytes)
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned bit = 1u << (dev & 31);
    if (__atomic_load_n(&once.done, __ATOMIC_ACQUIRE) & bit) return hipSuccess;
    e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) __atomic_fetch_or(&once.done, bit, __ATOMIC_RELEASE);
    return e;
}

// defined (and explicitly instantiated for 8 <= LOG2N <= 13) in instw_*.hip
template <int LOG2N>
hipError_t launch_ypass_wave(const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int planes,
                             int G, int gstride, hipStream_t st);

template <int LOG2N>
hipError_t launch_xpass_rect(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                             const PassGeom& g, int nb, int chunk, hipStream_t st);

template <int LOG2N>
struct SizeImpl {
    using LC = Launch<LOG2N>;

    template <int RL, bool PRUNED, int NP, int RP = 1>
    static hipError_t xa(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                         const PassGeom& g, int nb, int chunk, hipStream_t st)
    {
        static LdsOnce once;
        auto kern = k_xpass_abbe<LOG2N, RL, PRUNED, NP, RP>;
        hipError_t e = set_lds(once, kern, LC::LDS_BYTES);
        if (e != hipSuccess) return e;
        // L == 1: grid.x padded to a multiple of 32 for the XCD-aware row mapping in the kernel
        const int wrows = (g.rows + RP - 1) / RP;              // rows (RP = 2: row pairs) to hand out
        dim3 grid(LC::L == 1 ? (wrows + 31) / 32 * 32 : (wrows + LC::L - 1) / LC::L, (nb + chunk - 1) / chunk);
        hipLaunchKernelGGL(kern, grid, dim3(LC::THREADS), LC::LDS_BYTES, st, P, M, shifts, T, tw, g, nb, chunk);
        note_kernel(0, PRUNED ? "k_xpass_abbe<%d, %d, true, %d, %d>" : "k_xpass_abbe<%d, %d, false, %d, %d>", LOG2N, RL, NP, RP);   // as rocprofv3 prints it
        return hipGetLastError();
    }
    template <int RL>
    static hipError_t xa_np(int np, const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                            const PassGeom& g, int nb, int chunk, hipStream_t st)
    {
        if constexpr (LOG2N == 12 && RL == 0) {
            if (np == 1 && g.row_pairs) return xa<RL, true, 1, 2>(P, M, shifts, T, tw, g, nb, chunk, st);
        }
        if (np == 1) return xa<RL, true, 1>(P, M, shifts, T, tw, g, nb, chunk, st);
        if constexpr (RL >= 1) {       // RL = 0 (N = pn) has 9 live input slots: 2 or 4 pupil rows would spill
            if (np == 2) return xa<RL, true, 2>(P, M, shifts, T, tw, g, nb, chunk, st);
            if (np == 4) return xa<RL, true, 4>(P, M, shifts, T, tw, g, nb, chunk, st);
        }
        return hipErrorInvalidValue;
    }
    static hipError_t xpass_abbe(int variant, int np, const float2* P, const float2* M, const int* shifts, float2* T,
                                 const float2* tw, const PassGeom& g, int nb, int chunk, hipStream_t st)
    {
        switch (variant) {
            case 0: return xa_np<0>(np, P, M, shifts, T, tw, g, nb, chunk, st);
            case 1: return xa_np<1>(np, P, M, shifts, T, tw, g, nb, chunk, st);
            case 2: return xa_np<2>(np, P, M, shifts, T, tw, g, nb, chunk, st);
            default:
                if (np != 1) return hipErrorInvalidValue;
                return xa<-1, false, 1>(P, M, shifts, T, tw, g, nb, chunk, st);
        }
    }
    static hipError_t xpass_general(const AbbeLoader& ld, float2* T, const float2* tw, const PassGeom& g, int nb,
                                    hipStream_t st)
    {
        static LdsOnce once;
        auto kern = k_xpass<LOG2N, +1, AbbeLoader>;
        hipError_t e = set_lds(once, kern, LC::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((g.rows + LC::L - 1) / LC::L, nb), dim3(LC::THREADS), LC::LDS_BYTES, st, ld, T,
                           tw, g);
        note_kernel(0, "k_xpass<%d, 1, litho::AbbeLoader>", LOG2N);
        return hipGetLastError();
    }
    static hipError_t xpass_split(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                                  const PassGeom& g, int nb, int chunk, hipStream_t st)
    {
        if constexpr (LOG2N == 13) {
            using LH = Launch<LOG2N - 1>;
            static LdsOnce once;
            auto kern = k_xpass_split<LOG2N>;
            hipError_t e = set_lds(once, kern, LH::LDS_BYTES);
            if (e != hipSuccess) return e;
            dim3 grid((g.rows + 31) / 32 * 32, (nb + chunk - 1) / chunk);
            hipLaunchKernelGGL(kern, grid, dim3(LH::THREADS), LH::LDS_BYTES, st, P, M, shifts, T, tw, g, nb, chunk);
            note_kernel(0, "k_xpass_split<%d>", LOG2N);
            return hipGetLastError();
        } else {
            return hipErrorNotSupported;
        }
    }
    static hipError_t xpass_rect(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                                 const PassGeom& g, int nb, int chunk, hipStream_t st)
    {
        if constexpr (LOG2N >= 9 && LOG2N <= 13) return launch_xpass_rect<LOG2N>(P, M, shifts, T, tw, g, nb, chunk, st);
        else return hipErrorNotSupported;
    }
    static hipError_t xpass_real_fwd(const RealImageLoader& ld, float2* T, const float2* tw, const PassGeom& g,
                                     hipStream_t st)
    {
        static LdsOnce once;
        auto kern = k_xpass<LOG2N, -1, RealImageLoader>;
        hipError_t e = set_lds(once, kern, LC::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((g.rows + LC::L - 1) / LC::L, 1), dim3(LC::THREADS), LC::LDS_BYTES, st, ld, T, tw,
                           g);
        return hipGetLastError();
    }
    template <int RL, bool PRUNED>
    static hipError_t ya(const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int planes, int G,
                         int gstride, hipStream_t st)
    {
        static LdsOnce once;
        auto kern = k_ypass_acc<LOG2N, RL, PRUNED>;
        hipError_t e = set_lds(once, kern, LC::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((g.nt + LC::L - 1) / LC::L, planes * G), dim3(LC::THREADS), LC::LDS_BYTES, st, T,
                           slab, tw, g, nb, G, gstride);
        note_kernel(1, PRUNED ? "k_ypass_acc<%d, %d, true>" : "k_ypass_acc<%d, %d, false>", LOG2N, RL);
        return hipGetLastError();
    }
    static hipError_t ypass_acc(int variant, const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb,
                                int planes, int G, int gstride, hipStream_t st)
    {
        switch (variant) {
            case 0: return ya<0, true>(T, slab, tw, g, nb, planes, G, gstride, st);
            case 1: return ya<1, true>(T, slab, tw, g, nb, planes, G, gstride, st);
            case 2: return ya<2, true>(T, slab, tw, g, nb, planes, G, gstride, st);
            default: return ya<-1, false>(T, slab, tw, g, nb, planes, G, gstride, st);
        }
    }
    static hipError_t xpass_field_inv(const FieldLoader& ld, float2* T, const float2* tw, const PassGeom& g, hipStream_t st)
    {
        static LdsOnce once;
        auto kern = k_xpass<LOG2N, +1, FieldLoader>;
        hipError_t e = set_lds(once, kern, LC::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((g.rows + LC::L - 1) / LC::L, 1), dim3(LC::THREADS), LC::LDS_BYTES, st, ld, T, tw, g);
        return hipGetLastError();
    }
    static hipError_t ypass_addreal(const float2* T, float* img, float scale, const float2* tw, const PassGeom& g,
                                    hipStream_t st)
    {
        static LdsOnce once;
        auto kern = k_ypass_addreal<LOG2N>;
        hipError_t e = set_lds(once, kern, LC::LDS_BYTES);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((g.nt + LC::L - 1) / LC::L), dim3(LC::THREADS), LC::LDS_BYTES, st, T, img, scale, tw, g);
        return hipGetLastError();
    }
    static hipError_t xpass_w64(const float2* P, const float2* M, const int* shifts, float2* T, const float2* tw,
                                const PassGeom& g, int nb, hipStream_t st)
    {
        if constexpr (LOG2N == 12) {
            static LdsOnce once;
            constexpr size_t lds = 4 * Wave4096::LDS_FLOATS * sizeof(float);
            auto kern = k_xpass_w64<LOG2N>;
            hipError_t e = set_lds(once, kern, lds);
            if (e != hipSuccess) return e;
            const int items = ((g.rows + 3) / 4) * nb;
            const int res = 2 * device_cus();                        // 2 resident workgroups per CU
            const int wgs = items < res ? items : res;
            const int per = (items + wgs - 1) / wgs;
            hipLaunchKernelGGL(kern, dim3((items + per - 1) / per), dim3(256), lds, st, P, M, shifts, T, tw, g, nb, per);
            note_kernel(0, "k_xpass_w64<%d>", LOG2N);
            return hipGetLastError();
        } else {
            return hipErrorNotSupported;
        }
    }
    // wave-per-line / pair-of-waves / multi-column y-pass (csrc/wave_kernels.hpp, compiled in its own translation
    // units instw_*.hip with their own scheduling strategy); the T tile width (4 or 8 columns) comes from g.tcl
    static hipError_t ypass_w64(const float2* T, float* slab, const float2* tw, const PassGeom& g, int nb, int planes,
                                int G, int gstride, hipStream_t st)
    {
        if constexpr (LOG2N >= 8 && LOG2N <= 13) return launch_ypass_wave<LOG2N>(T, slab, tw, g, nb, planes, G, gstride, st);
        else return hipErrorNotSupported;
    }
    static hipError_t ypass_field(int sign, const float2* T, float2* field, const float2* tw, const PassGeom& g,
                                  hipStream_t st)
    {
        dim3 grid((g.nt + LC::L - 1) / LC::L);
        if (sign > 0) {
            static LdsOnce once;
            auto kern = k_ypass_field<LOG2N, +1>;
            hipError_t e = set_lds(once, kern, LC::LDS_BYTES);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, dim3(LC::THREADS), LC::LDS_BYTES, st, T, field, tw, g);
        } else {
            static LdsOnce once;
            auto kern = k_ypass_field<LOG2N, -1>;
            hipError_t e = set_lds(once, kern, LC::LDS_BYTES);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, grid, dim3(LC::THREADS), LC::LDS_BYTES, st, T, field, tw, g);
        }
        return hipGetLastError();
    }
};

// A host accessor (not a namespace-scope constant: that would be emitted for the device too).
#define LITHO_DEFINE_SIZE_OPS(L2)                                                                    \
    const SizeOps* size_ops_##L2()                                                                   \
    {                                                                                                \
        static const SizeOps ops{&SizeImpl<L2>::xpass_abbe, &SizeImpl<L2>::xpass_general,            \
                                 &SizeImpl<L2>::xpass_split, &SizeImpl<L2>::xpass_rect,              \
                                 &SizeImpl<L2>::xpass_real_fwd, &SizeImpl<L2>::ypass_acc,            \
                                 &SizeImpl<L2>::ypass_field, &SizeImpl<L2>::xpass_field_inv,         \
                                 &SizeImpl<L2>::ypass_addreal, &SizeImpl<L2>::xpass_w64,             \
                                 &SizeImpl<L2>::ypass_w64};                                          \
        return &ops;                                                                                 \
    }

}  // namespace litho
