// abbe_plan.hpp -- the HOST-ONLY half of the Abbe engine: sizes, the workspace layout, the launch planner and the
// call-level decisions (embedded evaluation, splitting a partly wrapping source list), as plain C++ with NO HIP dependency.
//
// The reference's loop has nothing to plan (imageformation.py:62-67: roll, calculateFFTAerial, abs()**2, +=); everything
// here decides how that loop is batched onto the device and WHERE in the caller's workspace each intermediate lives.
// abbe_engine.hip consumes these functions for the real launches; plan_dry_run.cpp exports the very same decisions through
// litho_abbe_plan_dry_run without touching a device, so that a CPU test can sweep every admissible size and assert that every
// region a plan uses lies inside litho_abbe_workspace_bytes and that simultaneously live regions are disjoint
// (tests/test_planner_cpu.py; round-4 review, item 4).  This header compiles with plain g++ (the test does that too).
#pragma once
#include <climits>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <initializer_list>

#include "../../include/litho_abbe.h"

#ifdef __HIPCC__
#define LITHO_HD __host__ __device__
#else
#define LITHO_HD
#endif

namespace litho {

// ----------------------------------------------------------------------------------
// geometry shared by the pass kernels
// ----------------------------------------------------------------------------------
struct PassGeom {
    int pn, c, N;
    int nt;                 // 4-column groups (ceil(pn/4)): one y-pass workgroup line each
    int tcl;                // log2 of the T tile width in columns (2..4): T is [tile][row][1<<tcl]
    int kx0, kx1;           // x-pass: valid input window [kx0,kx1) in centred coordinates
    int ky0, ky1;           // y-pass: valid input window = rows of T; a = k - ky0
    int rows;               // number of T rows (= ky1 - ky0)
    int general;            // 1: roll stays on P, modular gather (wrapping shifts)
    int rect_off;           // 1: N = 2048 y-pass by the S = 32 wave kernel instead of k_ypass_rect (test knob)
    int gcombine;           // 1: k_ypass_rect puts the two groups of a column block into ONE workgroup and combines their
                            //    accumulators through LDS before the slab flush (half the flush traffic)
    int row_pairs;          // 1: an x-pass workgroup takes two adjacent rows (N = pn = 4096: T streams through HBM, see k_xpass_abbe)
    int coop_dma;           // 1: 16-column tiles at N = pn = 4096 are read by k_ypass_coop_dma (next line prefetched by LDS-DMA), 0: k_ypass_coop
    unsigned xmask, ymask;  // bit e set: slot e can be non-zero for SOME thread (x / y input)
    long long t_point;      // float2 elements of T per source point = ceil(pn/tc)*rows*tc
};

// Slot sets.  RL = log2(N/pn) for power-of-two pn (else -1).  PRUNED: the input window lies in
// the "natural" support k in [-pn/4, pn/4] (pupil inside the unit disk of the [-2,2) sigma grid).
LITHO_HD constexpr unsigned natural_in_mask(int RL)
{
    return RL == 0 ? 0xF01Fu : RL == 1 ? 0xC007u : RL == 2 ? 0x8003u : 0xFFFFu;
}
LITHO_HD constexpr unsigned out_mask(int RL)
{
    return RL == 1 ? 0xF00Fu : RL == 2 ? 0xC003u : 0xFFFFu;     // bins u in [-pn/2, pn/2)
}

// plan words: [0]=min row,[1]=max row,[2]=min col,[3]=max col of non-zero pupil samples
//             [4]=min dy,[5]=max dy,[6]=min dx,[7]=max dx, [8]=source-point count
//             [9],[10]=min/max ROW with a non-zero sample on the columns c +- pn/4 (the edges of the natural support box),
//             [11],[12]=min/max COLUMN with a non-zero sample on the rows c +- pn/4, [13]=1 if a corner of that box is set
static constexpr int PLAN_WORDS = 14;

static constexpr int COARSE_PLANES = 4;                      // = the largest plane chunk
static constexpr int EDGE_MAX = 128;                         // longest box-edge support the coarse path handles
static constexpr int GAM_CHUNKS = 1024;                      // workgroups (partial sums) of k_nyquist_edges
static constexpr size_t GAM_PARTIAL = (size_t)GAM_CHUNKS * 2 * 2 * EDGE_MAX;      // [chunk][edge][2 * EDGE_MAX]
static inline size_t gam_float2(int pn) { return GAM_PARTIAL + 2 * 2 * EDGE_MAX + 2 * (size_t)pn; }
static constexpr size_t SIZEOF_FLOAT2 = 8;

struct EdgeGeom {
    int pn, c, h;            // grid, centre, half-width of the natural box
    int lo[2], len[2];       // edge 0: columns c +- h, support rows lo[0] .. lo[0] + len[0]; edge 1: rows c +- h, support columns
};

// y-pass groups = private partial images (slabs).  Up to 8 for large images; small images can afford more
// (their y-pass grid would otherwise be a handful of workgroups): as many as fit in 128 MiB, at most LITHO_G_CAP = 128.
// (64 until round 5: at 256^2 the coarse-grid y-pass -- 64 columns per workgroup, two groups per 512-thread workgroup -- was then
// a grid of 4 x 32 = 128 workgroups on 256 CUs; with 128 groups every CU has one: y-pass 70 -> 50 us per 809-item launch,
// config 1 0.593 -> 0.523 ms per image in three alternating A/B pairs.)
#ifndef LITHO_G_CAP
#define LITHO_G_CAP 128
#endif
static inline int g_cap(int pn)
{
    const size_t one = (size_t)((pn + 3) / 4) * 4 * pn * sizeof(float);
    size_t n = ((size_t)128 << 20) / one;
    return n < 8 ? 8 : (n > LITHO_G_CAP ? LITHO_G_CAP : (int)n);
}
static constexpr int SLAB_FLUSH_BATCHES = 64;
static constexpr size_t T_BUDGET_MAX = (size_t)1 << 30;
static constexpr size_t T_BUDGET_BIG = (size_t)4 << 30;      // images whose T items cannot be batched inside the Infinity Cache (pn >= 4096)
static constexpr size_t T_BUDGET_MIN = (size_t)256 << 20;

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// one T item of the general mode (full pn x pn window, 4-column tiles): the largest item a call can need at this size
static inline size_t general_item_bytes(int pn) { return (size_t)((pn + 3) / 4) * (size_t)pn * 4 * SIZEOF_FLOAT2; }

static inline size_t t_budget(int pn)
{
    const size_t one_general = general_item_bytes(pn);
    size_t b = 64 * one_general;
    if (b < T_BUDGET_MIN) b = T_BUDGET_MIN;                  // small images: room for batches of several points per y-pass group
    // pn >= 4096: a T item is 67 MB and more, T streams through HBM whatever the batch, and longer batches amortise the
    // y-pass's accumulator flush and ramp (4096^2, us per source point: 15 items 45.3, 30 44.1, 45 43.5, 60 43.2) --
    // 4 GiB of a 288 GB device.
    const size_t cap = pn >= 4096 ? T_BUDGET_BIG : T_BUDGET_MAX;
    if (b > cap) b = cap;
    if (b < one_general + one_general / 2) b = one_general + one_general / 2;
    return b;
}

// Sizes at which the coarse-grid path exists (its regions of the workspace are empty elsewhere: 1.5 GiB at 8192^2)
static inline bool coarse_eligible(int pn, int N)
{
    return N == 2 * pn && (pn == 256 || pn == 512 || pn == 1024 || pn == 2048 || pn == 4096);
}
static inline size_t ic_bytes(int pn, int N) { return coarse_eligible(pn, N) ? (size_t)COARSE_PLANES * pn * pn * sizeof(float) : 0; }
static inline size_t chat_bytes(int pn, int N) { return coarse_eligible(pn, N) ? (size_t)pn * pn * SIZEOF_FLOAT2 : 0; }
static inline size_t gam_bytes(int pn, int N) { return coarse_eligible(pn, N) ? gam_float2(pn) * SIZEOF_FLOAT2 : 0; }

// Grid size the engine RUNS a pn x pn problem at.  The specialised kernels and the coarse grid
// exist for pn = N and pn = N / 2; any other even size (a 1000^2 or 3000^2 mask; 10 nm pixels, where N = 4 pn) would fall to
// the generic, runtime-predicated kernels -- 3.4-3.6x slower per source point than the NEXT LARGER power of two.  Such a
// problem is embedded instead: mask spectrum and pupil centred in a zero-padded N / 2 (pn < N / 2) or N grid, same shift
// list, centre pn x pn of the accumulated intensity added to `out` -- the identical sum term by term as long as no shift
// wraps the pupil around the caller's own grid (checked on the original size; a wrapping list runs the general path as is).
// Measured (scripts/embed_ab.py, us per source point, embedded / plain): 1000^2 at N 2048 2.48 / 8.38, 2000^2 at N 4096
// 9.2 / 33.6, 1500^2 at N 2048 7.4 / 16.2, 3000^2 at N 4096 35.2 / 58.8; N = 4 pn: 256^2 0.63 / 0.94, 2048^2 28.4 / 37.7;
// but 300^2 in a 512 grid 0.52 / 0.50 -- so: N / 2 from 256 up (the coarse grid applies), N from 1024 up.
static inline int embedded_size(int pn, int N)
{
    if (pn == N || 2 * pn == N || (pn & 1)) return pn;
    if (2 * pn < N) return N / 2 >= 256 ? N / 2 : pn;
    return N >= 1024 ? N : pn;
}
// scratch of an embedded evaluation behind the workspace of the padded size: mask spectrum, COARSE_PLANES pupils, as many images
static inline size_t embed_extra_bytes(int pe)
{
    const size_t e = (size_t)pe * pe;
    return align_up(e * SIZEOF_FLOAT2, 256) + align_up(COARSE_PLANES * e * SIZEOF_FLOAT2, 256) + align_up(COARSE_PLANES * e * sizeof(float), 256);
}

// ----------------------------------------------------------------------------------
// The workspace of one grid size, as byte offsets from its start (carve() in abbe_engine.hip adds the base pointer)
// ----------------------------------------------------------------------------------
struct Region {
    size_t off, bytes;
    size_t end() const { return off + bytes; }
};
struct WsLayout {
    Region plan;        // 64 ints: plan words [0, 14), the (0,0) shift of litho_abbe_field at [16, 18), the split's ten words at [32, 42)
    Region twtab;       // N float2
    Region twtab2;      // pn float2: table of the coarse-grid transforms
    Region slab;        // g_cap * nt*4 * pn floats
    Region ic;          // coarse-grid intensity of the planes in flight: COARSE_PLANES * pn * pn floats
    Region chat;        // its spectrum: pn * pn float2
    Region gam;         // Nyquist-line work area: partial sums, Gamma, profiles
    Region T;           // the first-pass intermediate: t_budget(pn) bytes
    size_t total;       // = workspace_bytes_at(pn, N)
};
static inline WsLayout ws_layout(int pn, int N)
{
    const size_t nt = (pn + 3) / 4;
    WsLayout l;
    size_t p = 0;
    l.plan = {p, 256}; p += 256;
    l.twtab = {p, (size_t)N * SIZEOF_FLOAT2}; p += align_up((size_t)N * SIZEOF_FLOAT2, 256);
    l.twtab2 = {p, (size_t)pn * SIZEOF_FLOAT2}; p += align_up((size_t)pn * SIZEOF_FLOAT2, 256);
    l.slab = {p, (size_t)g_cap(pn) * nt * 4 * pn * sizeof(float)}; p += align_up(l.slab.bytes, 256);
    l.ic = {p, ic_bytes(pn, N)}; p += align_up(l.ic.bytes, 256);
    l.chat = {p, chat_bytes(pn, N)}; p += align_up(l.chat.bytes, 256);
    l.gam = {p, gam_bytes(pn, N)}; p += align_up(l.gam.bytes, 256);
    l.T = {p, t_budget(pn)}; p += align_up(l.T.bytes, 256);
    l.total = p;
    return l;
}
static inline size_t workspace_bytes_at(int pn, int N) { return ws_layout(pn, N).total; }
// what litho_abbe_workspace_bytes reports: the engine's own regions at this size, or -- for a size that runs embedded -- the
// larger of that (the general path of a wrapping source list) and the padded size's regions + the embedding scratch
static inline size_t workspace_bytes(int pn, int N)
{
    const size_t own = workspace_bytes_at(pn, N);
    const int pe = embedded_size(pn, N);
    if (pe == pn) return own;
    const size_t emb = workspace_bytes_at(pe, N) + embed_extra_bytes(pe);
    return own > emb ? own : emb;
}

static inline int check_sizes(int pn, int N)
{
    if (pn < 2 || pn > 16384 || (pn & 1)) return LITHO_E_ARG;
    if (N < 16 || N > 16384 || (N & (N - 1))) return LITHO_E_ARG;
    if (N < pn) return LITHO_E_NSMALL;
    return LITHO_OK;
}

static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static inline int env_int(const char* name, int dflt)
{
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}

// Tuning / test knobs.  Resolved ONCE per C-ABI call, never per launch: a field of the caller's litho_abbe_options that is
// >= 0 wins, otherwise the LITHO_ABBE_* environment variable, otherwise the default.
struct Knobs {
    int force_generic, force_general, groups, batch, xchunk, tile, w64, w64_8192, w64x, plane_chunk, xsplit, rect, xrect, coarse, gcombine, rowpairs, poison, embed, split, coopdma;
    static int pick(const litho_abbe_options* o, size_t off, const char* name, int dflt)
    {
        if (o && off + sizeof(int32_t) <= (size_t)o->size) {
            const int32_t v = *(const int32_t*)((const unsigned char*)o + off);
            if (v >= 0) return v;
        }
        return env_int(name, dflt);
    }
    static Knobs read(const litho_abbe_options* o)
    {
#define LITHO_KNOB(field, name, dflt) k.field = pick(o, offsetof(litho_abbe_options, field), name, dflt)
        Knobs k;
        LITHO_KNOB(force_generic, "LITHO_ABBE_FORCE_GENERIC", 0);
        LITHO_KNOB(force_general, "LITHO_ABBE_FORCE_GENERAL", 0);
        LITHO_KNOB(groups, "LITHO_ABBE_GROUPS", 0);
        LITHO_KNOB(batch, "LITHO_ABBE_BATCH", 0);
        LITHO_KNOB(xchunk, "LITHO_ABBE_XCHUNK", 0);
        LITHO_KNOB(tile, "LITHO_ABBE_TILE", 0);          // 0 = automatic (8 columns on the wave-kernel path, else 4)
        LITHO_KNOB(w64, "LITHO_ABBE_W64", 1);
        LITHO_KNOB(w64_8192, "LITHO_ABBE_W64_8192", 1);
        LITHO_KNOB(w64x, "LITHO_ABBE_W64X", 0);
        LITHO_KNOB(plane_chunk, "LITHO_ABBE_PLANE_CHUNK", 0);
        LITHO_KNOB(xsplit, "LITHO_ABBE_XSPLIT", 1);
        LITHO_KNOB(rect, "LITHO_ABBE_RECT", 1);
        LITHO_KNOB(xrect, "LITHO_ABBE_XRECT", 1);
        LITHO_KNOB(coarse, "LITHO_ABBE_COARSE", 1);
        LITHO_KNOB(gcombine, "LITHO_ABBE_GCOMBINE", 1);
        LITHO_KNOB(rowpairs, "LITHO_ABBE_ROWPAIRS", 0);
        LITHO_KNOB(poison, "LITHO_ABBE_POISON", 0);
        LITHO_KNOB(embed, "LITHO_ABBE_EMBED", 1);
        LITHO_KNOB(split, "LITHO_ABBE_SPLIT", 1);
        LITHO_KNOB(coopdma, "LITHO_ABBE_COOPDMA", 1);
#undef LITHO_KNOB
        return k;
    }
};

// Which specialised kernel variant fits this geometry (-1 = generic).
static inline int pick_variant(const PassGeom& g, const Knobs& kn)
{
    if (g.general || (g.pn & (g.pn - 1))) return -1;
    const int rl = ilog2(g.N) - ilog2(g.pn);
    if (rl < 0 || rl > 2) return -1;
    const unsigned nat = natural_in_mask(rl);
    if ((g.xmask & ~nat) || (g.ymask & ~nat)) return -1;
    return kn.force_generic ? -1 : rl;
}

// T tile width (columns): 4 unless LITHO_ABBE_TILE says 8 or 16 (tuning knob)
static inline void set_tile(PassGeom& g, int rows, int tc = 4)
{
    g.tcl = (tc == 16) ? 4 : (tc == 8) ? 3 : 2;
    const long long ntile = (g.pn + (1 << g.tcl) - 1) >> g.tcl;
    g.t_point = (ntile * rows) << g.tcl;
}

// bit e of the mask: some thread t of a line has its slot e (sample n = t + T*e) inside [lo,hi)
static inline unsigned slot_mask(int N, int lo, int hi)
{
    const int T = N / 16;
    unsigned m = 0;
    for (int e = 0; e < 16; ++e)
        for (int t = 0; t < T; ++t) {
            const int n = t + T * e;
            const int k = (n >= hi) ? n - N : n;
            if (k >= lo && k < hi) { m |= 1u << e; break; }
        }
    return m;
}

static inline void make_geom(PassGeom& g, int pn, int N, int r0, int c0, int h, int wdt, int general, int tile_cols = 4)
{
    g.pn = pn; g.c = pn / 2; g.N = N; g.nt = (pn + 3) / 4;
    g.kx0 = c0 - g.c; g.kx1 = c0 + wdt - g.c;
    g.ky0 = r0 - g.c; g.ky1 = r0 + h - g.c;
    g.rows = h; g.general = general; g.rect_off = 0; g.gcombine = 0; g.row_pairs = 0; g.coop_dma = 0;
    g.xmask = slot_mask(N, g.kx0, g.kx1);
    g.ymask = slot_mask(N, g.ky0, g.ky1);
    set_tile(g, h, tile_cols);
}

// Everything abbe_accumulate decides before its launch loop: which kernels run and how the work is batched.
struct AbbePlan {
    PassGeom g;
    int general, variant;       // 1: roll kept on P (wrapping shifts); kernel specialisation (-1 generic, else log2(N/pn))
    bool natural_box;           // the pupil's support box lies inside |k| <= pn/4 (and no shift wraps)
    int r0, c0, h, wdt;         // pupil support box (rows r0 .. r0+h, columns c0 .. c0+wdt)
    bool wave_y;                // y-pass by the wave-level family (k_ypass_wave / k_ypass_pair / k_ypass_rect)
    bool split_x, rect_x, fused_x;   // x-pass: k_xpass_split / k_xpass_rect / plane-fused k_xpass_abbe (else per-plane fall-backs)
    int PC, G, xchunk;          // planes in flight per launch pair, y-pass groups per plane, source points per x-pass workgroup
    int slabs;                  // slabs per plane the y-pass actually writes: G, or G / 2 when k_ypass_rect<.., 2> folds group pairs
    int64_t bs;                 // source points per batch
};

// pl = the plan words read back from the device (pupil box, shift extents, count); t_bytes = bytes of the T region this run
// may use (the layout's, or less when the lists of a split source list sit at its end); cus = compute units of the device.
static inline int plan_abbe(AbbePlan& pp, size_t t_bytes, const Knobs& kn, const int pl[PLAN_WORDS], int pn, int N, int planes, int cus)
{
    int r0 = pl[0], h = pl[1] - pl[0] + 1, c0 = pl[2], wdt = pl[3] - pl[2] + 1;
    const bool nowrap = (r0 + pl[4] >= 0) && (r0 + h - 1 + pl[5] <= pn - 1) &&
                        (c0 + pl[6] >= 0) && (c0 + wdt - 1 + pl[7] <= pn - 1);
    const int general = (!nowrap || kn.force_general) ? 1 : 0;
    if (general) { r0 = 0; c0 = 0; h = pn; wdt = pn; }
    PassGeom& g = pp.g;
    make_geom(g, pn, N, r0, c0, h, wdt, general, kn.tile > 0 ? kn.tile : 4);
    const int variant = pick_variant(g, kn);
    // The wave-level kernels, the split x-pass and the coarse-grid path hard-wire the NATURAL support |k| <= pn/4 (the
    // unit disk of the [-2,2) sigma grid): they load only the slots that cover it and the coarse grid assumes
    // |kappa| <= pn/2.  pick_variant's 16-slot masks are coarser than that (a one-sided box reaching k = 3 pn/8 - 1
    // still has the natural slot set), so the box itself is checked; anything wider runs the radix-16 kernels, whose
    // windows are runtime-predicated inside the admitted slots.
    const int cc = pn / 2, hh = pn / 4;
    const bool natural_box = !general && r0 >= cc - hh && r0 + h - 1 <= cc + hh && c0 >= cc - hh && c0 + wdt - 1 <= cc + hh;

    // wave-level y-pass kernels, N = 2 pn: N = 512, 1024, 2048 k_ypass_rect (8, 4, 2 columns per wave; fall-back
    // k_ypass_wave with S = 32 for 1024 and 2048), N = 4096 k_ypass_wave (S = 64), N = 8192 k_ypass_pair
    const int l2n = ilog2(N);
    const int lines_per_wg = (N / 16 >= 64) ? 1 : 64 / (N / 16);
    const bool rect_ok = kn.rect && (N == 2048 || N == 1024 || (N == 512 && (kn.tile <= 0 || kn.tile == 8)));
    const bool w64_ok = (pn * 2 == N) && ((N == 512 && rect_ok) || N == 1024 || N == 2048 || N == 4096 ||
                                          (N == 8192 && kn.w64_8192));   // w64_8192 defaults to 1
    // N = pn (the coarse-grid transform, and pixel sizes that give N = pn): full-output variants of the same kernels
    const bool full_ok = (pn == N) && (N == 1024 || N == 2048 || N == 4096 ||
                                       ((N == 512 || N == 256) && (kn.tile <= 0 || kn.tile == 8)));
    const bool w64_shape = ((w64_ok && variant == 1) || (full_ok && variant == 0)) && kn.w64 && natural_box;
    // T tile width.  The x-pass's T stores are bound by the memory system's rate for partial-line writes: measured
    // (scripts/ubench/write_bw.hip) 2.2 TB/s for 32-byte granules (4-column tiles), 3.4 TB/s for 64-byte granules
    // (8 columns), 5.2 TB/s for whole 128-byte lines.  The wave kernels read 8-column tiles at no extra cost, the
    // radix-16 y-pass does not (measured in round 1), so: 8 columns on the wave path, 4 elsewhere.
    // N = pn = 4096 (config 4's coarse grid): a T item is 67 MB, T streams through HBM, and there whole-line stores are
    // worth 7.5 us of the x-pass's 21.6 per item -- 16-column tiles, read by k_ypass_coop_dma (round 5: 21 us per item; round
    // 3's k_ypass_coop 24.4, k_ypass_wave 21.2 on 8-column tiles: 35.5 us per source point against 38.7 / 43.1).
    const bool coop16 = pn == N && N == 4096 && variant == 0;
    if (kn.tile <= 0 && w64_shape && !kn.w64x) set_tile(g, h, coop16 ? 16 : 8);
    const int tc = 1 << g.tcl;
    // k_ypass_rect: 4096 / N adjacent columns per wave (they must fit one T tile)
    const bool rect = (variant == 0 ? N <= 2048 : rect_ok && N <= 2048) && ((4096 / N) <= tc || (N == 256 && tc == 8));
    g.rect_off = rect ? 0 : 1;
    g.gcombine = kn.gcombine ? 1 : 0;
    g.row_pairs = (kn.rowpairs && g.tcl == 3) ? 1 : 0;
    g.coop_dma = kn.coopdma ? 1 : 0;
    const bool wave_y = w64_shape && (g.tcl == 2 || g.tcl == 3 || (g.tcl == 4 && N == 4096 && variant == 0)) && ((N != 512 && N != 256) || rect) &&
                        (variant == 1 || rect || N == 4096);

    // y-pass groups: the grid is (column blocks) x (planes in flight) x G workgroups; pick the smallest group
    // count that makes it a whole number of full-occupancy rounds (256 CUs x workgroups per CU).
    const int wave_cols = rect ? 4 * (4096 / N) : N == 1024 ? 8 : (N == 8192 ? 2 : 4);   // columns per wave-kernel workgroup
    const int wave_wpt = tc > wave_cols ? tc / wave_cols : 1;         // workgroups that share one T tile
    const int tile_blocks = !wave_y ? (g.nt + lines_per_wg - 1) / lines_per_wg
                            : wave_wpt == 1 ? (pn + wave_cols - 1) / wave_cols
                                            : wave_wpt * (((pn + tc - 1) / tc + 7) / 8 * 8);
    const int resident = cus * (wave_y ? ((N <= 2048 && !rect) ? 4 : 2) : l2n <= 12 ? 3 : (l2n == 13 ? 2 : 1));
    int a_ = tile_blocks, b_ = resident;
    while (b_) { const int t_ = a_ % b_; a_ = b_; b_ = t_; }
    int Gtot = resident / a_;                                  // groups that fill whole rounds
    if (kn.groups > 0) Gtot = kn.groups;
    if (Gtot < 1) Gtot = 1;
    if (Gtot > g_cap(pn)) Gtot = g_cap(pn);

    // Through-focus stacks: PC planes are in flight per launch pair.  The fused x-pass gathers the mask-spectrum
    // window of a source point once for all of them (NP = 4 / 2 / 1 planes per workgroup); the y-pass gives every
    // plane its own groups and slabs.  The T buffer of a launch pair holds PC x batch items for the planes in
    // flight.  The gather was never the x-pass's limit (its T stores are): PC = 2 takes 40 % of the x-pass's load
    // instructions away at equal x-pass time, but its 2 x batch items of T leave the Infinity Cache and the y-pass
    // pays 4-5 % for that (2-3 % of the total against plane-by-plane); on the coarse-grid path, whose y-pass is twice
    // as fast, it pays 26 % (2048^2 x 8 planes, us per point and plane: PC = 1 9.42 / 9.53, PC = 2 10.82 / 10.83,
    // PC = 4 with a quarter of the batch 12.7).  Default: plane by plane; LITHO_ABBE_PLANE_CHUNK = 2 / 4 selects the
    // fused launches (parity-tested).
    int PC = planes < 1 ? planes : 1;
    if (kn.plane_chunk > 0) PC = kn.plane_chunk < planes ? kn.plane_chunk : planes;
    if (PC > g_cap(pn)) PC = g_cap(pn);

    // Batch = source points per launch pair.  The intermediate T of one batch (PC planes x points) should stay
    // INSIDE the 256 MiB Infinity Cache between the two passes, and a y-pass workgroup wants several points per plane
    // to amortise its accumulator flush.  Measured (us per source point, round 2): 1024^2 (4.2 MB per item) 32 items
    // 3.01, 48 2.85, 56 2.84, 68 3.09; 2048^2 (16.8 MB) 8 items 14.65, 12 and 17 equal within the +-2.5 % scatter of
    // single samples (profiles/r02_tuning_sweeps.txt) -> budget 208 MiB.
    const size_t item_bytes = (size_t)g.t_point * SIZEOF_FLOAT2;
    const int64_t items_ws = (int64_t)(t_bytes / item_bytes);
    if (items_ws < 1) return LITHO_E_WORKSPACE;
    int64_t items = items_ws;
    int64_t items_cache = (int64_t)(((size_t)208 << 20) / item_bytes);
    // 4096^2 (67 MB per item): not even 8 items fit the cache, T round-trips HBM whatever the batch -- then the batch
    // is as long as the workspace allows (60 items of its 4 GiB: fewer accumulator flushes in the y-pass) and an x-pass
    // workgroup walks 30 items for its row (15 until round 5).  us per source point, coarse-grid path, alternating A/B (round 2, 1 GiB):
    // 8 items x chunks of 4 49.9 / 49.8; 12 x 6 47.9; 12 x 12 45.6; 15 x 5 47.5; 15 x 15 45.3 / 45.2; round 3 (4 GiB):
    // 30 x 15 44.1, 45 x 15 43.5, 60 x 15 43.2, 60 x 60 43.3.
    const bool beyond_cache = items_cache < 8;
    if (beyond_cache) items_cache = 60;
    if (items > items_cache) items = items_cache;
    if (PC > items_ws) PC = (int)items_ws;
    int G = Gtot / PC;                                         // groups per plane
    if (G < 1) G = 1;
    // Stacks keep the per-plane batch: T grows to PC x batch items and leaves the Infinity Cache, which costs the
    // y-pass less than flushing its accumulators twice as often (alternating A/B at 2048^2 x 8 planes, us per point
    // and plane, two boxes: PC = 1 13.40 / 13.76; PC = 2 with the batch halved 14.18, with the full batch 13.62 /
    // 14.05; PC = 4 13.68).
    int64_t bs = items;
    if (kn.batch > 0) bs = kn.batch;
    if (bs > items_ws / PC) bs = items_ws / PC;
    if (bs < 1) bs = 1;
    if (bs > 65535) bs = 65535;
    // Balance: every y-pass group gets the same number of points (batch multiple of G) and the x-pass
    // chunks divide the batch evenly (chunk = divisor of the batch nearest 4).
    const int64_t bs_cap = bs;
    if (kn.batch <= 0 && bs > G) bs -= bs % G;
    // Few, long batches (small images: config 1 is 3233 points in batches of up to 825): even batches instead of full ones
    // plus a short tail -- every launch pair costs 15-20 us before its first item (3233 = 4 x 768 + 161 was five launch
    // pairs, 4 x 809 is four).  A ragged split over the G groups (809 = 64 x 12 + 41) costs less than that.
    const int64_t S_plan = pl[8];
    if (kn.batch <= 0 && S_plan > bs && S_plan <= 64 * bs_cap) {
        const int64_t B = (S_plan + bs_cap - 1) / bs_cap;      // launch pairs needed at the cap
        int64_t even = (S_plan + B - 1) / B;
        if (even % G && even + (G - even % G) <= bs_cap) even += G - even % G;
        if (even >= 1 && (S_plan + even - 1) / even < (S_plan + bs - 1) / bs) bs = even;
    }
    int xchunk = kn.xchunk;                                    // source points per x-pass workgroup
    if (xchunk <= 0) {
        // ~4 source points per workgroup: the pupil rows (5 loads per plane) are amortised over the chunk, the
        // mask-spectrum window of each point over the planes (measured flat between 3 and 6 points)
        const int want = PC >= 4 ? 2 : 4;
        xchunk = want;
        if (want == 4) for (int cand : {4, 5, 3, 6, 2}) if (bs % cand == 0) { xchunk = cand; break; }
        if (want == 2) for (int cand : {2, 3, 1}) if (bs % cand == 0) { xchunk = cand; break; }
        if (beyond_cache && PC == 1) {                         // see above
            xchunk = (int)bs;
            // (round 5, beside k_ypass_coop_dma, bench.py --workload cfg4 --points 5400, alternating: 15 items per workgroup 195.4 /
            // 195.8 ms, 20 194.5 / 194.4, 30 194.0 / 194.1, 60 195.2 / 195.6 -- x-pass 13.95 -> 13.64 us per item at 30)
            for (int cand : {30, 15, 20, 12, 16, 10, 8, 6}) if (bs > cand && bs % cand == 0) { xchunk = cand; break; }
        }
        // 1024-point rows (64-thread workgroups, 16 per CU): longer chunks pay -- coarse-grid x-pass at 1024^2,
        // us per point: chunk 2 1.40, 3 1.27, 4 1.18, 6 1.12, 8 1.20, 12 1.03, 16 1.06, 24 1.34, 48 2.0
        if (N == 1024 && variant == 0 && PC == 1) for (int cand : {12, 16, 8, 6}) if (bs % cand == 0) { xchunk = cand; break; }
    }

    // N = 8192 = 2 pn: each row as two 4096-point transforms (k_xpass_split) instead of the 8192-point engine
    pp.split_x = !general && variant == 1 && N == 8192 && g.tcl >= 2 && kn.xsplit && natural_box;
    // Several box rows per wave on the wave-level engine, whole-line T stores (k_xpass_rect).  Measured (us per source
    // point, radix-16 x-pass -> k_xpass_rect): N = 1024 0.57 -> 0.36, N = 2048 1.43 -> 1.46, N = 512 0.27 -> 0.26: its
    // loads are not prefetched (no registers left), so it only pays where the radix-16 engine is at its weakest.
    // LITHO_ABBE_XRECT: 0 off, 1 N = 1024 only (default), 2 every N <= 2048 (parity tests).
    // The same kernel with every bin kept serves the coarse-grid transforms (variant 0, N = pn) -- only on request:
    // with twice the loads and stores per wave it is SLOWER than the radix-16 x-pass at every size (coarse-grid x-pass,
    // us per point, radix-16 -> rect: N' = 512 0.28 -> 0.32, 1024 1.25 -> 1.49, 2048 4.73 -> 6.77).
    pp.rect_x = natural_box && ((variant == 1 && pn * 2 == N) || (variant == 0 && pn == N)) && N >= 512 && N <= 2048 &&
                g.tcl == 3 && (kn.xrect >= 2 || (kn.xrect == 1 && variant == 1 && N == 1024));
    const bool wave_x_optin = wave_y && variant == 1 && N == 4096 && kn.w64x && g.tcl == 2;      // k_xpass_w64 (slower, parity-tested)
    pp.fused_x = !pp.split_x && !pp.rect_x && !general && variant >= 0 && !wave_x_optin;
    pp.general = general; pp.variant = variant; pp.r0 = r0; pp.c0 = c0; pp.h = h; pp.wdt = wdt;
    pp.natural_box = natural_box;
    pp.wave_y = wave_y; pp.PC = PC; pp.G = G; pp.xchunk = xchunk; pp.bs = bs;
    pp.slabs = (wave_y && rect && g.gcombine && G % 2 == 0) ? G / 2 : G;
    return LITHO_OK;
}

// ----------------------------------------------------------------------------------
// One run of the source-point loop (accumulate_planned in abbe_engine.hip): the direct plan at N, and -- when the coarse
// grid is admitted -- the plan of the pn-point transforms it runs instead, with the Nyquist-line edge geometry.
// ----------------------------------------------------------------------------------
struct RunPlan {
    AbbePlan direct;            // N-point transforms (always planned)
    AbbePlan coarse_plan;       // pn-point transforms (valid when coarse)
    bool coarse;
    EdgeGeom eg;
    const AbbePlan& run() const { return coarse ? coarse_plan : direct; }
};
// T bytes the once-per-plane reconstruction of the coarse-grid path needs (both of its transform pairs write one pn x pn item
// of 4-column tiles into the head of T, after the source-point loop is done with it)
static inline size_t reconstruct_t_bytes(int pn) { return general_item_bytes(pn); }
// `have_ops_c`: whether kernels of size pn exist (size_ops(ilog2(pn)) != nullptr: 16 <= pn <= 16384, always true where the
// coarse grid is eligible).  S = source points of this run.
static inline int plan_run(RunPlan& rp, size_t t_bytes, const Knobs& kn, const int pl[PLAN_WORDS], int pn, int N, int planes,
                           int64_t S, int cus, bool have_ops_c = true)
{
    const int rc = plan_abbe(rp.direct, t_bytes, kn, pl, pn, N, planes, cus);
    if (rc) return rc;
    // Coarse-grid path (N = 2 pn, pupil inside the natural box, no wrapping shift): the source-point loop runs
    // pn-point transforms on the grid q = 2 v (half the arithmetic per transformed line), the fine image is
    // reconstructed once per plane.  Needs the full-output wave kernels of size pn, empty box corners and box-edge
    // supports of at most EDGE_MAX samples; anything else takes the direct path.
    // The reconstruction is a fixed cost per call and plane (about ten small launches: 0.08 ms at 256^2 .. 0.5 ms at 4096^2
    // since round 4, when k_nyquist_reduce stopped taking 0.27 ms by itself), so short source lists stay on the direct path.
    // Break-even measured (scripts/coarse_breakeven.py, whole-call time, profiles/r04_coarse_breakeven.txt): S = 2,900 (256^2),
    // 400 (512^2), 180 (1024^2), 64 (2048^2), 48 (4096^2); thresholds a notch above.  LITHO_ABBE_COARSE = 2 ignores S.
    const int64_t s_min = pn == 256 ? 3072 : pn == 512 ? 512 : pn == 1024 ? 256 : pn == 2048 ? 96 : 64;
    bool coarse = kn.coarse && (kn.coarse >= 2 || S >= s_min) && coarse_eligible(pn, N) && rp.direct.variant == 1 && rp.direct.natural_box &&
                  pl[13] == 0;
    EdgeGeom& eg = rp.eg;
    eg.pn = pn; eg.c = pn / 2; eg.h = pn / 4; eg.lo[0] = eg.lo[1] = 0; eg.len[0] = eg.len[1] = 0;
    if (coarse) {
        eg.lo[0] = pl[10] >= pl[9] ? pl[9] : 0;   eg.len[0] = pl[10] >= pl[9] ? pl[10] - pl[9] + 1 : 0;
        eg.lo[1] = pl[12] >= pl[11] ? pl[11] : 0; eg.len[1] = pl[12] >= pl[11] ? pl[12] - pl[11] + 1 : 0;
        coarse = have_ops_c && eg.len[0] <= EDGE_MAX && eg.len[1] <= EDGE_MAX &&
                 plan_abbe(rp.coarse_plan, t_bytes, kn, pl, pn, pn, planes, cus) == LITHO_OK && rp.coarse_plan.variant == 0 && rp.coarse_plan.wave_y &&
                 rp.coarse_plan.PC <= COARSE_PLANES &&
                 // the reconstruction writes one general-mode item (2x a pruned coarse item) into the head of T: a T region cut
                 // short by the two lists of a split source list (only SPLIT_T_ROOM is guaranteed) must still hold it, or the
                 // reconstruction would run into the wrapping part's list (round-5 advice: only the dry-run TEST compared them)
                 t_bytes >= reconstruct_t_bytes(pn);
    }
    rp.coarse = coarse;
    return LITHO_OK;
}

// ----------------------------------------------------------------------------------
// Call level (abbe_accumulate): the grid the problem runs at, and where the two lists of a split source list go.
// ----------------------------------------------------------------------------------
static constexpr int64_t SPLIT_MIN_POINTS = 256;          // below that the two extra launches and the read-back cost more than they save
static constexpr size_t SPLIT_T_ROOM = (size_t)64 << 20;  // a split leaves at least this much of either T region
static constexpr int SPLIT_PER_BLOCK = 1024;

// the grid this problem runs at: its own, or the padded one of an embedded evaluation (embedded_size) when the workspace
// has room for it (litho_abbe_workspace_bytes says so; an older, smaller workspace simply runs the problem as it is)
static inline int run_size(int pn, int N, const Knobs& kn, size_t ws_bytes)
{
    int pe = kn.embed ? embedded_size(pn, N) : pn;
    if (pe != pn && ws_bytes < workspace_bytes_at(pe, N) + embed_extra_bytes(pe)) pe = pn;
    return pe;
}
static inline bool list_nowrap(const int pl[PLAN_WORDS], int pn)
{
    return pl[0] + pl[4] >= 0 && pl[1] + pl[5] <= pn - 1 && pl[2] + pl[6] >= 0 && pl[3] + pl[7] <= pn - 1;
}
// scratch of the embedded evaluation behind the padded size's regions
struct EmbedLayout { Region M2, P2, O2; };
static inline EmbedLayout embed_layout(int pe, int N)
{
    const size_t e2 = (size_t)pe * pe, base = workspace_bytes_at(pe, N);
    EmbedLayout e;
    e.M2 = {base, e2 * SIZEOF_FLOAT2};
    e.P2 = {base + align_up(e2 * SIZEOF_FLOAT2, 256), COARSE_PLANES * e2 * SIZEOF_FLOAT2};
    e.O2 = {e.P2.off + align_up(COARSE_PLANES * e2 * SIZEOF_FLOAT2, 256), COARSE_PLANES * e2 * sizeof(float)};
    return e;
}
// Splitting a source list whose shifts wrap the pupil around the grid for SOME of its points: the two compacted lists live at
// the end of the T region of the grid the non-wrapping part runs at (the padded grid's for an embedded size), and BOTH carves'
// T regions are cut short of them.
struct SplitLayout {
    bool ok;                    // the split is possible (the lists fit and leave SPLIT_T_ROOM of T on both carves)
    size_t list_bytes;          // bytes of one list
    Region list_a, list_b;      // non-wrapping / wrapping points
    size_t t_bytes_own;         // T bytes left to a run at the call's own size
    size_t t_end_pad;           // byte offset at which the T region of the padded carve must end (0 = not embedded)
    Region counts;              // block counts: the head of the own-size T region, dead before the loops start
};
static inline bool split_wanted(const Knobs& kn, bool nowrap, bool from_record, int64_t S)
{
    return !nowrap && !kn.force_general && kn.split && !from_record && (S >= SPLIT_MIN_POINTS || kn.split >= 2);
}
static inline SplitLayout split_layout(int pn, int pe, int N, int64_t S)
{
    SplitLayout s;
    const WsLayout own = ws_layout(pn, N);
    s.list_bytes = align_up((size_t)S * 2 * sizeof(int), 256);
    const size_t t_end = pe != pn ? workspace_bytes_at(pe, N) : own.total;
    const size_t t0_own = own.T.off;
    const size_t t0_pad = pe != pn ? ws_layout(pe, N).T.off : t0_own;
    s.ok = 2 * s.list_bytes < t_end;
    const size_t list_start = s.ok ? t_end - 2 * s.list_bytes : 0;
    s.ok = s.ok && list_start > t0_own + SPLIT_T_ROOM && list_start > t0_pad + SPLIT_T_ROOM;
    s.list_a = {list_start, s.list_bytes};
    s.list_b = {list_start + s.list_bytes, s.list_bytes};
    s.t_bytes_own = own.T.bytes;
    if (s.ok && t0_own + s.t_bytes_own > list_start) s.t_bytes_own = list_start - t0_own;
    s.t_end_pad = pe != pn ? list_start : 0;
    s.counts = {t0_own, (size_t)((S + SPLIT_PER_BLOCK - 1) / SPLIT_PER_BLOCK) * sizeof(int)};
    return s;
}
// plan words of an embedded run: the pupil's support box and its samples on the PADDED grid's natural-box edges move by the offset
static inline void embed_plan_words(const int pl[PLAN_WORDS], int pn, int pe, int pl2[PLAN_WORDS])
{
    const int off = (pe - pn) / 2;
    for (int i = 0; i < PLAN_WORDS; ++i) pl2[i] = pl[i];
    for (int i = 0; i < 4; ++i) pl2[i] += off;
    if (pl[10] >= pl[9]) { pl2[9] += off; pl2[10] += off; }
    if (pl[12] >= pl[11]) { pl2[11] += off; pl2[12] += off; }
}

// ----------------------------------------------------------------------------------
// The caller-held plan record (litho_abbe_plan.words, "the library's business"): everything a later call with the same pupil
// and source list needs in order to plan WITHOUT a read-back -- the 14 plan words, the grid the run was planned for, and (round
// 5) the outcome of the source-list split, so that a planned call with a partly wrapping (shifted) source keeps the fast path
// for its non-wrapping points: it re-runs the three small split kernels (deterministic, no read-back) and plans the two parts
// from the recorded counts and extents.  Sixteen words hold all that only packed: coordinates and shifts are < 2^15 in
// magnitude (pn <= 16384; shift extents beyond +-16384 are clamped there: they wrap on every grid), the "none seen" sentinels
// INT_MAX / INT_MIN travel as 32767 / -32768.
//   [0] box rows lo | hi << 16   [1] box columns   [2] dy min | max   [3] dx min | max   [4] source-point count
//   [5] edge rows ([9],[10])     [6] edge columns ([11],[12])        [7] corner flag | run size << 8
//   [8] RECORD_TAG | split flag  [9] non-wrapping count              [10],[11] its dy, dx extents   [12],[13] the wrapping part's
// ----------------------------------------------------------------------------------
static constexpr int32_t RECORD_TAG = 0x4C500500;            // "LP", format 5
// The C ABI takes any int32 shift (the general path reduces it modulo pn); extents beyond +-16384 = the largest pn already mean
// "wraps" on every grid, so they are CLAMPED there -- the no-wrap decision of a planned call is then the planning call's, and no
// real value can collide with a sentinel (round-5 advice: a truncated extent could judge no-wrap differently on reuse, and an
// extent of exactly 32767 / -32768 came back as INT_MAX / INT_MIN, whose sum in list_nowrap overflows).
static constexpr int RECORD_EXTENT_CLAMP = 16384;
static inline int32_t pack16(int lo, int hi)
{
    auto c = [](int v) {
        return v == INT_MAX ? 32767 : v == INT_MIN ? -32768 : v > RECORD_EXTENT_CLAMP ? RECORD_EXTENT_CLAMP : v < -RECORD_EXTENT_CLAMP ? -RECORD_EXTENT_CLAMP : v;
    };
    return (int32_t)(((uint32_t)(uint16_t)(int16_t)c(lo)) | ((uint32_t)(uint16_t)(int16_t)c(hi) << 16));
}
static inline void unpack16(int32_t w, int& lo, int& hi)
{
    auto u = [](int v) { return v == 32767 ? INT_MAX : v == -32768 ? INT_MIN : v; };
    lo = u((int)(int16_t)(uint16_t)((uint32_t)w & 0xFFFFu));
    hi = u((int)(int16_t)(uint16_t)((uint32_t)w >> 16));
}
// sw = the split's ten words as k_split_scan / k_split_write leave them: [0] non-wrapping count, [1] wrapping count,
// [2..5] dy min, dy max, dx min, dx max of the non-wrapping part, [6..9] of the wrapping part; nullptr = the list was not split
static inline void record_store(int32_t words[16], const int pl[PLAN_WORDS], int pe, const int* sw)
{
    for (int i = 0; i < 16; ++i) words[i] = 0;
    words[0] = pack16(pl[0], pl[1]); words[1] = pack16(pl[2], pl[3]);
    words[2] = pack16(pl[4], pl[5]); words[3] = pack16(pl[6], pl[7]);
    words[4] = pl[8];
    words[5] = pack16(pl[9], pl[10]); words[6] = pack16(pl[11], pl[12]);
    words[7] = (pl[13] ? 1 : 0) | (pe << 8);
    words[8] = RECORD_TAG | (sw ? 1 : 0);
    if (sw) {
        words[9] = sw[0];
        words[10] = pack16(sw[2], sw[3]); words[11] = pack16(sw[4], sw[5]);
        words[12] = pack16(sw[6], sw[7]); words[13] = pack16(sw[8], sw[9]);
    }
}
static inline bool record_is_ours(const int32_t words[16]) { return (words[8] & ~1) == RECORD_TAG; }
static inline int record_count(const int32_t words[16]) { return words[4]; }
static inline int record_run_size(const int32_t words[16]) { return words[7] >> 8; }
// returns whether the record carries a split; sw receives its ten words then
static inline bool record_load(const int32_t words[16], int pl[PLAN_WORDS], int sw[10])
{
    unpack16(words[0], pl[0], pl[1]); unpack16(words[1], pl[2], pl[3]);
    unpack16(words[2], pl[4], pl[5]); unpack16(words[3], pl[6], pl[7]);
    pl[8] = words[4];
    unpack16(words[5], pl[9], pl[10]); unpack16(words[6], pl[11], pl[12]);
    pl[13] = words[7] & 1;
    const bool split = (words[8] & 1) != 0;
    if (split) {
        sw[0] = words[9]; sw[1] = pl[8] - words[9];
        unpack16(words[10], sw[2], sw[3]); unpack16(words[11], sw[4], sw[5]);
        unpack16(words[12], sw[6], sw[7]); unpack16(words[13], sw[8], sw[9]);
    }
    return split;
}

}  // namespace litho
