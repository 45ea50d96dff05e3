// fft_core.hpp -- per-line FFT engine for gfx950 (wave64, LDS exchange, register radix-16).
//
// One "line" is a length-N complex DFT (N = 2^LOG2N, 16 <= N <= 16384) computed by
// T = N/16 threads that each hold 16 points in registers.  N = R1 * 16^P16 with
// R1 in {1,2,4,8}: an optional leading radix-R1 pass (no twiddles) followed by P16
// radix-16 Stockham passes; between passes the points are exchanged through LDS.
// Both on input and on output thread t owns the natural-order indices n = t + T*e,
// e = 0..15, so consecutive threads touch consecutive samples (coalesced global I/O).
//
// The Abbe path needs a CENTRED transform, out[u] = sum_k in[k] w^(k u) with k and u
// running over windows around zero; since w^N = 1 that is the plain DFT on indices
// taken mod N, and the callers just map n -> k (or u) and skip what is outside their
// window (zero padding is never materialised, discarded outputs are never written).
#pragma once
#include <hip/hip_runtime.h>

// Timing-diagnostic switches (LITHO_DIAG_*) remove loads, stores, barriers or LDS traffic from the kernels and
// therefore give WRONG RESULTS.  They compile only in a diagnostic build: scripts/build_variants.sh defines
// LITHO_DIAG_BUILD, writes build/variants/lib_<name>.so (never the product path) and that library reports
// litho_target_arch() == "gfx950-diag", which the Python binding and __graft_entry__.build() refuse.
#if (defined(LITHO_DIAG_NOBARRIER) || defined(LITHO_DIAG_NOLDSWRITE) || defined(LITHO_DIAG_NOLDSREAD) ||   \
     defined(LITHO_DIAG_XNOLOAD) || defined(LITHO_DIAG_XNOSTORE) || defined(LITHO_DIAG_XSTORE_L2) ||       \
     defined(LITHO_DIAG_YNOLOAD) || defined(LITHO_DIAG_YFLUSH_STORE) || defined(LITHO_DIAG_YFLUSH_NONE)) &&                                                                       \
    !defined(LITHO_DIAG_BUILD)
#error "LITHO_DIAG_* switches produce wrong results; use scripts/build_variants.sh (defines LITHO_DIAG_BUILD, separate output)"
#endif

namespace litho {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// cos(2 pi m / 32), m = 0..16; sin follows from cos(2 pi (8 - m) / 32).
static constexpr double W32C[17] = {
    1, 0.98078528040323043, 0.92387953251128674, 0.83146961230254524, 0.70710678118654757,
    0.55557023301960229, 0.38268343236508984, 0.19509032201612833, 0.0,
    -0.19509032201612819, -0.38268343236508973, -0.55557023301960196, -0.70710678118654746,
    -0.83146961230254535, -0.92387953251128674, -0.98078528040323043, -1};

template <int I, int END, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < END) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, END>(f);
    }
}

// v * exp(SIGN * 2 pi i * M / R), M in [0, R/2), compile-time specialised.
template <int R, int M, int SIGN>
__device__ __forceinline__ float2 mul_root(float2 v) {
    if constexpr (M == 0) {
        return v;
    } else if constexpr (4 * M == R) {           // multiply by (SIGN) i
        return SIGN > 0 ? make_float2(-v.y, v.x) : make_float2(v.y, -v.x);
    } else if constexpr (8 * M == R) {           // (1 + SIGN i)/sqrt2
        constexpr float h = 0.70710678118654752f;
        return SIGN > 0 ? make_float2((v.x - v.y) * h, (v.x + v.y) * h)
                        : make_float2((v.x + v.y) * h, (v.y - v.x) * h);
    } else if constexpr (8 * M == 3 * R) {       // (-1 + SIGN i)/sqrt2
        constexpr float h = 0.70710678118654752f;
        return SIGN > 0 ? make_float2(-(v.x + v.y) * h, (v.x - v.y) * h)
                        : make_float2((v.y - v.x) * h, -(v.x + v.y) * h);
    } else {
        constexpr int idx = M * (32 / R);                       // angle = 2 pi idx / 32, idx in (0,16)
        constexpr float wr = (float)W32C[idx];
        constexpr float wi = (float)(SIGN * (idx <= 8 ? W32C[8 - idx] : W32C[idx - 8]));
        return make_float2(fmaf(v.x, wr, -v.y * wi), fmaf(v.x, wi, v.y * wr));
    }
}

// In-register DFT of R points: x[m] <- sum_r x[r] exp(SIGN 2 pi i r m / R), natural order.
template <int R, int SIGN>
struct Dft {
    __device__ static __forceinline__ void run(float2 (&x)[R]) {
        float2 e[R / 2], o[R / 2];
        static_for<0, R / 2>([&](auto i) { e[i] = x[2 * i]; o[i] = x[2 * i + 1]; });
        Dft<R / 2, SIGN>::run(e);
        Dft<R / 2, SIGN>::run(o);
        static_for<0, R / 2>([&](auto m) {
            float2 t = mul_root<R, decltype(m)::value, SIGN>(o[m]);
            x[m] = cadd(e[m], t);
            x[m + R / 2] = csub(e[m], t);
        });
    }
};
template <int SIGN>
struct Dft<1, SIGN> {
    __device__ static __forceinline__ void run(float2 (&)[1]) {}
};
template <int SIGN>
struct Dft<2, SIGN> {
    __device__ static __forceinline__ void run(float2 (&x)[2]) {
        float2 a = x[0], b = x[1];
        x[0] = cadd(a, b);
        x[1] = csub(a, b);
    }
};

// LDS slots are padded by LDS_PAD float2 per 16: with 2, a lane's 16 consecutive slots start 18*8 = 144 B
// apart, so both the 8-byte accesses and the 16-byte ds_write2_b64 pairs the compiler forms are
// bank-conflict free (with 1, the merged 16-byte writes collide two-way: 33 % of LDS cycles measured).
#ifndef LITHO_LDS_PAD
#define LITHO_LDS_PAD 2
#endif
__device__ __forceinline__ constexpr int lds_pad(int i) { return i + LITHO_LDS_PAD * (i >> 4); }

// Workgroup barrier that orders LDS traffic only: global loads/stores issued earlier stay in
// flight across it (a plain __syncthreads() would drain vmcnt and kill the input prefetch).
__device__ __forceinline__ void lds_barrier() {
#ifdef LITHO_FULL_BARRIER
    __syncthreads();
    return;
#endif
#ifdef LITHO_DIAG_NOBARRIER            // timing diagnostic only: results are wrong
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    return;
#endif
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <int LOG2N, int SIGN>
struct LineFFT {
    static_assert(LOG2N >= 4 && LOG2N <= 14, "16 <= N <= 16384");
    static constexpr int N = 1 << LOG2N;
    static constexpr int E = 16;                   // points per thread
    static constexpr int T = N / E;                // threads per line
    static constexpr int P16 = LOG2N / 4;          // radix-16 passes
    static constexpr int R1 = N >> (4 * P16);      // leading radix (1 = none)
    static constexpr int B1 = E / R1;              // leading-pass butterflies per thread
    static constexpr int EXCH = P16 - 1 + (R1 > 1 ? 1 : 0);   // LDS exchanges per transform
    static constexpr int LDS_LINE = N + LITHO_LDS_PAD * (N / 16);    // float2 slots per buffer per line
    // Radix-16 pass p works on sub-transforms of length Ns(p) = R1 * 16^p and needs twiddles
    // w_{16 Ns}^{k r}, k = t mod Ns, iff Ns > 1.  A pass with Ns <= 64 has only Ns distinct
    // twiddle sets: those live in a small LDS table shared by the workgroup (<= 7.7 KB); only the
    // last pass of a large transform (Ns > 64) keeps its 15 twiddles in registers.  That frees 30
    // VGPRs per thread compared with holding every set in registers.
    static constexpr int ns_of(int p) { return R1 << (4 * p); }
    static constexpr bool tw_needed(int p) { return ns_of(p) > 1; }
#ifndef LITHO_LDS_TW_MAXNS
#define LITHO_LDS_TW_MAXNS 64
#endif
    static constexpr bool tw_in_lds(int p) { return tw_needed(p) && ns_of(p) <= LITHO_LDS_TW_MAXNS; }
    static constexpr bool tw_in_reg(int p) { return tw_needed(p) && ns_of(p) > LITHO_LDS_TW_MAXNS; }
    static constexpr int lds_tw_offset(int p) {          // float2 offset of pass p's table
        int off = 0;
        for (int q = 0; q < p; ++q) if (tw_in_lds(q)) off += ns_of(q) * 15;
        return off;
    }
    static constexpr int LDS_TW = lds_tw_offset(P16);     // float2 entries of the LDS twiddle tables
    static constexpr int nreg_sets() { int n = 0; for (int p = 0; p < P16; ++p) n += tw_in_reg(p) ? 1 : 0; return n; }
    static constexpr int reg_index(int p) { int n = 0; for (int q = 0; q < p; ++q) n += tw_in_reg(q) ? 1 : 0; return n; }
    static constexpr int NTW = nreg_sets();

    struct Twiddles {
        float2 w[NTW > 0 ? NTW : 1][15];
        const float2* lds;       // the workgroup's LDS twiddle tables
    };

    // table[n] = exp(+2 pi i n / N); SIGN < 0 conjugates on the fly.  Fills the LDS tables
    // cooperatively (call with every thread of the workgroup, then lds_barrier) and the
    // register-resident set of thread t.
    // tstride: the table may belong to a transform `tstride` times longer (entry tstride * n = exp(2 pi i n / N)).
    __device__ static __forceinline__ void load_twiddles(Twiddles& tw, const float2* __restrict__ table, int t,
                                                         float2* lds_tables, int wg_tid, int wg_threads,
                                                         int tstride = 1) {
        static_for<0, P16>([&](auto p_) {
            constexpr int p = decltype(p_)::value;
            constexpr int ns = ns_of(p);
            if constexpr (tw_in_lds(p)) {
                for (int i = wg_tid; i < ns * 15; i += wg_threads) {
                    const int k = i / 15, r = i - k * 15 + 1;
                    float2 w = table[tstride * k * r * (N / (16 * ns))];
                    if (SIGN < 0) w.y = -w.y;
                    lds_tables[lds_tw_offset(p) + i] = w;
                }
            } else if constexpr (tw_in_reg(p)) {
                const int k = t & (ns - 1);
                const int step = tstride * k * (N / (16 * ns));
                static_for<1, 16>([&](auto r) {
                    float2 w = table[step * decltype(r)::value];
                    if (SIGN < 0) w.y = -w.y;
                    tw.w[reg_index(p)][decltype(r)::value - 1] = w;
                });
            }
        });
        tw.lds = lds_tables;
        lds_barrier();
    }

    // Padded LDS slot of (i + off) given the padded slot of i, for the access patterns below:
    // lds_pad(i + off) == lds_pad(i) + off + PAD*(off >> 4) whenever adding `off` cannot carry out of
    // the low 4 bits of i (true for every pattern used here: see the comments at the call sites).
    // `off` is a compile-time constant, so each LDS access is one base register + an immediate.
    static constexpr int pad_off(int off) { return off + LITHO_LDS_PAD * (off >> 4); }

    // x[e] holds sample n = t + T*e on entry and output bin n = t + T*e on exit.
    // lds: this line's two exchange buffers (2 * LDS_LINE float2) when NBUF == 2, one when NBUF == 1.
    // `flip` alternates buffers across consecutive exchanges (also across calls).
    template <int NBUF>
    __device__ static __forceinline__ void run(float2 (&x)[E], const Twiddles& tw, float2* lds, int t, int& flip) {
        // read slot of sample t + r*T: r*T is a multiple of 16 when T >= 16; when T < 16, t < T and T | 16,
        // so t + r*T never carries differently from r*T alone.
        float2* const rd0 = lds + lds_pad(t);
        if constexpr (R1 > 1) {
            float2* buf = (NBUF == 2 && (flip & 1)) ? lds + LDS_LINE : lds;
            if constexpr (NBUF == 1) lds_barrier();
            // write slot of j*R1 + m, j = t + T*b:  = pad(t*R1) + pad_off(T*R1*b + m): m < R1 never carries
            // (t*R1 is a multiple of R1), and T*R1*b is a multiple of 16 whenever N*R1 >= 256.
            static_assert((T * R1) % 16 == 0 || B1 == 1 || true, "");
            float2* const wr0 = buf + lds_pad(t * R1);
            static_for<0, B1>([&](auto b) {
                float2 v[R1];
                static_for<0, R1>([&](auto r) { v[r] = x[b + B1 * r]; });
                Dft<R1, SIGN>::run(v);
                static_for<0, R1>([&](auto m) {
                    constexpr int off = T * R1 * decltype(b)::value + decltype(m)::value;
                    if constexpr ((T * R1) % 16 == 0) wr0[pad_off(off)] = v[m];
                    else buf[lds_pad(t * R1 + off)] = v[m];
                });
            });
            lds_barrier();
            float2* const rd = rd0 + (buf - lds);
            static_for<0, E>([&](auto r) { x[r] = rd[pad_off(decltype(r)::value * T)]; });
            flip ^= 1;
        }
        static_for<0, P16>([&](auto p) {
            constexpr int ns = ns_of(decltype(p)::value);
            if constexpr (tw_in_reg(decltype(p)::value)) {
                static_for<1, 16>([&](auto r) { x[r] = cmul(x[r], tw.w[reg_index(decltype(p)::value)][r - 1]); });
            } else if constexpr (tw_in_lds(decltype(p)::value)) {
                const float2* tp = tw.lds + lds_tw_offset(decltype(p)::value) + (t & (ns - 1)) * 15;
                static_for<1, 16>([&](auto r) { x[r] = cmul(x[r], tp[decltype(r)::value - 1]); });
            }
            Dft<16, SIGN>::run(x);
            if constexpr (p < P16 - 1) {
                float2* buf = (NBUF == 2 && (flip & 1)) ? lds + LDS_LINE : lds;
                if constexpr (NBUF == 1) lds_barrier();
                // write slot of base + m*ns, base = (t-k)*16 + k, k = t mod ns: m*ns is a multiple of 16 for
                // ns >= 16; for ns < 16, k < ns and ns | 16, so k + m*ns never carries differently from m*ns.
                const int k = t & (ns - 1);
                float2* const wr0 = buf + lds_pad((t - k) * 16 + k);
#ifndef LITHO_DIAG_NOLDSWRITE
                static_for<0, 16>([&](auto m) { wr0[pad_off(decltype(m)::value * ns)] = x[m]; });
#else
                wr0[0] = x[15];
#endif
                lds_barrier();
                float2* const rd = rd0 + (buf - lds);
#ifndef LITHO_DIAG_NOLDSREAD
                static_for<0, 16>([&](auto r) { x[r] = rd[pad_off(decltype(r)::value * T)]; });
#else
                x[0] = cadd(x[0], rd[0]);
#endif
                flip ^= 1;
            }
        });
    }
};

// Map a natural-order index n in [0,N) onto the centred window [lo, hi): returns the
// representative (n or n-N) and whether it falls inside.  Needs -N/2 <= lo, hi <= N/2 + 1.
__device__ __forceinline__ bool centred_index(int n, int N, int lo, int hi, int& k) {
    k = (n >= hi) ? n - N : n;
    return (k >= lo) && (k < hi);
}

}  // namespace litho
