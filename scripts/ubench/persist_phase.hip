// persist_phase.hip -- what a PERSISTENT two-phase kernel (x-phase -> device-scope barrier -> y-phase -> barrier, per
// batch, accumulators never leaving the registers) could save over the engine's launch pair per batch, measured on the
// memory / synchronisation side alone.  Geometry = BASELINE config 3's batch: T = 12 items x 1025 rows x 2048 columns
// complex64 (201 MB, Infinity-Cache resident), 256 workgroups x 512 threads = one per CU (forced by 96 KB of LDS).
//   phase W: every workgroup writes its share of T (8-byte lanes, sc1 write-through stores, 64-byte runs: the x-pass's
//            pattern, rows dealt to workgroups as the engine deals them);
//   phase R: every workgroup reads a COLUMN block of T (what another XCD's workgroups wrote) with 16-byte loads and
//            folds it into registers; the launch-pair version then read-modify-writes a 16.8 MB slab, the persistent one
//            does not.
// Variants timed per batch:  A  two kernels per batch (W, then R + slab RMW)        = today's structure
//                            B  one persistent kernel, two grid barriers per batch  = the proposal
//                            C  the persistent kernel with empty phases             = the barrier cost itself
// The grid barrier = workgroup barrier, one agent-scope RELEASE fetch-add per workgroup, agent-scope ACQUIRE polling
// (the compiler emits the L2 write-back / invalidate the multi-XCD L2s need), workgroup barrier.  Phase R checks every
// value it reads (a stale line from an earlier batch would carry the wrong batch tag): `stale` must print 0.
// hipcc --offload-arch=gfx950 -O3 scripts/ubench/persist_phase.hip -o scripts/ubench/persist_phase.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

static constexpr int ROWS = 1025, COLS = 2048, ITEMS = 12, TC = 8, WGS = 256, THREADS = 512;
static constexpr size_t ITEM_ELEMS = (size_t)ROWS * COLS;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target)
{
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        // (bounded spin: if the 256 workgroups were ever not co-resident this must end as a wrong timing, not as a hung GPU)
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 18)) __builtin_amdgcn_s_sleep(1);
        if (spins >= (1u << 18)) __hip_atomic_fetch_add(counter + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // give-ups
    }
    __syncthreads();
}

// x-like: workgroup w owns rows a = w, w + 256, ... of every item; 8 bytes per lane, 64-byte runs in [tile][row][8] layout
__device__ __forceinline__ void phase_write(float2* T, int w, unsigned tag)
{
    for (int s = 0; s < ITEMS; ++s) {
        const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(T + (size_t)s * ITEM_ELEMS, 0, (int)(unsigned)(ITEM_ELEMS * 8), 0x00020000);
        for (int a = w; a < ROWS; a += WGS)
            for (int q = threadIdx.x; q < COLS; q += THREADS) {
                const unsigned off = (unsigned)((((size_t)(q / TC) * ROWS + a) * TC + (q % TC)) * 8);
                u2 v; v.x = tag; v.y = (unsigned)(a * COLS + q);
                __builtin_amdgcn_raw_buffer_store_b64(v, r, off, 0, 16 /* sc1 */);
            }
    }
}

// y-like: workgroup w owns the column tile w (8 columns x 1025 rows x 64 bytes contiguous per item): 16-byte loads
__device__ __forceinline__ unsigned phase_read(const float2* T, int w, unsigned tag, float& acc)
{
    unsigned stale = 0;
    for (int s = 0; s < ITEMS; ++s) {
        const u4* tile = reinterpret_cast<const u4*>(T + (size_t)s * ITEM_ELEMS + (size_t)w * ROWS * TC);
        for (int i = threadIdx.x; i < ROWS * TC / 2; i += THREADS) {
            const u4 v = tile[i];
            stale += (v.x != tag) + (v.z != tag);
            acc += __uint_as_float(v.y & 0x3fffffffu) + __uint_as_float(v.w & 0x3fffffffu);
        }
    }
    return stale;
}

__global__ __launch_bounds__(THREADS) void k_write(float2* T, unsigned tag) { phase_write(T, blockIdx.x, tag); }
__global__ __launch_bounds__(THREADS) void k_read(const float2* T, float* slab, unsigned tag, unsigned* stale_out)
{
    float acc = 0.f;
    const unsigned st = phase_read(T, blockIdx.x, tag, acc);
    if (st) atomicAdd(stale_out, st);
    // slab read-modify-write: 2048 x 2048 floats over 256 workgroups (the engine's flush: one slab per launch)
    float* mine = slab + (size_t)blockIdx.x * (COLS * COLS / WGS);
    for (int i = threadIdx.x; i < COLS * COLS / WGS; i += THREADS) mine[i] += acc;
}

template <bool WORK>
__global__ __launch_bounds__(THREADS) void k_persistent(float2* T, float* slab, unsigned* counter, int batches, unsigned tag0, unsigned* stale_out)
{
    extern __shared__ unsigned char force_one_wg_per_cu[];
    float acc = 0.f;
    unsigned st = 0, target = 0;
    for (int b = 0; b < batches; ++b) {
        if (WORK) phase_write(T, blockIdx.x, tag0 + b);
        target += WGS; grid_barrier(counter, target);
        if (WORK) st += phase_read(T, blockIdx.x, tag0 + b, acc);
        target += WGS; grid_barrier(counter, target);           // T may be overwritten only after everybody has read it
    }
    if (st) atomicAdd(stale_out, st);
    float* mine = slab + (size_t)blockIdx.x * (COLS * COLS / WGS);            // ONE flush at the end of the image
    for (int i = threadIdx.x; i < COLS * COLS / WGS; i += THREADS) mine[i] += acc;
    if (force_one_wg_per_cu[threadIdx.x] == 77 && acc == 1.f) slab[0] = 0.f;
}

int main()
{
    float2* T; float* slab; unsigned *counter, *stale;
    CK(hipMalloc(&T, ITEM_ELEMS * ITEMS * 8)); CK(hipMalloc(&slab, (size_t)COLS * COLS * 4));
    CK(hipMalloc(&counter, 8)); CK(hipMalloc(&stale, 4));
    CK(hipMemset(slab, 0, (size_t)COLS * COLS * 4)); CK(hipMemset(stale, 0, 4));
    const int B = 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 96 << 10;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_persistent<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_persistent<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    auto giveups = [&] { unsigned h[2]; CK(hipMemcpy(h, counter, 8, hipMemcpyDeviceToHost)); return h[1]; };
    auto stale_now = [&] { unsigned h; CK(hipMemcpy(&h, stale, 4, hipMemcpyDeviceToHost)); return h; };
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        // A: launch pair per batch
        CK(hipMemset(stale, 0, 4));
        hipEventRecord(e0);
        for (int b = 0; b < B; ++b) {
            hipLaunchKernelGGL(k_write, dim3(WGS), dim3(THREADS), 0, 0, T, 1000u * rep + b);
            hipLaunchKernelGGL(k_read, dim3(WGS), dim3(THREADS), 0, 0, T, slab, 1000u * rep + b, stale);
        }
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
        printf("A  two kernels per batch (write T; read T + slab RMW): %7.2f us per batch   stale %u\n", ms * 1e3 / B, stale_now());
        // B: persistent, two grid barriers per batch
        CK(hipMemset(counter, 0, 8)); CK(hipMemset(stale, 0, 4));
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_persistent<true>, dim3(WGS), dim3(THREADS), lds, 0, T, slab, counter, B, 5000u + 1000u * rep, stale);
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
        printf("B  persistent kernel, 2 grid barriers per batch      : %7.2f us per batch   stale %u   barrier give-ups %u\n", ms * 1e3 / B, stale_now(), giveups());
        // C: barriers only
        CK(hipMemset(counter, 0, 8));
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_persistent<false>, dim3(WGS), dim3(THREADS), lds, 0, T, slab, counter, 4 * B, 0u, stale);
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
        printf("C  persistent kernel, empty phases                   : %7.2f us per grid barrier\n", ms * 1e3 / (8 * B));
        // D: the two kernels alone, back to back (what each costs with its launch)
        hipEventRecord(e0);
        for (int b = 0; b < B; ++b) hipLaunchKernelGGL(k_write, dim3(WGS), dim3(THREADS), 0, 0, T, 7u);
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
        const float msw = ms;
        hipEventRecord(e0);
        for (int b = 0; b < B; ++b) hipLaunchKernelGGL(k_read, dim3(WGS), dim3(THREADS), 0, 0, T, slab, 7u, stale);
        hipEventRecord(e1); CK(hipEventSynchronize(e1)); hipEventElapsedTime(&ms, e0, e1);
        printf("D  write kernel alone %7.2f us, read + RMW kernel alone %7.2f us per launch\n", msw * 1e3 / B, ms * 1e3 / B);
    }
    return 0;
}
