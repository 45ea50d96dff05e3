// VALU issue-rate microbenchmark #2 for gfx950: explicit registers, so that operand count, operand kind
// (VGPR / SGPR / literal) and VGPR bank placement (index mod 4) are under control.
// Build: hipcc --offload-arch=gfx950 -O3 valu_bank.hip -o valu_bank.bin
// Each body is 32 independent instructions over v[8..135]; destinations rotate so that no instruction
// depends on one issued fewer than 16 instructions earlier.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <string>
#include <vector>

#define CLOB                                                                                                         \
    "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23",     \
        "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38",      \
        "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53",      \
        "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68",      \
        "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83",      \
        "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98",      \
        "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",        \
        "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124",       \
        "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "s40", "s41", "s42",  \
        "s43"

// 16 instructions; D/A/B/C are macros producing register names from the instruction index i
#define I16(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)

// register naming helpers: "v[8 + expr]" is legal assembler syntax
#define S_(x) #x
#define S(x) S_(x)

// --- scalar forms -------------------------------------------------------------------------------------------
// all operands in DIFFERENT banks: d = v[8+8i+0], a = +1, b = +2, c = +3
#define ADD_DIFF(i) "v_add_f32 v[8+8*" S(i) "+0], v[8+8*" S(i) "+1], v[8+8*" S(i) "+2]\n"
#define ADD_SAME(i) "v_add_f32 v[8+8*" S(i) "+0], v[8+8*" S(i) "+4], v[8+8*((" S(i) "+1)%%16)+4]\n"
#define FMA_DIFF(i) "v_fma_f32 v[8+8*" S(i) "+0], v[8+8*" S(i) "+1], v[8+8*" S(i) "+2], v[8+8*" S(i) "+3]\n"
// all three sources in bank 0 (indices = 0 mod 4), three different registers
#define FMA_SAME(i) "v_fma_f32 v[8+8*" S(i) "+1], v[8+8*" S(i) "+4], v[8+8*((" S(i) "+1)%%16)+4], v[8+8*((" S(i) "+2)%%16)+4]\n"
// two sources share a bank
#define FMA_TWO(i) "v_fma_f32 v[8+8*" S(i) "+0], v[8+8*" S(i) "+1], v[8+8*" S(i) "+5], v[8+8*" S(i) "+2]\n"
// one source is an SGPR
#define FMA_SGPR(i) "v_fma_f32 v[8+8*" S(i) "+0], v[8+8*" S(i) "+1], s40, v[8+8*" S(i) "+2]\n"
#define FMA_SGPR2(i) "v_fma_f32 v[8+8*" S(i) "+0], v[8+8*" S(i) "+1], s40, v[8+8*" S(i) "+5]\n"
// VOP2 fmac: d += a * b
#define FMAC_DIFF(i) "v_fmac_f32 v[8+8*" S(i) "+0], v[8+8*" S(i) "+1], v[8+8*" S(i) "+2]\n"
#define FMAC_LIT(i) "v_fmac_f32 v[8+8*" S(i) "+0], 0x3f6c835e, v[8+8*" S(i) "+2]\n"
#define FMAMK(i) "v_fmamk_f32 v[8+8*" S(i) "+0], v[8+8*" S(i) "+1], 0x3f6c835e, v[8+8*" S(i) "+2]\n"
#define MUL_DIFF(i) "v_mul_f32 v[8+8*" S(i) "+0], v[8+8*" S(i) "+1], v[8+8*" S(i) "+2]\n"
#define MUL_LIT(i) "v_mul_f32 v[8+8*" S(i) "+0], 0x3f6c835e, v[8+8*" S(i) "+2]\n"
// --- packed forms (register pairs: [lo:lo+1], lo even) ----------------------------------------------------------
#define PKADD(i) "v_pk_add_f32 v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+2:8+8*" S(i) "+3], v[8+8*" S(i) "+4:8+8*" S(i) "+5]\n"
#define PKMUL(i) "v_pk_mul_f32 v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+2:8+8*" S(i) "+3], v[8+8*" S(i) "+4:8+8*" S(i) "+5]\n"
#define PKMUL_BC(i) "v_pk_mul_f32 v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+2:8+8*" S(i) "+3], v[8+8*" S(i) "+4:8+8*" S(i) "+5] op_sel_hi:[1,0]\n"
#define PKMUL_SG(i) "v_pk_mul_f32 v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+2:8+8*" S(i) "+3], s[40:41]\n"
#define PKFMA(i) "v_pk_fma_f32 v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+2:8+8*" S(i) "+3], v[8+8*" S(i) "+4:8+8*" S(i) "+5], v[8+8*" S(i) "+6:8+8*" S(i) "+7]\n"
#define PKFMA_BC(i) "v_pk_fma_f32 v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+2:8+8*" S(i) "+3], v[8+8*" S(i) "+4:8+8*" S(i) "+5], v[8+8*" S(i) "+6:8+8*" S(i) "+7] op_sel_hi:[1,0,1]\n"
#define PKFMA_SG(i) "v_pk_fma_f32 v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+2:8+8*" S(i) "+3], s[40:41], v[8+8*" S(i) "+6:8+8*" S(i) "+7]\n"
// pk_fma whose accumulator is the destination (2 distinct source pairs + dst)
#define PKFMA_ACC(i) "v_pk_fma_f32 v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+2:8+8*" S(i) "+3], v[8+8*" S(i) "+4:8+8*" S(i) "+5], v[8+8*" S(i) "+0:8+8*" S(i) "+1]\n"
// swapped halves of a source (complex multiply needs (im, re)): op_sel:[1,0,0] op_sel_hi:[0,1,1]
#define PKFMA_SWAP(i) "v_pk_fma_f32 v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+2:8+8*" S(i) "+3], v[8+8*" S(i) "+4:8+8*" S(i) "+5], v[8+8*" S(i) "+6:8+8*" S(i) "+7] op_sel:[1,0,0] op_sel_hi:[0,1,1] neg_lo:[0,1,0]\n"
// mixed stream resembling a radix-2 butterfly with a twiddle: add, sub, mul, mul, fma, fma per 2 complex points
#define BFLY_SC(i) \
    "v_add_f32 v[8+8*" S(i) "+0], v[8+8*" S(i) "+0], v[8+8*" S(i) "+2]\n" \
    "v_sub_f32 v[8+8*" S(i) "+4], v[8+8*" S(i) "+0], v[8+8*" S(i) "+2]\n"
#define BFLY_PK(i) \
    "v_pk_add_f32 v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+0:8+8*" S(i) "+1], v[8+8*" S(i) "+2:8+8*" S(i) "+3]\n"

template <int KIND>
__global__ void k(float* out, int iters, float seed)
{
    asm volatile("s_mov_b32 s40, 0x3f800347\n s_mov_b32 s41, 0x3f7ff972\n" ::: "s40", "s41");
    // initialise v8..v135 with small finite numbers
    asm volatile(
        "v_cvt_f32_u32 v8, v0\n v_mul_f32 v8, 0x3a83126f, v8\n"
        "v_mov_b32 v9, v8\n v_mov_b32 v10, v8\n v_mov_b32 v11, v8\n v_mov_b32 v12, v8\n v_mov_b32 v13, v8\n v_mov_b32 v14, v8\n v_mov_b32 v15, v8\n" ::: "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15");
#define CP8(b) "v_mov_b32 v[" S(b) "+0], v8\n v_mov_b32 v[" S(b) "+1], v9\n v_mov_b32 v[" S(b) "+2], v10\n v_mov_b32 v[" S(b) "+3], v11\n v_mov_b32 v[" S(b) "+4], v12\n v_mov_b32 v[" S(b) "+5], v13\n v_mov_b32 v[" S(b) "+6], v14\n v_mov_b32 v[" S(b) "+7], v15\n"
    asm volatile(CP8(16) CP8(24) CP8(32) CP8(40) CP8(48) CP8(56) CP8(64) CP8(72) CP8(80) CP8(88) CP8(96) CP8(104) CP8(112) CP8(120) CP8(128) ::: CLOB);
    for (int i = 0; i < iters; ++i) {
#define BODY(F) asm volatile(I16(F) I16(F) I16(F) I16(F) ::: CLOB)
        if (KIND == 0) BODY(ADD_DIFF);
        if (KIND == 1) BODY(ADD_SAME);
        if (KIND == 2) BODY(FMA_DIFF);
        if (KIND == 3) BODY(FMA_SAME);
        if (KIND == 4) BODY(FMA_TWO);
        if (KIND == 5) BODY(FMA_SGPR);
        if (KIND == 6) BODY(FMA_SGPR2);
        if (KIND == 7) BODY(FMAC_DIFF);
        if (KIND == 8) BODY(FMAC_LIT);
        if (KIND == 9) BODY(FMAMK);
        if (KIND == 10) BODY(MUL_DIFF);
        if (KIND == 11) BODY(MUL_LIT);
        if (KIND == 12) BODY(PKADD);
        if (KIND == 13) BODY(PKMUL);
        if (KIND == 14) BODY(PKMUL_BC);
        if (KIND == 15) BODY(PKMUL_SG);
        if (KIND == 16) BODY(PKFMA);
        if (KIND == 17) BODY(PKFMA_BC);
        if (KIND == 18) BODY(PKFMA_SG);
        if (KIND == 19) BODY(PKFMA_ACC);
        if (KIND == 20) BODY(PKFMA_SWAP);
    }
    float r;
    asm volatile("v_add_f32 %0, v8, v16\n v_add_f32 %0, %0, v24\n v_add_f32 %0, %0, v33\n v_add_f32 %0, %0, v41" : "=v"(r) :: CLOB);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + seed;
}

template <int KIND>
void run(int waves_per_simd, float* d, const char* name, int flops_per_instr)
{
    const int iters = 2000, threads = 256;                      // 4 waves per block = 1 per SIMD
    const int blocks = 256 * waves_per_simd;                    // 256 CUs
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    k<KIND><<<blocks, threads>>>(d, 10, 1.f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(a);
        k<KIND><<<blocks, threads>>>(d, iters, 1.f);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        best = ms < best ? ms : best;
    }
    const double instr_per_simd = (double)iters * 64 * waves_per_simd;
    const double ns = best * 1e6 / instr_per_simd;
    printf("%-34s waves/SIMD=%d  %.3f ns/instr/SIMD (%.2f cyc @2.4GHz)  %.3f ns per lane-op  -> %.1f TFLOP/s chip\n", name,
           waves_per_simd, ns, ns * 2.4, ns / (flops_per_instr > 2 ? 2 : 1),
           flops_per_instr * 64.0 * 1024 / ns * 1e-3);
}

int main()
{
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float));
    for (int w : {1, 2, 4}) {
        run<0>(w, d, "v_add_f32 (banks differ)", 1);
        run<1>(w, d, "v_add_f32 (same bank)", 1);
        run<10>(w, d, "v_mul_f32", 1);
        run<11>(w, d, "v_mul_f32 literal", 1);
        run<2>(w, d, "v_fma_f32 3 vgpr (banks differ)", 2);
        run<3>(w, d, "v_fma_f32 3 vgpr (same bank)", 2);
        run<4>(w, d, "v_fma_f32 3 vgpr (two share)", 2);
        run<5>(w, d, "v_fma_f32 vgpr,sgpr,vgpr", 2);
        run<6>(w, d, "v_fma_f32 vgpr,sgpr,vgpr same bank", 2);
        run<7>(w, d, "v_fmac_f32 vgpr", 2);
        run<8>(w, d, "v_fmac_f32 literal", 2);
        run<9>(w, d, "v_fmamk_f32 literal", 2);
        run<12>(w, d, "v_pk_add_f32", 2);
        run<13>(w, d, "v_pk_mul_f32", 2);
        run<14>(w, d, "v_pk_mul_f32 op_sel broadcast", 2);
        run<15>(w, d, "v_pk_mul_f32 sgpr pair", 2);
        run<16>(w, d, "v_pk_fma_f32", 4);
        run<17>(w, d, "v_pk_fma_f32 op_sel broadcast", 4);
        run<18>(w, d, "v_pk_fma_f32 sgpr pair", 4);
        run<19>(w, d, "v_pk_fma_f32 acc=dst", 4);
        run<20>(w, d, "v_pk_fma_f32 swap+neg", 4);
    }
    return 0;
}
