/*
 * litho_abbe.h -- C ABI of the MI355X (gfx950) Abbe aerial-image engine.
 *
 * Drop-in boundary for the hot path of quarterwave0/LithographySimulator.  The
 * reference is pure Python over torch and has no FFI of its own; these entry points
 * are what a binding for that path would call (ctypes stub: INTEGRATION.md), one per
 * reference callable.  Each comment names the reference interface it replaces.
 *
 * Conventions (all functions):
 *   - return 0 on success, a negative LITHO_E_* code otherwise; never throw, never abort;
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - complex64 arrays are interleaved (re, im) fp32, row-major, exactly torch's layout;
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = default stream)
 *     and the call returns without waiting for it, except where a comment says
 *     "reads back": those calls copy a few bytes to the host and wait for `stream`;
 *   - nothing is allocated on behalf of the caller: scratch comes from the caller's
 *     `workspace` (size from litho_abbe_workspace_bytes);
 *   - pn = mask pixelNumber (even, 2..16384), N = FFT size from
 *     Mask.calculateEpsilonN (power of two, pn <= N <= 16384).
 */
#ifndef LITHO_ABBE_H
#define LITHO_ABBE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LITHO_OK 0
#define LITHO_E_ARG (-1)        /* bad size / null pointer / unsupported (pn odd, N not 2^k) */
#define LITHO_E_NSMALL (-2)     /* N < pn: the reference fails here too           */
#define LITHO_E_WORKSPACE (-3)  /* workspace too small                                        */
#define LITHO_E_HIP (-4)        /* a HIP runtime call failed; see litho_last_error()           */
#define LITHO_E_INDEX (-5)      /* aberration vector of length 4 (pupil.py:91-92)   */

/* Library version and the gfx target it was compiled for ("gfx950"). */
int litho_version(void);
const char *litho_target_arch(void);
/* Text of the last HIP error seen by this thread ("" if none). */
const char *litho_last_error(void);

/* ---- FFT sizing: Mask.calculateEpsilonN / _nearest2SqInt (mask.py:63-72). Host only. */
int litho_epsilon_n(double deltaK, double pixelSize, double wavelength,
                    double *epsilon_host, int *N_host);

/* ---- Source sampling: LightSource.generateAnnular (lightsource.py:34-50) and
 * generateQuasar (lightsource.py:52-73).  kind 0 = annular, 1 = quasar.  Writes the int64
 * 0/1 bitmap [pn,pn] the reference returns (fp16-exact sigma grid). */
int litho_source_bitmap(int kind, double sigma_in, double sigma_out, int pn,
                        double shift_x, double shift_y, int count, double rotation,
                        int64_t *bitmap, void *stream);

/* ---- imageformation.py:59 `(argwhere(lightsource) - pn//2).int()`: compacts any int64
 * bitmap [pn,pn] (non-zero = lit) into int32 [S,2] (dy,dx) in row-major order.
 * `shifts` must have room for pn*pn pairs (or `capacity` pairs; more lit pixels than
 * that is LITHO_E_ARG).  scratch: (pn+1) int32.  Reads back: *count_host = S.
 * Asynchronous form: count_host == NULL -- nothing is read back and the call does not wait; S stays on the
 * device in scratch[pn] (an int32) for litho_abbe_accumulate_counted. */
int litho_source_compact(const int64_t *bitmap, int pn, int32_t *shifts, int64_t capacity,
                         int32_t *scratch, int64_t *count_host, void *stream);

/* ---- Pupil: Pupil.generateWavefrontError / generatePupilFunction
 * (pupil.py:32-38, 46-111).  coeffs_f16_host: J fp16 bit patterns (uint16) in OSA/ANSI
 * order, as given by the caller, BEFORE the defocus rescale of coefficient 4; the
 * rescaled vector is written back to coeffs_f16_host (the reference mutates its
 * argument).  flags bit 0: skip that rescale (single-term generateZ,
 * pupil.py:46-77).  Outputs (either may be NULL): wavefront = fp16 W [pn,pn] as uint16
 * bit patterns, pupil = complex64 phi [pn,pn].  Asynchronous for J <= 32 (the term table
 * travels in the kernel arguments); longer vectors are staged through a transient device
 * buffer and the call waits for `stream`. */
int litho_pupil(uint16_t *coeffs_f16_host, int J, int pn, double NA, double wavelength, int flags,
                uint16_t *wavefront, void *pupil, void *stream);

/* ---- Through-focus pupil stack (the pupil export "batched over P defocus planes").  The reference has no
 * stack call: its counterpart is a Python loop `ab = aberrations.clone(); ab[4] = d_p; Pupil(pn, wavelength, NA, ab)
 * .generatePupilFunction()` per plane, i.e. pupil.py:88-100 + 102-111 run once per defocus value with the rescale of
 * pupil.py:91-92 applied to each d_p.  Here: ONE launch per 64 planes writes wavefront fp16 [planes,pn,pn] and / or pupil
 * complex64 [planes,pn,pn] in place; the sigma grid, r, theta and the J - 1 other Zernike terms are evaluated once per pixel,
 * the fp16 running sum is re-run per plane in the reference's order.  Bit-identical to `planes` litho_pupil calls.
 * coeffs_f16_host: J >= 5 fp16 bit patterns (coefficient 4 is ignored; J < 5: LITHO_E_INDEX, as `ab[4] = d` raises);
 * defocus_f16_host: `planes` fp16 bit patterns, the values of coefficient 4 BEFORE the rescale.  Nothing is written back
 * (the loop above works on clones).  Asynchronous for J <= 32; longer vectors run plane by plane through litho_pupil. */
int litho_pupil_stack(const uint16_t *coeffs_f16_host, int J, const uint16_t *defocus_f16_host, int planes, int pn,
                      double NA, double wavelength, uint16_t *wavefront, void *pupil, void *stream);

/* ---- generatePhi (pupil.py:102-111): pupil = exp(1j*2*pi*WE) for a complex64 wavefront
 * error WE [pn,pn], zero where the fp16 radius exceeds 1. */
int litho_pupil_phase(const void *wavefront_c64, int pn, void *pupil, void *stream);

/* ---- Workspace for the three calls below.  (For a mask size that runs embedded -- see litho_abbe_embedded_size -- this
 * includes the padded grid's regions and the padded copies; a smaller workspace of at least the size's own regions is
 * accepted and simply runs the problem un-embedded.) */
int litho_abbe_workspace_bytes(int pn, int N, size_t *bytes_host);

/* ---- Which grid a pn x pn problem RUNS at.  The specialised kernels (and the coarse grid) exist for pn = N and pn = N / 2;
 * every other even size -- a 1000^2 or 3000^2 mask; 10 nm pixels, where N = 4 pn -- is evaluated EMBEDDED: mask spectrum and
 * pupil centred in a zero-padded N / 2 (from 256 up) or N (from 1024 up) grid inside the workspace, the same shift list, the
 * centre pn x pn of the accumulated intensity added to `out`: the identical sum term by term, 1.7-3.6x faster than the
 * generic kernels.  The reference rolls the pupil modulo ITS grid (imageformation.py:63), so a source
 * list with a shift that wraps the pupil around the caller's grid is detected (from the plan read-back) and runs the general
 * path at the caller's size instead.  No reference counterpart: it runs any size through torch.fft (imageformation.py:32-45). */
int litho_abbe_embedded_size(int pn, int N, int *size_host);

/* ---- Abbe accumulation: the loop of abbeImage, imageformation.py:54-67.
 *   out[p][q] += sum_{s<S} | E_{p,s}[q] |^2,   E = calculateFFTAerial(roll(P_p, shift_s), M)
 * maskFT  complex64 [pn,pn]; pupil complex64 [planes,pn,pn] (planes >= 1: a through-focus
 * stack sharing maskFT and the source list); shifts int32 [S,2] = (dy,dx) =
 * (row - pn/2, col - pn/2); out fp32 [planes,pn,pn], accumulated into (the caller zeroes
 * it, and all-reduces it across GPUs when the source list is sharded).
 * Reads back 56 bytes once (pupil support box, its edge supports, shift extents, count) to plan the launch -- and 40 more,
 * once, when some but not all shifts of the list wrap the pupil around the grid (a shifted source: the list is then split on
 * the device into a part that keeps the fast paths and a part that needs the general one; options.split).
 * How the sum is evaluated is the library's business: for N = 2 pn, pn = 256 .. 4096 and a source list long
 * enough to repay it, the loop runs pn-point transforms on the grid q = 2 v and the fine image is reconstructed
 * once per call and plane; same result to rounding.  Environment, read once per call:
 * LITHO_ABBE_COARSE = 0 (never) / 1 (default: by source count) / 2 (whenever eligible); the other LITHO_ABBE_*
 * variables select kernel variants for parity tests and tuning. */
int litho_abbe_accumulate(const void *maskFT, const void *pupil, int planes,
                          const int32_t *shifts, int64_t S, int pn, int N, float *out,
                          void *workspace, size_t workspace_bytes, void *stream);

/* Same, with the source-point count left ON THE DEVICE by the asynchronous litho_source_compact:
 * count_dev = &scratch[pn] of that call, capacity = room in `shifts` (the count is clamped to it).  The count
 * comes back with the planning read-back, so compaction + accumulation + post-process of one image wait for the
 * stream exactly once.  *count_host (may be NULL) receives S. */
int litho_abbe_accumulate_counted(const void *maskFT, const void *pupil, int planes,
                                  const int32_t *shifts, const int32_t *count_dev, int64_t capacity,
                                  int pn, int N, float *out, void *workspace, size_t workspace_bytes,
                                  void *stream, int64_t *count_host);

/* ---- Plan reuse for sequences of images that share the pupil (stack) and the source list -- many masks through one
 * optical setting.  The two calls above read 56 bytes back to plan EVERY call (pupil support box, shift extents, count).
 * With a caller-held record the first call plans as usual and fills it; later calls with the same record issue no
 * planning launch and never wait for the stream (images can be queued back to back).  CONTRACT: the caller passes a
 * valid record only while `pupil`, `shifts` and the count are unchanged (set plan->valid = 0 after changing them); pn, N
 * and planes are checked.  count_dev may be NULL (then `capacity` is the number of source points, as in
 * litho_abbe_accumulate).  A source list that was SPLIT by the planning call (a shifted source: some shifts wrap the pupil
 * around the grid, see litho_abbe_accumulate) is split again by every planned call -- the record carries the two counts and
 * extents, the three small split kernels are re-run without a read-back -- so the non-wrapping points keep their fast path
 * (until round 5 a planned call ran the whole list on the general path).  litho_abbe_last_plan field [15]: 0 = planned
 * afresh, 1 = from the record, 2 = planned afresh and the list was split, 3 = split, from the record. */
typedef struct litho_abbe_plan {
    int32_t words[16];      /* the library's business (packed: plan words, source-point count, the grid size the run was planned
                             * for -- the call's own or the padded one of an embedded evaluation --, the outcome of the split;
                             * csrc/abbe_plan.hpp record_store).  A record another library version wrote is ignored (re-planned). */
    int32_t valid;          /* 0: empty, the call fills it; 1: use it */
    int32_t pn, N, planes;  /* what it was made for */
} litho_abbe_plan;
int litho_abbe_accumulate_planned(const void *maskFT, const void *pupil, int planes,
                                  const int32_t *shifts, const int32_t *count_dev, int64_t capacity,
                                  int pn, int N, float *out, void *workspace, size_t workspace_bytes,
                                  void *stream, litho_abbe_plan *plan, int64_t *count_host);

/* ---- The same call with the launch planner's options passed explicitly instead of through the process environment.
 * The reference has no counterpart (its loop has nothing to tune, imageformation.py:62-67); this is how tests, bench.py
 * and embedding applications select an evaluation path per CALL -- per thread, per stream -- without touching
 * LITHO_ABBE_* variables.  Every field: < 0 = not set (the LITHO_ABBE_<NAME> environment variable if present, else the
 * default); the meanings are those of DESIGN.md section 6.  `size` = sizeof(litho_abbe_options) as the caller compiled it
 * (fields beyond it count as not set, so the struct can grow).  plan, options, count_dev and count_host may each be NULL
 * (count_dev NULL: `capacity` is the number of source points). */
typedef struct litho_abbe_options {
    int32_t size;
    int32_t coarse;          /* 0 direct path, 1 coarse grid when the source list repays it (default), 2 whenever eligible */
    int32_t batch;           /* source points per launch pair (0 = automatic) */
    int32_t groups;          /* y-pass groups (partial-image slabs) per launch (0 = automatic) */
    int32_t xchunk;          /* source points per x-pass workgroup (0 = automatic) */
    int32_t tile;            /* T tile width in columns: 4, 8, 16 (0 = automatic) */
    int32_t plane_chunk;     /* planes of a stack in flight per launch pair (0 = automatic: 1) */
    int32_t w64, rect, w64_8192, xsplit, xrect, w64x, gcombine, rowpairs;   /* kernel families, DESIGN.md section 6 */
    int32_t force_generic, force_general;                                  /* runtime-predicated kernels / modular gather */
    int32_t poison;          /* 1: scratch starts the call as NaN bit patterns (tests) */
    int32_t embed;           /* 0: run mask sizes other than N and N / 2 on the generic kernels at their own size instead of
                              * embedded in the next such grid (litho_abbe_embedded_size; default 1) */
    int32_t split;           /* 0: a source list in which SOME shifts wrap the pupil around the grid (shifted, off-axis sources) runs
                              * the general path for every point instead of being split into a non-wrapping part (every fast
                              * path) and a wrapping one (general path) on the device (default 1: from 256 source points; 2: always) */
    int32_t coopdma;         /* 0: the 4096-point coarse-grid y-pass over 16-column tiles loads through registers (k_ypass_coop, round 3)
                              * instead of prefetching the next line by LDS-DMA (k_ypass_coop_dma, default 1) */
} litho_abbe_options;
int litho_abbe_accumulate_opts(const void *maskFT, const void *pupil, int planes, const int32_t *shifts,
                               const int32_t *count_dev, int64_t capacity, int pn, int N, float *out,
                               void *workspace, size_t workspace_bytes, void *stream, litho_abbe_plan *plan,
                               const litho_abbe_options *options, int64_t *count_host);

/* ---- Dry run of the launch planner: what litho_abbe_accumulate* WOULD do for a problem, without touching a device.
 * The reference has no counterpart (its loop has nothing to plan, imageformation.py:62-67); this exists so that the host
 * logic that replaced those six lines -- batching, kernel families, embedded evaluation of odd sizes, the split of a partly
 * wrapping source list, and above all WHERE in the caller's workspace every intermediate lives -- can be verified on a CPU
 * for every admissible size (tests/test_planner_cpu.py).  Inputs = what the call learns from its 56-byte read-back:
 * plan_words[14] (pupil support box rows lo/hi, columns lo/hi; shift extents dy lo/hi, dx lo/hi; source-point count;
 * rows lo/hi of non-zero samples on the natural box's column edges, columns lo/hi on its row edges (INT_MAX / INT_MIN =
 * none); corner flag), and -- needed only when the call would split the list -- split_words[10] (non-wrapping count,
 * wrapping count, dy lo/hi dx lo/hi of either part).  cus = compute units (<= 0: 256); workspace_bytes = what the caller
 * would pass (0: what litho_abbe_workspace_bytes reports).  Regions are byte ranges of the workspace.  Same code as the real
 * path (csrc/abbe_plan.hpp).  result->status = what the call would return before its first launch. */
typedef struct litho_abbe_region { int64_t offset, bytes; } litho_abbe_region;
typedef struct litho_abbe_dry_part {
    int32_t present;            /* 0: this part does not run */
    int32_t run_size;           /* grid it runs at: pn, or the padded size of an embedded evaluation */
    int32_t general, variant, coarse, natural_box, wave_y, xkind;   /* as litho_abbe_last_plan reports them */
    int32_t batch, planes_in_flight, groups, slabs, xchunk, tile;
    int64_t source_points;
    int64_t t_item_bytes;       /* one T item of the run's geometry */
    litho_abbe_region plan, twtab, twtab2, slab_region, slab_used, ic_used, chat_used, gam_used, T_region, T_used, recon_T_used,
                      embed_M, embed_P, embed_O;     /* (bytes 0 = not used by this part) */
} litho_abbe_dry_part;
typedef struct litho_abbe_dry_run {
    int32_t size;               /* sizeof(litho_abbe_dry_run) as the caller compiled it */
    int32_t status;             /* LITHO_OK, or the error the call would return */
    int32_t run_size, nowrap, split, reserved;
    int64_t workspace_bytes;    /* litho_abbe_workspace_bytes(pn, N) */
    litho_abbe_region list_a, list_b, split_counts;   /* the two lists of a split source list; the block counts (head of T, dead before the loops) */
    litho_abbe_dry_part part[2];   /* [0] the whole list, or the non-wrapping part of a split; [1] the wrapping part */
} litho_abbe_dry_run;
int litho_abbe_plan_dry_run(int pn, int N, int planes, const int32_t *plan_words, const int32_t *split_words,
                            const litho_abbe_options *options, int cus, size_t workspace_bytes, litho_abbe_dry_run *result);

/* ---- Single-point field: calculateFFTAerial(pf, maskFFFT, pixelNumber, N)
 * (imageformation.py:32-45).  field = complex64 [pn,pn].  Reads back the pupil's support box (56 bytes). */
int litho_abbe_field(const void *pf, const void *maskFT, int pn, int N, void *field,
                     void *workspace, size_t workspace_bytes, void *stream);

/* ---- Post-process: imageformation.py:69-77 (abs, bilinear resample by 1/epsilon,
 * zero pad; output size from litho_postprocess_size: 4096 -> 4094).
 * raw fp32 [planes,pn,pn] -> out fp32 [planes,n_out,n_out]. */
int litho_postprocess_size(int pn, double epsilon, int *n_out_host);
int litho_postprocess(const float *raw, int planes, int pn, double epsilon, float *out,
                      void *stream);

/* ---- The same pass with a constant-threshold resist model fused in (the reference lists "photoresist response
 * modeling, simple or otherwise" as an open goal, README.md:21; there is no reference code, so the definition is
 * this one): resist[p][y][x] = 1 where fp32(image * gain) >= fp32(threshold), else 0, on the post-processed
 * [planes,n_out,n_out] grid; gain = exposure dose (or dose / S for a normalised image).  out (the fp32 aerial image)
 * may be NULL when only the contour mask is wanted; resist uint8, required. */
int litho_postprocess_resist(const float *raw, int planes, int pn, double epsilon, double gain, double threshold,
                             float *out, uint8_t *resist, void *stream);

/* ---- Layout rasteriser: the device side of the GDSII import (lithographysimulator_amd/layout.py).  SURVEY.md section
 * 8(f) row 4: the reference has NO counterpart (README.md:20-22 lists GDSII import among its unbuilt goals), it is the
 * caller side of Mask(geometry, pixelSize) (mask.py:5-30), so there is no parity target; checked bit for bit against
 * oracle/layout_oracle.py.  edges: fp64 [n_edges][4] = (x0, y0, x1, y1) of closed, counter-clockwise polygons, same
 * length unit as x0 / y0 / pixel.  geometry int16 [pn][pn]: pixel (r, c) = 1 when its centre
 * (x0 + (c + 0.5) pixel, y0 + (r + 0.5) pixel) has a non-zero winding number (half-open: a centre on a left / bottom
 * edge is inside, on a right / top edge outside), else 0.  work: litho_rasterize_work_bytes(pn) device bytes. */
size_t litho_rasterize_work_bytes(int pn);
int litho_rasterize_edges(const double *edges, int64_t n_edges, int pn, double x0, double y0, double pixel, void *work,
                          size_t work_bytes, int16_t *geometry, void *stream);

/* ---- Mask spectrum pre-step: Mask._ffFraunhofer (mask.py:74-90).  geometry int16
 * [pn,pn]; spectrum complex64 [pn,pn].  Uses the same workspace as the Abbe calls. */
int litho_mask_spectrum(const int16_t *geometry, int pn, double epsilon, int N, void *spectrum,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- Introspection for bench.py / tests: what the last litho_abbe_accumulate on this
 * thread planned.  fields: [0]=mode (0 pruned box, 1 general/wrapping), [1]=box row0,
 * [2]=box col0, [3]=box rows, [4]=box cols, [5]=points per batch, [6]=x-pass launches,
 * [7]=kernel variant (-1 generic, else log2(N/pn) of the pruned specialisation), [8]=planes in flight per
 * launch pair (through-focus stacks), [9]=y-pass groups per plane, [10]=source points per x-pass workgroup,
 * [11]=x-pass kernel family (1 plane-fused k_xpass_abbe, 2 k_xpass_split, 3 k_xpass_rect, 0 fall-backs),
 * [12]=1 when the coarse-grid path ran (pn-point transforms on the grid q = 2 v, fine image reconstructed once
 * per plane), [13]=1 when the y-pass ran a wave-level kernel (k_ypass_rect / k_ypass_wave / k_ypass_pair), [14]=1 when
 * the pupil's support box lies inside the natural support |k| <= pn/4, [15]=1 when the call planned from a caller-held
 * record (litho_abbe_accumulate_planned), 2 when the source list was split into a non-wrapping and a wrapping part (the other
 * fields then describe the part that ran last, the wrapping one if there is one; [6] counts both). */
int litho_abbe_last_plan(int64_t fields_host[16]);

/* Names of the x-pass and y-pass kernels the last litho_abbe_accumulate on this thread launched in its source-point
 * loop, spelt as rocprofv3 prints them without "void litho::" and the argument list (e.g. "k_ypass_rect<11, 8, true>").
 * Each buffer holds `capacity` bytes (96 is enough); empty strings before the first call. */
int litho_abbe_last_kernels(char *xpass_host, char *ypass_host, size_t capacity);

/* ---- Per-kernel timing for bench.py: when on, litho_abbe_accumulate brackets every x-pass
 * and y-pass launch with HIP events recorded on `stream` (the first 4096 launch pairs of a call)
 * and waits for the last one before returning.  fields: [0]=x-pass total ms, [1]=x-pass
 * batches (one batch = the x-pass launches of one launch pair), [2]=T items (source point x plane) they
 * covered, [3..5]=the same for the y-pass, [6]=1 when the y-pass ran the wave-per-line kernel
 * (k_ypass_wave) instead of k_ypass_acc, [7]=planes in flight per launch pair. */
int litho_abbe_set_profiling(int on);
int litho_abbe_last_profile(double fields_host[8]);

#ifdef __cplusplus
}
#endif
#endif /* LITHO_ABBE_H */
