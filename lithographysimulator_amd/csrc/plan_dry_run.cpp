// plan_dry_run.cpp -- litho_abbe_plan_dry_run: the launch planner's decisions for a problem, without a device.
// Plain C++ (no HIP header, no HIP call): linked into liblitho_abbe.so, and compiled on its own with g++ by
// tests/test_planner_cpu.py.  Every decision comes from csrc/abbe_plan.hpp, the same functions abbe_engine.hip launches from.
#include <cstring>

#include "abbe_plan.hpp"

namespace litho {
namespace {

litho_abbe_region reg(size_t off, size_t bytes) { return litho_abbe_region{(int64_t)off, (int64_t)bytes}; }
litho_abbe_region reg(const Region& r) { return reg(r.off, r.bytes); }

// one run of the source-point loop (accumulate_planned) at grid size `pr` inside the layout `l`, whose T region is `t_bytes` long
int fill_part(litho_abbe_dry_part& d, const WsLayout& l, size_t t_bytes, const Knobs& kn, const int pl[PLAN_WORDS], int pr, int N,
              int planes, int64_t S, int cus)
{
    RunPlan rp;
    const int rc = plan_run(rp, t_bytes, kn, pl, pr, N, planes, S, cus, true);
    if (rc) return rc;
    const AbbePlan& run = rp.run();
    d.present = 1;
    d.run_size = pr;
    d.general = run.general; d.variant = rp.direct.variant; d.coarse = rp.coarse ? 1 : 0;
    d.natural_box = rp.direct.natural_box ? 1 : 0; d.wave_y = run.wave_y ? 1 : 0;
    d.xkind = run.fused_x ? 1 : (run.split_x ? 2 : (run.rect_x ? 3 : 0));
    d.batch = (int32_t)run.bs; d.planes_in_flight = run.PC; d.groups = run.G; d.slabs = run.slabs; d.xchunk = run.xchunk;
    d.tile = 1 << run.g.tcl;
    d.source_points = S;
    d.t_item_bytes = (int64_t)((size_t)run.g.t_point * SIZEOF_FLOAT2);
    const size_t slab_plane = (size_t)run.g.nt * 4 * pr * sizeof(float);
    d.plan = reg(l.plan);
    d.twtab = reg(l.twtab);
    d.twtab2 = rp.coarse ? reg(l.twtab2) : reg(l.twtab2.off, 0);
    d.slab_region = reg(l.slab);
    d.slab_used = reg(l.slab.off, (size_t)run.PC * run.G * slab_plane);          // zero_slabs: PC planes x G slabs (stride G)
    d.T_region = reg(l.T.off, t_bytes);
    d.T_used = reg(l.T.off, (size_t)run.PC * (size_t)run.bs * (size_t)d.t_item_bytes);
    if (rp.coarse) {
        d.ic_used = reg(l.ic.off, (size_t)run.PC * pr * pr * sizeof(float));
        d.chat_used = reg(l.chat);
        d.gam_used = reg(l.gam);
        d.recon_T_used = reg(l.T.off, reconstruct_t_bytes(pr));
    } else {
        d.ic_used = reg(l.ic.off, 0); d.chat_used = reg(l.chat.off, 0); d.gam_used = reg(l.gam.off, 0); d.recon_T_used = reg(l.T.off, 0);
    }
    return LITHO_OK;
}

// the embedded evaluation of a non-wrapping list (accumulate_embedded): padded layout, scratch behind it, T cut at t_end_pad
int fill_embedded(litho_abbe_dry_part& d, const Knobs& kn, const int pl[PLAN_WORDS], int pn, int pe, int N, int planes, int64_t S,
                  size_t t_end_pad, int cus)
{
    const WsLayout l2 = ws_layout(pe, N);
    size_t t_bytes = l2.T.bytes;
    if (t_end_pad > 0 && l2.T.off + t_bytes > t_end_pad) t_bytes = t_end_pad - l2.T.off;
    int pl2[PLAN_WORDS];
    embed_plan_words(pl, pn, pe, pl2);
    const int pc = planes < COARSE_PLANES ? planes : COARSE_PLANES;        // planes are padded COARSE_PLANES at a time
    const int rc = fill_part(d, l2, t_bytes, kn, pl2, pe, N, pc, S, cus);
    if (rc) return rc;
    const EmbedLayout el = embed_layout(pe, N);
    d.embed_M = reg(el.M2);
    d.embed_P = reg(el.P2.off, (size_t)pc * pe * pe * SIZEOF_FLOAT2);
    d.embed_O = reg(el.O2.off, (size_t)pc * pe * pe * sizeof(float));
    return LITHO_OK;
}

}  // namespace
}  // namespace litho

extern "C" int litho_abbe_plan_dry_run(int pn, int N, int planes, const int32_t* plan_words, const int32_t* split_words,
                                       const litho_abbe_options* options, int cus, size_t workspace_bytes, litho_abbe_dry_run* result)
{
    using namespace litho;
    if (!result || !plan_words || result->size < (int32_t)sizeof(litho_abbe_dry_run) || planes < 1) return LITHO_E_ARG;
    int rc = check_sizes(pn, N);
    if (rc) return rc;
    if (options && (options->size < (int32_t)sizeof(int32_t) || options->size > 4096)) return LITHO_E_ARG;
    const int32_t size = result->size;
    memset(result, 0, sizeof(*result));
    result->size = size;
    if (cus <= 0) cus = 256;
    const Knobs kn = Knobs::read(options);
    result->workspace_bytes = (int64_t)litho::workspace_bytes(pn, N);
    const size_t ws_bytes = workspace_bytes ? workspace_bytes : (size_t)result->workspace_bytes;
    const WsLayout own = ws_layout(pn, N);
    if (ws_bytes < own.total) { result->status = LITHO_E_WORKSPACE; return LITHO_OK; }
    int pl[PLAN_WORDS];
    for (int i = 0; i < PLAN_WORDS; ++i) pl[i] = plan_words[i];
    const int64_t S = pl[8];
    // what the planning kernels can produce, nothing else: a support box inside the grid (or the empty box INT_MAX / INT_MIN),
    // shifts no longer than the grid, a count that fits the grid's pixels, edge supports inside the grid or empty
    const bool empty_box = pl[1] < pl[0];
    if (!empty_box && (pl[0] < 0 || pl[1] >= pn || pl[2] < 0 || pl[3] >= pn || pl[3] < pl[2])) return LITHO_E_ARG;
    if (S < 0 || S > (int64_t)pn * pn) return LITHO_E_ARG;
    if (S > 0 && (pl[4] > pl[5] || pl[6] > pl[7] || pl[4] < -pn || pl[5] > pn || pl[6] < -pn || pl[7] > pn)) return LITHO_E_ARG;
    for (int e = 9; e <= 11; e += 2)
        if (pl[e + 1] >= pl[e] && (pl[e] < 0 || pl[e + 1] >= pn)) return LITHO_E_ARG;
    const int pe = run_size(pn, N, kn, ws_bytes);
    result->run_size = pe;
    const bool nowrap = list_nowrap(pl, pn);
    result->nowrap = nowrap ? 1 : 0;
    if (S <= 0 || pl[1] < pl[0]) { result->status = LITHO_OK; return LITHO_OK; }      // nothing to add
    if (split_wanted(kn, nowrap, false, S)) {
        const SplitLayout sl = split_layout(pn, pe, N, S);
        if (sl.ok) {
            if (!split_words) return LITHO_E_ARG;
            if (split_words[0] < 0 || split_words[1] < 0 || (int64_t)split_words[0] + split_words[1] != S) return LITHO_E_ARG;
            for (int part = 0; part < 2; ++part)
                if (split_words[part] > 0 && (split_words[2 + 4 * part] > split_words[3 + 4 * part] || split_words[4 + 4 * part] > split_words[5 + 4 * part] ||
                                              split_words[2 + 4 * part] < -pn || split_words[3 + 4 * part] > pn ||
                                              split_words[4 + 4 * part] < -pn || split_words[5 + 4 * part] > pn)) return LITHO_E_ARG;
            result->split = 1;
            result->list_a = reg(sl.list_a); result->list_b = reg(sl.list_b); result->split_counts = reg(sl.counts);
            for (int part = 0; part < 2; ++part) {
                const int64_t n = split_words[part];
                if (n <= 0) continue;
                int plp[PLAN_WORDS];
                for (int i = 0; i < PLAN_WORDS; ++i) plp[i] = pl[i];
                for (int i = 0; i < 4; ++i) plp[4 + i] = split_words[2 + 4 * part + i];
                plp[8] = (int)n;
                if (part == 0 && pe != pn) rc = fill_embedded(result->part[0], kn, plp, pn, pe, N, planes, n, sl.t_end_pad, cus);
                else rc = fill_part(result->part[part], own, sl.t_bytes_own, kn, plp, pn, N, planes, n, cus);
                if (rc) { result->status = rc; return LITHO_OK; }
            }
            return LITHO_OK;
        }
    }
    if (pe == pn || !nowrap || kn.force_general) rc = fill_part(result->part[0], own, own.T.bytes, kn, pl, pn, N, planes, S, cus);
    else rc = fill_embedded(result->part[0], kn, pl, pn, pe, N, planes, S, 0, cus);
    result->status = rc;
    return LITHO_OK;
}
