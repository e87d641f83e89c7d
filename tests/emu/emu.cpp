// CPU fiber emulator for the wave-level kernels (TEST INFRASTRUCTURE ONLY, never linked into the product).
// Compiles circkit_amd/csrc/canon_*.h against tests/emu/wave_prims_emu.h: each wavefront is 64 ucontext fibers scheduled
// round-robin; every collective primitive is a rendezvous of all 64 fibers (a lane that skips a
// collective deadlocks the wave, which is reported -- the same discipline the GPU build relies on).
#define CK_WAVE_PRIMS_OVERRIDE "../../tests/emu/wave_prims_emu.h"      // (relative to circkit_amd/csrc/wave_prims.h)
#include <ucontext.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "../../circkit_amd/csrc/wave_prims.h"

namespace ck { namespace emu {
// A workgroup = NW waves of 64 fibers.  Wave-level collectives rendezvous the 64 fibers of one wave; block_barrier()
// rendezvous all fibers of the workgroup.  run_wave() is the 1-wave special case.
constexpr int MAXW = 16;
struct WaveSync {
    uint64_t buf[2][64];
    int cnt[2];
    uint32_t complete[2];
};
struct BlockState {
    ucontext_t main_ctx, ctx[64 * MAXW];
    std::vector<char> stacks;
    int nfib = 64, cur = 0;
    WaveSync ws[MAXW];
    uint32_t gen[64 * MAXW];
    bool done[64 * MAXW];
    int bar_cnt[2];
    uint32_t bar_complete[2], bar_gen[64 * MAXW];
    void (*body)(void*) = nullptr;
    void* arg = nullptr;
};
static BlockState* B = nullptr;

uint32_t cur_lane() { return (uint32_t)(B->cur & 63); }
uint32_t cur_wave() { return (uint32_t)(B->cur >> 6); }

static void yield_next(const char* what)
{
    int from = B->cur, nx = from;
    for (int t = 0; t < B->nfib; ++t) {
        nx = (nx + 1) % B->nfib;
        if (!B->done[nx]) break;
    }
    if (nx == from || B->done[nx]) {
        fprintf(stderr, "emu: deadlock -- fiber %d (wave %d lane %d) waits in a %s the others never reach\n", from, from >> 6,
                from & 63, what);
        abort();
    }
    B->cur = nx;
    swapcontext(&B->ctx[from], &B->ctx[nx]);
}

void gather(uint64_t v, uint64_t out[64])
{
    const int f = B->cur, l = f & 63;
    WaveSync& w = B->ws[f >> 6];
    const uint32_t g = B->gen[f]++;
    const int s = g & 1;
#ifdef CK_EMU_CHECK_SITES
    // every lane must be in the SAME collective: compare the call sites (build with -O0 -DCK_EMU_CHECK_SITES)
    static void* site[MAXW][2][64];
    site[f >> 6][s][l] = __builtin_return_address(1);
    if (w.cnt[s] > 0 && site[f >> 6][s][l] != site[f >> 6][s][(l + 63) & 63] && false) {}
#endif
    w.buf[s][l] = v;
    if (++w.cnt[s] == 64) {
#ifdef CK_EMU_CHECK_SITES
        for (int i = 1; i < 64; ++i)
            if (site[f >> 6][s][i] != site[f >> 6][s][0]) {
                Dl_info base;
                dladdr((void*)&gather, &base);              // offsets into the shared object: addr2line -f -C -e libcanon_emu.so <offset>
                fprintf(stderr, "emu: lanes 0 and %d are in different collectives: %#lx vs %#lx\n", i,
                        (unsigned long)((char*)site[f >> 6][s][0] - (char*)base.dli_fbase), (unsigned long)((char*)site[f >> 6][s][i] - (char*)base.dli_fbase));
                abort();
            }
#endif
        w.complete[s] = g; w.cnt[s] = 0;
    }
    while (w.complete[s] != g) yield_next("wave collective");
    for (int i = 0; i < 64; ++i) out[i] = w.buf[s][i];
}

void block_barrier()
{
    const int f = B->cur;
    const uint32_t g = B->bar_gen[f]++;
    const int s = g & 1;
    if (++B->bar_cnt[s] == B->nfib) { B->bar_complete[s] = g; B->bar_cnt[s] = 0; }
    while (B->bar_complete[s] != g) yield_next("workgroup barrier");
}

static void trampoline()
{
    const int f = B->cur;
    B->body(B->arg);
    B->done[f] = true;
    for (int t = 1; t <= B->nfib; ++t) {        // hand over to the next unfinished fiber, or back to main
        int nx = (f + t) % B->nfib;
        if (!B->done[nx]) { B->cur = nx; setcontext(&B->ctx[nx]); }
    }
    setcontext(&B->main_ctx);
}

void run_block(void (*body)(void*), void* arg, int nwaves)
{
    static BlockState st;
    B = &st;
    const size_t STK = 256 * 1024;
    if (st.stacks.empty()) st.stacks.resize(64 * MAXW * STK);
    st.body = body; st.arg = arg; st.nfib = 64 * nwaves;
    for (int w = 0; w < MAXW; ++w) {
        st.ws[w].cnt[0] = st.ws[w].cnt[1] = 0;
        st.ws[w].complete[0] = 0xFFFFFFFFu; st.ws[w].complete[1] = 0xFFFFFFFEu;
    }
    st.bar_cnt[0] = st.bar_cnt[1] = 0;
    st.bar_complete[0] = 0xFFFFFFFFu; st.bar_complete[1] = 0xFFFFFFFEu;
    for (int i = 0; i < st.nfib; ++i) {
        st.gen[i] = 0; st.bar_gen[i] = 0; st.done[i] = false;
        getcontext(&st.ctx[i]);
        st.ctx[i].uc_stack.ss_sp = st.stacks.data() + i * STK;
        st.ctx[i].uc_stack.ss_size = STK;
        st.ctx[i].uc_link = &st.main_ctx;
        makecontext(&st.ctx[i], (void (*)())trampoline, 0);
    }
    st.cur = 0;
    swapcontext(&st.main_ctx, &st.ctx[0]);
    for (int i = 0; i < st.nfib; ++i)
        if (!st.done[i]) { fprintf(stderr, "emu: fiber %d did not finish\n", i); abort(); }
}

void run_wave(void (*body)(void*), void* arg) { run_block(body, arg, 1); }
}}  // namespace ck::emu

#include "../../circkit_amd/csrc/canon_core.h"
#include "../../circkit_amd/csrc/canon_fast.h"
#include "../../circkit_amd/csrc/canon_stream.h"
#include "../../circkit_amd/csrc/canon_mixed.h"
#include "../../circkit_amd/csrc/xxh3_core.h"

namespace {
struct Launch { ck::CanonArgs a; uint32_t* lds; const uint32_t* lut; const uint32_t* lutn = nullptr; const uint32_t* lean_lutn = nullptr; uint32_t* blk_count; uint32_t block, nblocks, wib; bool all_records = false, alpha = false, solo = true; const uint32_t* htab = nullptr; };
void mixed_body(void* p)         // one fiber of a 4-wave workgroup of canon_mixed[_n][_h]_kernel (NM = the batch's MODE_ALPHA, HASH = out_hash given)
{
    Launch* L = (Launch*)p;
    const uint32_t wib = ck::emu::cur_wave();
    const uint64_t payload_end = L->a.offsets[L->a.n_records];
    uint32_t* slice = L->lds + wib * L->a.slice_dw;
    if (L->a.out_hash) {
        ck::RescueState<true, false> st;
        st.hc.k0 = ck::xsec64(8 * (ck::lane_id() >> 2) + 16 * (ck::lane_id() & 3));
        st.hc.k1 = ck::xsec64(8 * (ck::lane_id() >> 2) + 16 * (ck::lane_id() & 3) + 8);
        if (L->alpha) ck::canon_mixed_segment<true, true>(L->a, slice, L->lut, st, L->blk_count, L->block, wib, 4, payload_end, L->htab, L->lean_lutn);
        else ck::canon_mixed_segment<false, true>(L->a, slice, L->lut, st, L->blk_count, L->block, wib, 4, payload_end, L->htab, L->lean_lutn);
    } else {
        ck::RescueState<false, false> st;
        if (L->alpha) ck::canon_mixed_segment<true, false>(L->a, slice, L->lut, st, L->blk_count, L->block, wib, 4, payload_end, nullptr, L->lean_lutn);
        else ck::canon_mixed_segment<false, false>(L->a, slice, L->lut, st, L->blk_count, L->block, wib, 4, payload_end);
    }
}
void wave_body(void* p)          // one fiber of a 4-wave workgroup of the LDS tier (canon_kernel<4>)
{
    Launch* L = (Launch*)p;
    const uint32_t wib = ck::emu::cur_wave();
    ck::canon_wave_loop(L->a, L->lds + wib * L->a.slice_dw, L->lut, L->blk_count, L->block, L->nblocks, wib, 4, L->lutn);
    ck::block_barrier();
    ck::team_pass(L->a, L->lds, L->lut, L->lutn, L->blk_count, L->block, wib, 4, L->solo);
}
void rescue_body(void* p)
{
    Launch* L = (Launch*)p;
    const bool aux = L->a.out_index || L->a.out_strand || (L->a.flags & ck::CK_FLAG_FWD_ONLY);
    if (aux) { ck::RescueState<true, true> st; st.hc = ck::fast_hash_const(); ck::canon_rescue_segment<true, true, false>(L->a, L->lut, st, L->blk_count, L->block, L->wib, 4, L->all_records); }
    else if (L->a.out_hash) { ck::RescueState<true, false> st; st.hc = ck::fast_hash_const(); if (L->alpha) ck::canon_rescue_segment<true, false, true>(L->a, L->lut, st, L->blk_count, L->block, L->wib, 4, L->all_records); else ck::canon_rescue_segment<true, false, false>(L->a, L->lut, st, L->blk_count, L->block, L->wib, 4, L->all_records); }
    else { ck::RescueState<false, false> st; if (L->alpha) ck::canon_rescue_segment<false, false, true>(L->a, L->lut, st, L->blk_count, L->block, L->wib, 4, L->all_records); else ck::canon_rescue_segment<false, false, false>(L->a, L->lut, st, L->blk_count, L->block, L->wib, 4, L->all_records); }
}
template <class C>
void stream_body(void* p)       // one fiber of a C::WPB-wave workgroup
{
    Launch* L = (Launch*)p;
    // same choice of build as launch_canon()
    const bool aux = L->a.out_index || L->a.out_strand || (L->a.flags & ck::CK_FLAG_FWD_ONLY);
    if (aux) ck::canon_stream_wave_loop<C, true, true>(L->a, L->lut, L->lds, L->blk_count, L->block, L->nblocks);
    else if (L->a.out_hash) {
        // the one-record-per-wave, one-word builds finish XXH3 per record group (same condition as canon_stream_kernel: groups of
        // up to 16 records since round 4)
        // ...; two records per wave without the 4-bit routine: canon_pair.h, one record per half-wave)
        constexpr bool GH = C::ROWS == 1 && C::RPW == 1 && C::GROUP <= 16, GHP = C::ROWS == 1 && C::GROUP <= 16;
        if (L->alpha && C::ROWS == 1) ck::canon_stream_wave_loop<C, true, false, GH, C::ROWS == 1>(L->a, L->lut, L->lds, L->blk_count, L->block, L->nblocks, L->lds + C::LDS_DW);
        else ck::canon_stream_wave_loop<C, true, false, GHP>(L->a, L->lut, L->lds, L->blk_count, L->block, L->nblocks, L->lds + C::LDS_DW);
    }
    else if (L->alpha && C::ROWS == 1) ck::canon_stream_wave_loop<C, false, false, false, C::ROWS == 1>(L->a, L->lut, L->lds, L->blk_count, L->block, L->nblocks);
    else ck::canon_stream_wave_loop<C, false, false>(L->a, L->lut, L->lds, L->blk_count, L->block, L->nblocks, L->lds + C::LDS_DW);     // (two records per wave: the pair build's scratch)
}
// geometries the staged streaming kernel is exercised with (index = `staged` argument - 1)
struct StreamVariant { void (*body)(void*); int wpb; uint32_t group, lds_dw; };
template <class C> constexpr StreamVariant variant() { return StreamVariant{ stream_body<C>, C::WPB, C::GROUP, C::LDS_DW }; }
const StreamVariant kStream[] = {
    variant<ck::StreamCfg<16, 2, 1>>(), variant<ck::StreamCfg<8, 3>>(), variant<ck::StreamCfg<4, 2>>(),
    variant<ck::StreamCfg<8, 2>>(), variant<ck::StreamCfg<2, 4>>(), variant<ck::StreamCfg<1, 3>>(),
    variant<ck::StreamCfg<4, 4, 1>>(), variant<ck::StreamCfg<8, 6, 1>>(), variant<ck::StreamCfg<4, 4>>(),
    // ROWS = 2: records of up to 2032 bases, two packed words per lane
    variant<ck::StreamCfg<16, 2, 1, 2>>(), variant<ck::StreamCfg<4, 2, 2, 2>>(), variant<ck::StreamCfg<8, 3, 1, 2>>(),
    // the product's geometry of the builds with the fused XXH3 (round 4): 8 waves, one record each, two images
    variant<ck::StreamCfg<8, 2, 1>>(),
    // ... and of the pair build: 4 waves, two records each (one per half-wave), groups of 8
    variant<ck::StreamCfg<4, 2, 2>>(), variant<ck::StreamCfg<8, 2, 2>>(),
    // the bytes-only N build's: 4 waves, one record each, two images
    variant<ck::StreamCfg<4, 2, 1>>(),
    // the two-word build's: 8 waves, one record each, two images of 2 KiB per record
    variant<ck::StreamCfg<8, 2, 1, 2>>(),
};
}

namespace {
struct HashLaunch { const uint8_t* p; uint32_t n; uint64_t out[64]; const uint8_t* comp = nullptr; uint32_t view = 0; };
void hash_body(void* q)
{
    HashLaunch* H = (HashLaunch*)q;
    // comp given: the canonical record as a view of the INPUT record (hash-only batches, xxh3_kernel's view branch)
    H->out[ck::lane_id()] = H->comp ? ck::xxh3_64_wave_view(H->p, H->n, H->view, H->comp, ck::xwave_const()) : ck::xxh3_64_wave(H->p, H->n);
}
}

// Same launch sequence as the host library: streaming kernel over everything (workgroups of 4 waves, each with
// its own deferral segment), then one LDS tier over the segmented list.  n_waves is rounded up to whole workgroups.
extern "C" int emu_canonicalize_batch(const uint8_t* bytes, const uint64_t* offsets, uint64_t n_records,
                                      uint8_t* out_bytes, uint32_t* out_index, uint8_t* out_strand,
                                      uint64_t* out_hash, uint32_t slice_dw, uint32_t n_waves,
                                      uint32_t* n_deferred, uint32_t flags, uint32_t* n_fast, uint32_t* n_fused_hash, int staged, uint32_t* n_rescued, int alpha)
{
    uint8_t comp[256];
    for (int v = 0; v < 256; ++v) comp[v] = (uint8_t)v;
    const char *x = "AGCTYRWSKMDVHBN", *y = "TCGARYWSMKHBDVN";
    for (int i = 0; x[i]; ++i) { comp[(uint8_t)x[i]] = y[i]; comp[(uint8_t)x[i] + 32] = y[i] + 32; }
    const uint32_t G = (n_waves + 3) / 4;
    // staged == 0: the batch's mode is 3 (launch_canon: too many long records to stage anything) -- no streaming
    // kernel, the rescue pass takes every record
    if (staged < 0 || staged > (int)(sizeof(kStream) / sizeof(kStream[0]))) return -1;
    const StreamVariant* sv = &kStream[staged ? staged - 1 : 0];
    const bool all_records = staged == 0;
    const uint64_t per_step = sv->group, steps = (n_records + per_step - 1) / per_step;
    const uint32_t cap = (uint32_t)(per_step * ((steps + G - 1) / G)) + 4;
    std::vector<uint32_t> lds((sv->lds_dw > slice_dw * 4 ? sv->lds_dw : slice_dw * 4) + 1024 + 16 + ck::gh_lds_dw<16, true>()), list_f((size_t)G * cap), list_a((size_t)G * cap);
    for (uint32_t tid = 0; tid < ck::GH_SECRET_DW; ++tid) ck::group_hash_secret_init(lds.data() + sv->lds_dw + 2 * sv->group * ck::GH_STRIDE_DW + ck::GH_CONST_DW, tid);
    for (uint32_t tid = 0; tid < 4; ++tid) ck::group_hash_init(lds.data() + sv->lds_dw + 2 * sv->group * ck::GH_STRIDE_DW, tid);       // (canon_stream_kernel: behind the 2 x GROUP slots)
    std::vector<uint32_t> cnt_f(G, 0), cnt_a(G, 0);
    uint32_t status = 0, lut[ck::FAST_LUT_DW];
    ck::fast_lut_init(lut, 0, 1);
    uint32_t lutn[ck::FAST_LUTN_DW];
    ck::fast_lutn_init(lutn, 0, 1);
    Launch L;
    L.a = ck::CanonArgs{};
    L.a.bytes = bytes; L.a.offsets = offsets; L.a.n_records = n_records;
    L.a.out_bytes = out_bytes; L.a.out_index = out_index; L.a.out_strand = out_strand; L.a.out_hash = out_hash;
    std::vector<uint8_t> hashed(n_records + 1, 0);
    L.a.hashed = hashed.data();
    std::vector<uint32_t> view(n_records + 1, 0xDEADBEEFu);       // launch_canon: a hash-only batch (no out_bytes) leaves views
    L.a.out_view = out_hash && !out_bytes ? view.data() : nullptr;
    L.a.status = &status; L.a.comp_lut = comp; L.a.flags = flags;
    L.a.defer_list = list_f.data(); L.a.defer_count = cnt_f.data(); L.a.out_seg_cap = cap;
    L.lds = lds.data(); L.lut = lut; L.lutn = lutn; L.nblocks = G;
    L.alpha = (alpha & 1) != 0;  // which builds of the streaming kernel and the rescue pass (launch_canon: MODE_ALPHA of the batch's mode)
    L.solo = (alpha & 2) == 0;   // bit 1: the tier's team pass without its wave-0-alone fallback (tests that pin a mode)
    uint32_t total_f = 0, total_a = 0;
    for (uint32_t b = 0; b < G && !all_records; ++b) {
        uint32_t blk = 0;
        L.block = b; L.blk_count = &blk;
        ck::emu::run_block(sv->body, &L, sv->wpb);
        cnt_f[b] = blk; total_f += blk;
    }
    if (all_records) total_f = (uint32_t)n_records;
    L.all_records = all_records;
    L.alpha = (alpha & 1) != 0;  // which build of the rescue pass (launch_canon: MODE_ALPHA of the batch's mode)
    if (n_fast) *n_fast = (uint32_t)n_records - total_f;
    // rescue pass: the streaming kernel's leftovers that are eligible by themselves (same build choice as launch_canon)
    std::vector<uint32_t> list_r((size_t)G * cap), cnt_r(G, 0);
    uint32_t total_r = 0;
    L.a.list = list_f.data(); L.a.list_count = cnt_f.data(); L.a.in_nseg = G; L.a.in_seg_cap = cap; L.a.segs_per_block = 1; L.a.all_seg_cap = cap;
    L.a.defer_list = list_r.data(); L.a.defer_count = cnt_r.data(); L.a.out_seg_cap = cap;
    // launch_canon: a mode-3 batch that wants bytes only is canon_mixed_kernel's (bit 2 of `alpha` selects it here)
    const bool mixed = all_records && (alpha & 4) && !out_index && !out_strand && !(flags & ck::CK_FLAG_FWD_ONLY);
    uint32_t htab[ck::LEAN_HASH_TABLE_DW];
    for (uint32_t tid = 0; tid < 4; ++tid) ck::lean_hash_table_init(htab, tid);
    L.htab = htab;
    uint32_t lean_lutn[ck::LEAN_LUTN_DW];
    ck::lean_lutn_init(lean_lutn, 0, 1);
    L.lean_lutn = lean_lutn;
    for (uint32_t b = 0; b < G; ++b) {
        uint32_t blk = 0;
        L.block = b; L.blk_count = &blk;
        if (mixed) { L.a.slice_dw = slice_dw; ck::emu::run_block(mixed_body, &L, 4); }
        else for (uint32_t w = 0; w < 4; ++w) { L.wib = w; ck::emu::run_wave(rescue_body, &L); }
        cnt_r[b] = blk; total_r += blk;
    }
    if (n_rescued) *n_rescued = total_f - total_r;
    L.a.list = list_r.data(); L.a.list_count = cnt_r.data(); L.a.in_nseg = G; L.a.in_seg_cap = cap; L.a.segs_per_block = 1;
    L.a.defer_list = list_a.data(); L.a.defer_count = cnt_a.data(); L.a.out_seg_cap = cap; L.a.slice_dw = slice_dw;
    for (uint32_t b = 0; b < G; ++b) {
        uint32_t blk[4] = { 0, 0, 0, 0 };           // deferral counter + the team's three words
        L.block = b; L.blk_count = blk;
        ck::emu::run_block(wave_body, &L, 4);
        cnt_a[b] = blk[0]; total_a += blk[0];
    }
    if (n_deferred) *n_deferred = total_a;
    if (out_hash) {     // the xxh3 pass for whatever the streaming kernel did not hash
        for (uint64_t r = 0; r < n_records; ++r) {
            if (hashed[r]) continue;
            HashLaunch H{ (out_bytes ? out_bytes : bytes) + offsets[r], (uint32_t)(offsets[r + 1] - offsets[r]), {0} };
            if (!out_bytes) { H.comp = comp; H.view = view[r]; }
            ck::emu::run_wave(hash_body, &H);
            out_hash[r] = H.out[0];
        }
    }
    if (n_fused_hash) { *n_fused_hash = 0; for (uint64_t r = 0; r < n_records; ++r) *n_fused_hash += hashed[r]; }
    return (int)status;
}

extern "C" uint64_t emu_xxh3_64(const uint8_t* p, uint32_t n)
{
    HashLaunch H{ p, n, {0} };
    ck::emu::run_wave(hash_body, &H);
    for (int i = 1; i < 64; ++i)
        if (H.out[i] != H.out[0]) { fprintf(stderr, "emu: xxh3 lanes disagree\n"); abort(); }
    return H.out[0];
}

// seg_records (canon_core.h): the segments of a walking stage must tile [0, n) exactly once, in order.  Returns the number of
// segments that hold records, or -1 at the first gap / overlap / oversized segment.
extern "C" int64_t emu_seg_cover(uint64_t n, uint32_t nseg, uint32_t all_cap, uint32_t taper_seg0, uint32_t taper_log2)
{
    ck::CanonArgs a{};
    a.n_records = n; a.all_seg_cap = all_cap; a.taper_seg0 = taper_seg0; a.taper_log2 = taper_log2;
    uint64_t next = 0;
    int64_t used = 0;
    for (uint32_t s = 0; s < nseg; ++s) {
        uint64_t first; uint32_t count;
        ck::seg_records(a, s, first, count);
        if (count > all_cap) return -1;
        if (count == 0) continue;
        if (first != next) return -1;
        next = first + count;
        ++used;
    }
    return next == n ? used : -1;
}
