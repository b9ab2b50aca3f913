// tc_encode_host.hpp -- host orchestration of the encode path (device pointers in,
// device pointers out).  Included by textcomp.hip only.
#pragma once
#include <chrono>
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "tc_mtf.hpp"
#include "tc_radix_host.hpp"
#include "tc_rle.hpp"
#include "tc_pack.hpp"
#include "tc_sa.hpp"
#include "tc_msd.hpp"
#include "tc_seg.hpp"
#include "tc_chain.hpp"

// ---------------------------------------------------------------- small helpers
static inline void tc_memset_async(tc_ctx *ctx, void *p, int v, size_t bytes) {
    TC_HIP(ctx, hipMemsetAsync(p, v, bytes, ctx->stream));
}
static inline void tc_d2h(tc_ctx *ctx, void *dst, const void *src, size_t bytes) {
    TC_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
}
static inline void tc_h2d(tc_ctx *ctx, void *dst, const void *src, size_t bytes) {
    TC_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
}
static inline int ceil_log2_u64(u64 v) {  // bits needed to hold values < v
    int b = 0;
    while (b < 63 && (1ull << b) < v) b++;
    return b;
}
static inline int env_int(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}

// Alphabet of a Seq (Maybe Word8): counts257[0] = #Nothing, [1+b] = #byte b.
// nubSeq' (reference MTF/Internal.hs:79-99): present symbols, sorted, Nothing first.
struct Alphabet {
    u32 sigma = 0;
    i16 sym_of_code[TC_MAX_SIGMA];
    u16 code_of_sym[TC_MAX_SIGMA];  // index = sym + 1
    void build(const u32 *counts257) {
        sigma = 0;
        for (int v = 0; v < 257; v++) {
            code_of_sym[v] = 0;
            if (counts257[v]) {
                code_of_sym[v] = (u16)sigma;
                sym_of_code[sigma++] = (i16)(v - 1);
            }
        }
    }
};

// ------------------------------------------------------------------ suffix array
#ifndef SA_KDIR_BITS
#define SA_KDIR_BITS 26
#endif
struct SaBuffers {
    u64 *k0, *k1;
    u32 *v0, *v1;
    u32 *isa;
    u32 *v2;
    u32 *act[2][4];  // [set][slot, idx, grp, tpos]
    // sparse mode (few tied suffixes): round buffers + rank table, each `sparse_cap` long
    u64 *sk[2];
    u32 *sv[2];
    u32 *t_idx, *t_rank;
    u64 *t_bits;   // N bits
    u32 *t_dir;    // per 64-bit word of t_bits
    u32 *t_bsum;
    u32 *fin_rc;   // finish_kernel: region counters + region offsets
    u32 *kdir;     // 2^SA_KDIR_BITS + 1
    u64 sparse_cap;
    u32 *hist;
    u64 *rstatus;
    u64 *gstatus;  // 2*tiles + 2
    u32 *counts;   // 256 byte counts
    // MSD round 0 (tc_msd.hpp), carved only for texts long enough to take it
    u32 *msd_pstart[MSD_LEVELS + 1], *msd_pcnt[MSD_LEVELS + 1];   // [l]: parents of level l + 1; [3]: level-3 buckets
    u32 *msd_tpre[MSD_LEVELS], *msd_seg[MSD_LEVELS];
    u32 *msd_joint;   // [256^3] child counts of the level-3 parents, gathered by the level-2 counting pass
    u32 msd_grid;
    TiedTable tp;     // key-only levels: hash table of the tied keys (tc_sa.hpp)
    SegBuffers seg;   // segmented sort of the doubling rounds (tc_seg.hpp)
    // chain rounds (tc_chain.hpp): reference rank per group head slot, code per text position, block summaries of the scan
    u32 *chain_ref, *chain_code, *chain_summ;
};

// the MSD round 0 pays from this many suffixes on (level-3 buckets of >= ~64 members on DNA)
static inline u64 msd_min_n() { return (u64)env_int("TC_SA_MSD_MIN_LOG2", 27) >= 40 ? ~0ull : 1ull << env_int("TC_SA_MSD_MIN_LOG2", 27); }
static inline bool msd_wanted(u64 N) { return env_int("TC_SA_MSD", 1) != 0 && N >= msd_min_n() && N > 4 * MSD_TILE; }

static size_t sa_carve(Arena &A, u64 N, SaBuffers &b, bool own_v1) {
    b.k0 = A.get<u64>(N);
    b.k1 = A.get<u64>(N);
    b.v0 = A.get<u32>(N);
    b.v1 = own_v1 ? A.get<u32>(N) : nullptr;
    b.isa = A.get<u32>(N + 1);
    b.v2 = A.get<u32>(N);
    b.sparse_cap = N / 8 + 1024;
    for (int s = 0; s < 2; s++) {
        for (int q = 0; q < 3; q++) b.act[s][q] = A.get<u32>(N);
        b.act[s][3] = A.get<u32>(b.sparse_cap);
        b.sk[s] = A.get<u64>(b.sparse_cap);
        b.sv[s] = A.get<u32>(b.sparse_cap);
    }
    b.t_idx = A.get<u32>(b.sparse_cap);
    b.t_rank = A.get<u32>(b.sparse_cap);
    b.t_bits = A.get<u64>(N / 64 + 2);
    b.t_dir = A.get<u32>(N / 64 + 2);
    b.t_bsum = A.get<u32>(N / 64 / BDIR_TILE + 2);
    b.fin_rc = A.get<u32>(FIN_REGIONS * FIN_RSTRIDE + 128);
    b.kdir = A.get<u32>(((size_t)1 << SA_KDIR_BITS) + 2 + ((size_t)1 << SA_KDIR_BITS) / KDF_CHUNK + 64);   // directory + block minima of its fill
    b.hist = A.get<u32>(RDX_MAX_PASSES * RDX_BINS);
    b.rstatus = A.get<u64>(radix_status_words(N));
    b.gstatus = A.get<u64>(2 * (size_t)tc_cdiv(N, GRP_TILE) + 4);
    b.counts = A.get<u32>(260);
    {
        SegBuffers &g = b.seg;
        g.cap_runs = (size_t)(N / SEG_CAP + 2);
        g.cap_tiles = (size_t)(N / SEG_PT + 2) + g.cap_runs;
        g.segbits = A.get<u64>(seg_bit_words(N));
        g.ybits = A.get<u64>(seg_bit_words(N));
        for (int q = 0; q < 2; q++) {
            g.lstart[q] = A.get<u32>(g.cap_runs);
            g.lsize[q] = A.get<u32>(g.cap_runs);
            g.ltbase[q] = A.get<u32>(g.cap_runs);
            g.lshift[q] = A.get<u32>(g.cap_runs);
        }
        g.tile_seg = A.get<u32>(g.cap_tiles);
        g.hist = A.get<u32>(g.cap_runs * 256);
        g.mm = A.get<u32>(g.cap_runs * 2);
        g.counters = A.get<u32>(64);
    }
    for (int l = 0; l <= MSD_LEVELS; l++) b.msd_pstart[l] = b.msd_pcnt[l] = nullptr;
    if (msd_wanted(N)) {
        b.msd_grid = 256 * MSD_BPC;   // fixed for the carve; the launch uses min(this, CUs x workgroups per CU)
        size_t np = 1;
        for (int l = 0; l <= MSD_LEVELS; l++, np *= 256) {
            b.msd_pstart[l] = A.get<u32>(np);
            b.msd_pcnt[l] = A.get<u32>(np);
            if (l < MSD_LEVELS) {
                b.msd_tpre[l] = A.get<u32>(np + 1);
                b.msd_seg[l] = A.get<u32>((np + b.msd_grid) * 256);
            }
        }
        b.msd_joint = A.get<u32>((size_t)256 * 256 * 256);
        b.tp.key = A.get<u64>((size_t)1 << TP_SLOT_BITS);
        b.tp.grp = A.get<u32>((size_t)1 << TP_SLOT_BITS);
        b.tp.cnt = A.get<u32>((size_t)1 << TP_SLOT_BITS);
        b.tp.bloom = A.get<u32>(((size_t)1 << TP_BLOOM_LOG2) / 32);
    }
    // (carved last: everything above keeps the offsets it had before the chain rounds existed)
    b.chain_ref = A.get<u32>(N + 1);
    b.chain_code = A.get<u32>(N + 1);
    b.chain_summ = A.get<u32>(chain_summ_words());
    return A.off;
}

// The sort of one doubling round (tc_seg.hpp): keys (grp << 32 | rank, grp non-decreasing) and values, m members, ranks
// below 2^rbits.  The result is in (kx, vx); (ky, vy) is scratch of the same size.  One host synchronisation per
// partition level that has long runs (none: one, for the count of long runs).
static void seg_sort_pairs(tc_ctx *ctx, SegBuffers &g, u64 *kx, u32 *vx, u64 *ky, u32 *vy, u32 m, int rbits) {
    hipStream_t s = ctx->stream;
    const u32 nwords = (u32)seg_bit_words(m);
    TC_HIP(ctx, hipMemsetAsync(g.ybits, 0, (size_t)nwords * sizeof(u64), s));
    TC_HIP(ctx, hipMemsetAsync(g.counters, 0, 8 * sizeof(u32), s));
    u32 igrid = tc_cdiv((u64)nwords * 64, 256);
    if (igrid > 16384) igrid = 16384;
    seg_init_kernel<<<igrid, 256, 0, s>>>(kx, m, g.segbits, nwords, g.lstart[0], g.lsize[0], g.ltbase[0], g.lshift[0],
                                          (u32)(rbits > 8 ? rbits - 8 : 0), g.counters, (u32)g.cap_runs);
    TC_LAUNCH_CHECK(ctx);
    // (a level either splits a run by 8 more rank bits or -- all members in one digit -- re-lists it with a better shift:
    // at most 4 of the first kind and 4 of the second per run)
    const int nlev = 8;
    int cur = 0;
    for (int L = 0;; L++) {
        TC_HIP(ctx, hipMemcpyAsync(&ctx->h_scalars[24], g.counters + 2 * cur, 2 * sizeof(u32), hipMemcpyDeviceToHost, s));
        TC_HIP(ctx, hipStreamSynchronize(s));
        const u32 S = (u32)(ctx->h_scalars[24] & 0xffffffffu), T = (u32)(ctx->h_scalars[24] >> 32);
        if (S == 0) break;
        if (L >= nlev) TC_FAIL(ctx, TC_ERR_INTERNAL, "segmented sort: %u runs still unsorted after %d levels", S, nlev);
        if (S > g.cap_runs || T > g.cap_tiles) TC_FAIL(ctx, TC_ERR_INTERNAL, "segmented sort: %u long runs / %u tiles exceed the tables", S, T);
        const int nxt = cur ^ 1;
        TC_HIP(ctx, hipMemsetAsync(g.hist, 0, (size_t)S * 256 * sizeof(u32), s));
        seg_mm_init_kernel<<<tc_cdiv(S, 256), 256, 0, s>>>(g.mm, S);
        TC_LAUNCH_CHECK(ctx);
        TC_HIP(ctx, hipMemsetAsync(g.counters + 2 * nxt, 0, 2 * sizeof(u32), s));
        u32 wgrid = tc_cdiv(S, 4);
        if (wgrid > 8192) wgrid = 8192;
        seg_tilemap_kernel<<<wgrid, 256, 0, s>>>(g.lsize[cur], g.ltbase[cur], S, g.tile_seg, (u32)g.cap_tiles);
        TC_LAUNCH_CHECK(ctx);
        seg_count_kernel<<<T, 256, 0, s>>>(kx, ky, g.lstart[cur], g.lsize[cur], g.ltbase[cur], g.lshift[cur], g.tile_seg, g.counters + 2 * cur, g.hist, g.mm);
        TC_LAUNCH_CHECK(ctx);
        seg_scan_kernel<<<wgrid, 256, 0, s>>>(g.lstart[cur], g.lsize[cur], g.lshift[cur], g.counters + 2 * cur, g.hist, g.mm, g.segbits, 0,
                                              g.lstart[nxt], g.lsize[nxt], g.ltbase[nxt], g.lshift[nxt], g.counters + 2 * nxt, (u32)g.cap_runs);
        TC_LAUNCH_CHECK(ctx);
        seg_scatter_kernel<<<T, 256, 0, s>>>(kx, vx, ky, vy, g.lstart[cur], g.lsize[cur], g.ltbase[cur], g.lshift[cur], g.tile_seg, g.counters + 2 * cur,
                                             g.hist, g.mm, g.ybits);
        TC_LAUNCH_CHECK(ctx);
        cur = nxt;
    }
    seg_small_kernel<<<tc_cdiv(m, SEG_SPAN), SEG_NT, 0, s>>>(kx, vx, ky, vy, m, g.segbits, g.ybits);
    TC_LAUNCH_CHECK(ctx);
#ifdef SEG_PROFILE
    {   // cycles per phase of thread 0, per window (diagnostic build only)
        u64 h[16];
        TC_HIP(ctx, hipStreamSynchronize(s));
        TC_HIP(ctx, hipMemcpyFromSymbol(h, HIP_SYMBOL(seg_prof), sizeof h));
        const double w = (double)(h[7] | 1);
        fprintf(stderr, "seg_small: %llu members, %llu windows (that sort), mid members per window %.0f | cycles per window: bits %.0f attr+masks %.0f (prefix) %.0f image %.0f tiny+compact %.0f tiny store %.0f network %.0f store %.0f\n",
                (unsigned long long)m, (unsigned long long)h[7], h[8] / w, h[0] / w, h[1] / w, 0.0, h[2] / w, h[3] / w, h[4] / w, h[5] / w, h[6] / w);
        memset(h, 0, sizeof h);
        TC_HIP(ctx, hipMemcpyToSymbol(HIP_SYMBOL(seg_prof), h, sizeof h));
    }
#endif
}

static void sa_choose_config(tc_ctx *ctx, const u32 *counts, u64 n, SaConfig &c) {
    (void)ctx;
    u32 sig = 0;
    double H = 0;
    for (int v = 0; v < 256; v++) {
        c.lut[v] = 0;
        if (counts[v]) {
            c.lut[v] = (u16)(++sig);
            double p = (double)counts[v] / (double)n;
            H -= p * log2(p);
        }
    }
    c.sigma_text = sig;
    c.B = sig + 1;
    if (c.B <= 16) {
        c.w = 8;
        c.s = 1;
        u32 pw = c.B;
        while (pw * c.B <= 256) {
            pw *= c.B;
            c.s++;
        }
    } else {
        c.s = 1;
        c.w = (u32)ceil_log2_u64(c.B);
    }
    u32 pmax = 56 / c.w;  // the low 8 key bits carry the preceding text byte
    // fields so that an iid text of this entropy has ~2^-8 of its suffixes still tied
    double need = (double)ceil_log2_u64(n + 1) + 8.0;
    double per_field = H * c.s;
    u32 P = pmax;
    if (per_field > 1e-9) {
        double pf = ceil(need / per_field);
        if (pf < (double)pmax) P = (u32)pf;
    }
    if (P < 1) P = 1;
    int forced = env_int("TC_SA_FIELDS", 0);
    if (forced > 0) P = (u32)forced;
    if (P > pmax) P = pmax;
    c.P = P;
    c.h0 = P * c.s;
    c.entropy = H;
}

// Builds SA (d_sa, N entries), last column (d_L, N bytes) and primary for the
// device text.  d_sa may be null (workspace buffer used).  counts256_out (host,
// optional) receives the byte histogram.
static void sa_run(tc_ctx *ctx, SaBuffers &b, const u8 *d_text, u64 n, u32 *d_sa, u8 *d_L,
                   u64 *primary, u32 *counts256_out);

static void sa_build(tc_ctx *ctx, Arena &A, const u8 *d_text, u64 n, u32 *d_sa, u8 *d_L,
                     u64 *primary, u32 *counts256_out, bool dry) {
    const u64 N = n + 1;
    SaBuffers b;
    sa_carve(A, N, b, d_sa == nullptr);
    if (dry) return;
    sa_run(ctx, b, d_text, n, d_sa, d_L, primary, counts256_out);
    // The sharded tile tickets of the radix pass assume blocks start in roughly increasing
    // blockIdx order; if a bounded look-back spin tripped, redo with the single counter.
    if (!ctx->safe_tickets) {
        TC_HIP(ctx, hipMemcpyAsync(&ctx->h_scalars[62], ctx->d_err, sizeof(u32), hipMemcpyDeviceToHost,
                                   ctx->stream));
        TC_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if ((u32)ctx->h_scalars[62] & 2u) {
            tc_memset_async(ctx, ctx->d_err, 0, sizeof(u32));
            ctx->safe_tickets = 1;
            ctx->ticket_fallbacks++;
            sa_run(ctx, b, d_text, n, d_sa, d_L, primary, counts256_out);
        }
    }
    ctx->stats.ticket_fallbacks = ctx->ticket_fallbacks;
}

static void sa_run(tc_ctx *ctx, SaBuffers &b, const u8 *d_text, u64 n, u32 *d_sa, u8 *d_L,
                   u64 *primary, u32 *counts256_out) {
    const u64 N = n + 1;
    hipStream_t s = ctx->stream;
    tc_stats &st = ctx->stats;

    if (env_int("TC_SA_TRACE", 0) == 2)
        fprintf(stderr, "textcomp: buffers text %p k0 %p k1 %p v0 %p v1 %p sa %p L %p\n", (const void *)d_text, (void *)b.k0,
                (void *)b.k1, (void *)b.v0, (void *)b.v1, (void *)d_sa, (void *)d_L);
    // 1. alphabet
    tc_memset_async(ctx, b.counts, 0, 256 * sizeof(u32));
    {
        u32 grid = tc_cdiv(n, 256 * 64);
        if (grid > 2048) grid = 2048;
        if (grid < 1) grid = 1;
        hist256_kernel<<<grid, 256, 0, s>>>(d_text, n, b.counts);
        TC_LAUNCH_CHECK(ctx);
    }
    u32 counts[256];
    tc_d2h(ctx, counts, b.counts, sizeof counts);
    TC_HIP(ctx, hipStreamSynchronize(s));
    if (counts256_out) memcpy(counts256_out, counts, sizeof counts);
    SaConfig cfg;
    sa_choose_config(ctx, counts, n, cfg);
    if (cfg.sigma_text == 1 && n > 0) {
        // unary text (a zero-filled buffer, "AAAA..."): the suffixes are ordered by length, no
        // sort needed (prefix doubling would take log2 n full rounds here)
        u32 *sa_out = d_sa ? d_sa : b.v1;
        unary_sa_kernel<<<tc_cdiv(N, 256), 256, 0, s>>>(d_text, (u32)n, sa_out, d_L);
        TC_LAUNCH_CHECK(ctx);
        *primary = n;
        st.sigma = 2; st.rounds = 0; st.radix_launches = 0; st.ms_radix = 0;
        return;
    }

    // 2. round-0 keys (+ digit histograms), radix sort, groups.
    //    Fast path: sort only the top 8*G key bits globally, then finish_kernel orders the
    //    (tiny, on high-entropy text) equal-prefix buckets by the remaining bits and emits
    //    SA / L / the tied set.  Oversize buckets or a large tied set => the full path:
    //    all P passes, group_kernel<INIT>, dense ISA if needed.
    u32 *va = d_sa ? d_sa : b.v1;
    const u32 gtiles = tc_cdiv(N, GRP_TILE);
    auto run_group = [&](bool init, GroupArgs ga, u32 *sa_arr) {
        u32 tiles = tc_cdiv(ga.count, GRP_TILE);
        tc_memset_async(ctx, b.gstatus, 0, (2 * (size_t)gtiles + 4) * sizeof(u64));
        ga.text = d_text; ga.sa = sa_arr; ga.L = d_L;
        ga.status_max = b.gstatus; ga.status_sum = b.gstatus + gtiles;
        ga.ticket = reinterpret_cast<u32 *>(b.gstatus + 2 * (size_t)gtiles);
        ga.scalars = ctx->d_scalars; ga.err = ctx->d_err;
        u32 grid = init ? tc_persistent_grid_for(ctx, group_kernel<true>, GRP_NT, 2)
                        : tc_persistent_grid_for(ctx, group_kernel<false>, GRP_NT, 2);
        if (grid > tiles) grid = tiles;
        if (init) group_kernel<true><<<grid, GRP_NT, 0, s>>>(ga);
        else group_kernel<false><<<grid, GRP_NT, 0, s>>>(ga);
        TC_LAUNCH_CHECK(ctx);
    };
    auto fetch_m = [&]() {
        tc_d2h(ctx, ctx->h_scalars, ctx->d_scalars, 2 * sizeof(u64));
        TC_HIP(ctx, hipStreamSynchronize(s));
        return ctx->h_scalars[1];
    };
    // dense ranks of a large set go by regions (tc_sa.hpp, "dense ranks by regions"): group_kernel leaves
    // (start, rank) pairs in `pairs`, these two kernels store them.  `part`: N free u64 slots.
    const int rbits0 = ceil_log2_u64(N);
    const int rshift = rbits0 > 8 ? rbits0 - 8 : 0;
    const u64 bin_min = 1ull << env_int("TC_SA_BIN_MIN_LOG2", 25);   // (below ~2^25 members the direct stores are as fast)
    auto apply_pairs = [&](const u64 *pairs, u32 count, u64 *part) {
        u32 *cursor = b.hist;   // (free between the radix passes of two rounds)
        rank_cursor_kernel<<<1, 256, 0, s>>>(cursor, rshift);
        TC_LAUNCH_CHECK(ctx);
        rank_bin_kernel<<<tc_cdiv(count, RBIN_TILE), RBIN_NT, 0, s>>>(pairs, count, rshift, cursor, part, N);
        TC_LAUNCH_CHECK(ctx);
        rank_scatter_kernel<<<tc_cdiv(N, RSCAT_NT * RSCAT_ITEMS), RSCAT_NT, 0, s>>>(part, N, rshift, cursor, b.isa);
        TC_LAUNCH_CHECK(ctx);
    };
    // (the second active set is free whenever the first one is being built: its slot + idx arrays are
    // adjacent in the arena and together hold N u64)
    u64 *part_act1 = reinterpret_cast<u64 *>(b.act[1][0]);
    const bool part_act1_ok = (size_t)((char *)b.act[1][2] - (char *)b.act[1][0]) >= N * sizeof(u64) &&
                              ((uintptr_t)b.act[1][0] & 7) == 0;
    auto build_keys_and_sort = [&](const RadixPlan &plan, RadixBuffers &rb, bool sa_in_alt_at_end) {
        KeyBuildParams kp;
        kp.B = cfg.B; kp.w = cfg.w; kp.s = cfg.s; kp.P = cfg.P;
        memcpy(kp.lut, cfg.lut, sizeof kp.lut);
        kp.plan.npass = plan.npass;
        for (int p = 0; p < plan.npass; p++) {
            kp.plan.shift[p] = plan.shift[p];
            kp.plan.mask[p] = plan.mask[p];
        }
        tc_memset_async(ctx, b.hist, 0, sizeof(u32) * RDX_MAX_PASSES * RDX_BINS);
        // one shared histogram when every pass is exactly one 8-bit field
        bool onehist = cfg.w == 8 && plan.npass < RDX_MAX_PASSES && env_int("TC_KB_ONEHIST", 1) != 0;
        for (int p = 0; p < plan.npass; p++)
            if (plan.mask[p] != 255u || plan.shift[p] % 8 != 0) onehist = false;
        const u32 kgrid = tc_cdiv(N, SA_TILE);
        // fused first pass: keys are generated inside the first radix pass (no key array
        // written + re-read); needs the shared-histogram configuration
        const bool fuse = onehist && RDX_TILE == SA_TILE && env_int("TC_KEYGEN_FUSED", 1) != 0;
        st.keygen_fused = fuse ? 1u : 0u;
        RadixKeyGen kg;
        kg.hash_ok = 0; kg.hsh = 0; kg.tlo = 0; kg.thi = 0;
        if (fuse) {
            kg.n_text = (u32)n; kg.B = cfg.B; kg.w = cfg.w; kg.s = cfg.s; kg.P = cfg.P;
            memcpy(kg.lut, cfg.lut, sizeof kg.lut);
            u32 ggrid = tc_cdiv(N, 256 * 16 * 4);
            if (ggrid > 2048) ggrid = 2048;
            if (cfg.s == 3) ghist_kernel<3><<<ggrid, 256, 0, s>>>(d_text, (u32)n, kp, b.hist);
            else ghist_kernel<0><<<ggrid, 256, 0, s>>>(d_text, (u32)n, kp, b.hist);
            TC_LAUNCH_CHECK(ctx);
            keyhist_fix_kernel<<<1, 256, 0, s>>>(d_text, (u32)n, kp, b.hist);
        } else if (onehist) {
            // unrolled instances for the DNA-like configuration (3 symbols per field)
            if (cfg.s == 3 && cfg.P == 6) keybuild_kernel<true, 3, 6><<<kgrid, SA_NT, 0, s>>>(d_text, (u32)n, kp, b.k0, b.hist);
            else if (cfg.s == 3 && cfg.P == 5) keybuild_kernel<true, 3, 5><<<kgrid, SA_NT, 0, s>>>(d_text, (u32)n, kp, b.k0, b.hist);
            else keybuild_kernel<true, 0, 0><<<kgrid, SA_NT, 0, s>>>(d_text, (u32)n, kp, b.k0, b.hist);
            TC_LAUNCH_CHECK(ctx);
            keyhist_fix_kernel<<<1, 256, 0, s>>>(d_text, (u32)n, kp, b.hist);
        } else {
            keybuild_kernel<false, 0, 0><<<kgrid, SA_NT, 0, s>>>(d_text, (u32)n, kp, b.k0, b.hist);
        }
        TC_LAUNCH_CHECK(ctx);
        // ping-pong arranged so that the sorted values land in `va` (or, when a finish
        // pass follows, in the OTHER buffer so that the finish pass writes `va`)
        rb.keys = b.k0; rb.keys_alt = b.k1;
        const bool even = plan.npass % 2 == 0;
        const bool start_in_va = sa_in_alt_at_end ? !even : even;
        if (start_in_va) { rb.vals = va; rb.vals_alt = b.v0; }
        else { rb.vals = b.v0; rb.vals_alt = va; }
        rb.hist = b.hist;
        rb.status = b.rstatus; rb.status_cap = radix_status_words(N);
        ctx->pev_used = 0;
        radix_sort_pairs(ctx, rb, (u32)N, plan, /*gen_idx=*/true, /*hist_ready=*/true, /*timed=*/true,
                         d_text, fuse ? &kg : nullptr,
                         /*xcd_group=*/!ctx->safe_tickets && env_int("TC_XCD_GROUP", 1) != 0);
    };
    const int rbits = ceil_log2_u64(N);
    int keybits = (int)(cfg.P * cfg.w);
    const u32 P_full = cfg.P;   // fields chosen for the full path (every field is a pass there)
    u32 *sa = va;
    const u64 *skeys = nullptr;
    const u64 *tkeys = nullptr;
    int tkeys_shift = 0;
    u64 m = 0;
    u32 fm_dropped = 0;
    bool hopeless = false, isa_ready = false;
    bool many_ties = false;   // the LSD way's finish pass drowned in ties: the full path that follows will want dense ranks
    u64 h_start = cfg.h0;
    bool have_groups = false;
    tc_memset_async(ctx, ctx->d_scalars, 0, 16 * sizeof(u64));

    if (env_int("TC_SA_FINISH", 1) != 0 && env_int("TC_SA_DENSE", 0) == 0) {
        // global passes: enough top bits that an iid text of this entropy leaves ~4 suffixes
        // per bucket, and few enough remaining bits for the finish pass (<= 32)
        double e8 = cfg.w == 8 ? cfg.entropy * cfg.s : cfg.entropy * 8.0 / cfg.w;
        int G = e8 > 1e-9 ? (int)ceil((log2((double)N) - 4.0) / e8) : 64;  // ~16 suffixes per bucket at most
        if (G < (keybits - 32 + 7) / 8) G = (keybits - 32 + 7) / 8;
        if (G < 1) G = 1;
        int forcedG = env_int("TC_SA_GLOBAL_PASSES", 0);
        if (forcedG > 0) G = forcedG;
        // a candidate for the MSD way sorts by 7 fields: its LSD fallback then needs >= 3 global passes
        const bool msd_cand = cfg.w == 8 && b.msd_pstart[0] != nullptr && msd_wanted(N) &&
                              env_int("TC_SA_FIELDS", 0) == 0 && forcedG == 0;
        if (msd_cand && G < MSD_LEVELS) G = MSD_LEVELS;
        int topbits = 8 * G < keybits ? 8 * G : keybits;
        // cheap look before the leap: if a sample of suffixes already collides heavily on the
        // globally sorted prefix, the tied set would exceed the sparse capacity anyway
        if (keybits - topbits <= 32 && n >= (1u << 20) && env_int("TC_SA_SAMPLE", 1) != 0) {
            RadixKeyGen kgs;
            kgs.n_text = (u32)n; kgs.B = cfg.B; kgs.w = cfg.w; kgs.s = cfg.s; kgs.P = cfg.P;
            memcpy(kgs.lut, cfg.lut, sizeof kgs.lut);
            u32 *d_dups = reinterpret_cast<u32 *>(ctx->d_scalars + 14);
            sample_dup_kernel<<<1, 1024, 0, s>>>(d_text, (u32)n, kgs, topbits, d_dups);
            TC_LAUNCH_CHECK(ctx);
            tc_d2h(ctx, &ctx->h_scalars[14], ctx->d_scalars + 14, sizeof(u64));
            TC_HIP(ctx, hipStreamSynchronize(s));
            st.sample_dups = (u32)ctx->h_scalars[14];
            hopeless = st.sample_dups > SAMP_N / 10;
        }
        if (keybits - topbits <= 32 && !hopeless) {
            // fields beyond the globally sorted ones cost no pass here (the finish pass ranks by all
            // remaining bits at once), so take as many as fit: fewer suffixes stay tied
            if (env_int("TC_SA_FIELDS", 0) == 0) {
                u32 pf = (u32)((topbits + 32) / (int)cfg.w);
                if (pf > 56 / cfg.w) pf = 56 / cfg.w;
                if (pf > cfg.P) {
                    cfg.P = pf;
                    cfg.h0 = cfg.P * cfg.s;
                    keybits = (int)(cfg.P * cfg.w);
                    h_start = cfg.h0;
                }
            }
            // Round 0, two ways.  MSD (tc_msd.hpp; long texts over a small alphabet): three partition
            // levels by field 0, 1, 2 with whole-line stores, then every level-3 bucket ordered in LDS.
            // LSD (tc_radix.hpp): the top fields by stable passes, then finish_kernel.  Both hand over
            // SA / last column for the untied suffixes and the tied set in act[1] (64 regions).  A text
            // whose level-3 buckets are too long for the MSD finish falls through to the LSD way.
            u32 *counters = reinterpret_cast<u32 *>(ctx->d_scalars + 12);
            // The MSD way is for texts that look iid at the depth of its levels: (i) the entropy estimate
            // puts a level-3 bucket well under the finish kernel's chunk, (ii) the sample met next to no
            // repeated 12-symbol prefix (repeat-rich DNA has dozens among 8192; iid text of this length
            // none) -- otherwise the attempt would be paid for and then thrown away.
            const double lvl_bits = (cfg.entropy * cfg.s < 8.0 ? cfg.entropy * cfg.s : 8.0) * MSD_LEVELS;
            // expected level-3 bucket: a third of a small chunk (5-letter DNA at 1 GiB) -> the small finish
            // instance; up to ~5/8 of a big chunk (4-letter DNA at 1 GiB: 4096) -> the big one, which also
            // writes the keys in final order (rank lookups by binary search: its buckets are too long to scan)
            const double msd_bucket = (double)N / exp2(lvl_bits);
            const bool msd_big = msd_bucket > (double)MSDF_CAP_SMALL / 3.0 || env_int("TC_SA_MSD_BIG", 0) != 0;
            const bool msd_fits = msd_bucket <= (double)MSDF_CAP_BIG * 0.8;
            // (what an iid text of this entropy leaves among SAMP_N samples at the sampled depth, with slack)
            const double iid_dups = (double)SAMP_N * SAMP_N / 2.0 / exp2((cfg.entropy * cfg.s < 8.0 ? cfg.entropy * cfg.s : 8.0) * (topbits / 8));
            const bool msd_iid = (double)st.sample_dups <= (double)env_int("TC_SA_MSD_MAX_DUPS", 8) + 3.0 * iid_dups;
            // (the big instance copes with repeats -- over-long buckets leave as tied groups, ranks of untied
            // suffixes come by binary search in its sorted keys -- but repeat-rich DNA is slower this way
            // than by the LSD way, whose finish orders 14+ symbols instead of 12: 1 GiB genome-like
            // 188 ms against 116 ms.  So the sample decides for both instances.)
            // (round 4 tried to send repeat-rich DNA this way as well -- the big instance's whole buckets go through the KEY
            // ROUND below and come out tied on all 21 symbols -- but on such text the levels and the big finish themselves are
            // slow: 8.1 instead of 5.9 ms per level and 32 instead of 7 ms for the finish at 1 GiB (one workgroup per level-3
            // parent: the parents of the repeat family are the tail), 112 ms against 99 by the LSD way.  TC_SA_MSD=3: that
            // experiment; the key round itself stays for the whole buckets an iid-looking text still has.)
            const bool keyround = env_int("TC_SA_KEYROUND", 1) != 0;
            const bool try_msd = msd_cand && cfg.P == 7 && (env_int("TC_SA_MSD", 1) == 2 ||
                                                            (msd_fits && (msd_iid || (msd_big && keyround && env_int("TC_SA_MSD", 1) == 3))));
            // no suffix array asked for (encode, BWT): the levels can move keys only (tc_msd.hpp, VALS = false).  Both
            // finish instances; the big one's over-long buckets (whole tied groups: msd_whole_kernel works from the
            // suffix starts) send the text through the levels again with the starts moving along
            // -- so it is only tried when few ties are expected: an iid text of this entropy leaves about
            // N^2 / 2^(entropy x key symbols) suffixes equal on the whole key (1 GiB: 5-letter DNA 2 300, measured
            // 2 404; 4-letter DNA 262 000, measured 261 586 -- more than the table of tied keys is made for)
            const double tied_est = (double)N * (double)N / exp2(cfg.entropy * (double)(cfg.P * cfg.s));
            bool keyonly = d_sa == nullptr && (tied_est < (double)TP_MAX_TIED / 4.0 || env_int("TC_SA_MSD_KEYONLY", 1) == 2) &&
                           env_int("TC_SA_MSD_KEYONLY", 1) != 0;
            for (int way = try_msd ? 0 : 1; way < 2 && !have_groups; way++) {
            const bool msd = way == 0;
            const int tb = msd ? 8 * MSD_LEVELS : topbits;   // key bits that are globally ordered
            RadixPlan plan;
            RadixBuffers rb;
            FinishArgs fa;
            fa.N = (u32)N; fa.tshift = 64 - tb;
            fa.lshift = 64 - keybits; fa.lbits = keybits - tb;
            fa.sa_out = va; fa.L = d_L;
            // the lean pass appends to 64 regions of the SECOND active set (one counter each); they are
            // then packed into the first one, which everything below works on
            fa.out_slot = b.act[1][0]; fa.out_idx = b.act[1][1]; fa.out_grp = b.act[1][2];
            fa.act_cap = (u32)N; fa.counters = counters;
            fa.rcount = b.fin_rc; fa.rcap = (u32)(N / FIN_REGIONS);
            fa.fix_cap = (u32)(b.sparse_cap - 1024);
            fa.ovbits = b.act[1][0];   // (after the packing) the second active set is unused again
            u32 *roff = b.fin_rc + FIN_REGIONS * FIN_RSTRIDE;
            tc_memset_async(ctx, ctx->d_scalars + 12, 0, 2 * sizeof(u64));
            tc_memset_async(ctx, b.fin_rc, 0, (FIN_REGIONS * FIN_RSTRIDE + 128) * sizeof(u32));
            int npass_stat = 0;
            const u64 *kbuf_sorted = nullptr;
            if (msd) {
                RadixKeyGen kg;
                kg.n_text = (u32)n; kg.B = cfg.B; kg.w = cfg.w; kg.s = cfg.s; kg.P = cfg.P;
                memcpy(kg.lut, cfg.lut, sizeof kg.lut);
                radix_keygen_hash(kg);
                if (env_int("TC_KEYGEN_HASH", 1) == 0) kg.hash_ok = 0;
                MsdTextDigit td;
                td.text = d_text; td.n = (u32)n; td.B = cfg.B; td.s = cfg.s;
                memcpy(td.lut, cfg.lut, sizeof td.lut);
                td.hash_ok = kg.hash_ok; td.hsh = kg.hsh; td.tlo = kg.tlo; td.thi = kg.thi;
                u32 G = b.msd_grid < (u32)ctx->num_cus * MSD_BPC ? b.msd_grid : (u32)ctx->num_cus * MSD_BPC;
                if (ctx->reserved_cus > 0 && G > (u32)(ctx->num_cus - ctx->reserved_cus) * MSD_BPC)
                    G = (u32)(ctx->num_cus - ctx->reserved_cus) * MSD_BPC;   // (CUs left to the exchange: tc_comm_create)
                if (env_int("TC_MSD_GRID", 0) > 0 && (u32)env_int("TC_MSD_GRID", 0) < G) G = (u32)env_int("TC_MSD_GRID", 0);
                u32 *maxchild = reinterpret_cast<u32 *>(ctx->d_scalars + 15);
                msd_root_kernel<<<1, 1, 0, s>>>(b.msd_pstart[0], b.msd_pcnt[0], (u32)N, maxchild);
                TC_LAUNCH_CHECK(ctx);
                // level 1 writes (k0, v0); level 2 (k1, va); level 3 (k0, v0); the finish reads (k0, v0)
                // and writes va / L
                u64 *kbuf[2] = {b.k0, b.k1};
                u32 *vbuf[2] = {keyonly ? nullptr : b.v0, keyonly ? nullptr : va};
                st.msd_keyonly = keyonly ? 1u : 0u;
                ctx->pev_used = 0;
                st.keygen_fused = 1;
                // the last level is "aligned" (one workgroup per parent): its child counts are gathered by the
                // level before it, which saves that level's counting pass over the keys (TC_SA_MSD_JOINT=0: off)
                // (its LDS table has a row per digit made of real symbols only: sigma^s <= 128 of them)
                MsdJointRows jr;
                u32 nrows = 0;
                {
                    memset(jr.row, 0xff, sizeof jr.row);
                    memset(jr.dig, 0, sizeof jr.dig);
                    u32 nd = 1;
                    for (u32 j = 0; j < cfg.s; j++) nd *= cfg.B;
                    for (u32 d = 0; d < nd && d < 256; d++) {
                        bool real = true;
                        for (u32 v = d, j = 0; j < cfg.s; j++, v /= cfg.B) real = real && (v % cfg.B) != 0;
                        if (real) {
                            if (nrows < 128) { jr.row[d] = (u8)nrows; jr.dig[nrows] = (u8)d; }
                            nrows++;
                        }
                    }
                }
                const bool joint = env_int("TC_SA_MSD_JOINT", 1) != 0 && nrows <= 128;
                if (joint) tc_memset_async(ctx, b.msd_joint, 0, (size_t)256 * 256 * 256 * sizeof(u32));
                u32 np = 1;
                for (int l = 0; l < MSD_LEVELS; l++, np *= 256) {
                    MsdLevel ML;
                    ML.pstart = b.msd_pstart[l]; ML.pcnt = b.msd_pcnt[l]; ML.tpre = b.msd_tpre[l];
                    ML.nparents = np; ML.shift = 56 - 8 * l; ML.seg = b.msd_seg[l];
                    ML.cstart = b.msd_pstart[l + 1]; ML.ccnt = b.msd_pcnt[l + 1];
                    ML.aligned = (joint && l == MSD_LEVELS - 1) ? 1 : 0;
                    ML.ntot = (u32)N; ML.cnt_in = b.msd_joint; ML.flags = counters + 1;
                    ML.dbg = ctx->d_scalars + 64 + 16 * l;
                    const u64 *kin = l ? kbuf[(l - 1) & 1] : nullptr;
                    const u32 *vin = l ? vbuf[(l - 1) & 1] : nullptr;
                    msd_prep_kernel<<<1, 1024, 0, s>>>(ML.pcnt, np, b.msd_tpre[l]);
                    TC_LAUNCH_CHECK(ctx);
                    if (l == 0) msd_count_kernel<true, false><<<G, MSD_NT, 0, s>>>(ML, nullptr, td, nullptr, jr);
                    else if (ML.aligned) { /* counts already in msd_joint */ }
                    else if (joint && l == MSD_LEVELS - 2) msd_count_kernel<false, true><<<G, MSD_NT, 0, s>>>(ML, kin, td, b.msd_joint, jr);
                    else msd_count_kernel<false, false><<<G, MSD_NT, 0, s>>>(ML, kin, td, nullptr, jr);
                    TC_LAUNCH_CHECK(ctx);
                    msd_scan_kernel<<<np, 256, 0, s>>>(ML, G, l == MSD_LEVELS - 1 ? maxchild : nullptr);
                    TC_LAUNCH_CHECK(ctx);
                    const bool ev = ctx->profile && ctx->pev_used < 16;
                    if (ev) TC_HIP(ctx, hipEventRecord(ctx->pev[2 * ctx->pev_used], s));
                    if (keyonly) {
                        if (l == 0) msd_partition_kernel<true, false><<<G, MSD_NT, 0, s>>>(ML, nullptr, nullptr, kbuf[0], nullptr, d_text, kg);
                        else msd_partition_kernel<false, false><<<G, MSD_NT, 0, s>>>(ML, kin, nullptr, kbuf[l & 1], nullptr, d_text, kg);
                    } else if (l == 0) msd_partition_kernel<true><<<G, MSD_NT, 0, s>>>(ML, nullptr, nullptr, kbuf[0], vbuf[0], d_text, kg);
                    else msd_partition_kernel<false><<<G, MSD_NT, 0, s>>>(ML, kin, vin, kbuf[l & 1], vbuf[l & 1], d_text, kg);
                    TC_LAUNCH_CHECK(ctx);
                    if (ev) {
                        TC_HIP(ctx, hipEventRecord(ctx->pev[2 * ctx->pev_used + 1], s));
                        ctx->pev_used++;
                    }
                }
                rb.keys = kbuf[(MSD_LEVELS - 1) & 1];
                rb.vals = vbuf[(MSD_LEVELS - 1) & 1];
                MsdFinishArgs mf;
                mf.keys = rb.keys; mf.vals = rb.vals;
                mf.pcnt = b.msd_pcnt[MSD_LEVELS - 1];
                mf.cstart = b.msd_pstart[MSD_LEVELS]; mf.ccnt = b.msd_pcnt[MSD_LEVELS];
                mf.sa_out = va; mf.L = d_L;
                mf.out_slot = fa.out_slot; mf.out_idx = fa.out_idx; mf.out_grp = fa.out_grp;
                mf.rcount = fa.rcount; mf.rcap = fa.rcap; mf.counters = counters;
                mf.kout = kbuf[MSD_LEVELS & 1];   // (the key buffer the last level did not write)
                kbuf_sorted = mf.kout;
                // (the last level's segment table is dead by now: the list of over-long buckets goes there)
                mf.whole_list = b.msd_seg[MSD_LEVELS - 1];
                mf.whole_cap = 1u << 20;
                mf.out_khi = b.v0;   // (key-only: the value buffers are free; region layout as out_idx)
                // (equal-mass bins from the level-1 digit counts: tc_msd.hpp; TC_MSD_FINISH_LUT=0: the generic instances bin by key bits)
                MsdFinishLut *flut = reinterpret_cast<MsdFinishLut *>(b.msd_seg[MSD_LEVELS - 1] + 2 * (size_t)mf.whole_cap);
                msd_finish_lut_kernel<<<1, 256, 0, s>>>(b.msd_pcnt[1], flut);
                TC_LAUNCH_CHECK(ctx);
                mf.lut = env_int("TC_MSD_FINISH_LUT", 1) != 0 ? flut : nullptr;
                if (msd_big && keyonly) {
                    msd_finish_kernel<MSDF_BIG_NT, MSDF_BIG_ITEMS, 1, 5, true, false><<<np / 256, MSDF_BIG_NT, 0, s>>>(mf);
                    // (no msd_whole_kernel: listed buckets raise bit 1 of the flags, which ends the key-only attempt below)
                } else if (msd_big) {
                    msd_finish_kernel<MSDF_BIG_NT, MSDF_BIG_ITEMS, 1, 5, true><<<np / 256, MSDF_BIG_NT, 0, s>>>(mf);
                    TC_LAUNCH_CHECK(ctx);
                    msd_whole_kernel<<<1024, MSDW_NT, 0, s>>>(mf);
                } else if (keyonly && env_int("TC_MSD_FINISH_KO", 1) != 0) {
                    msd_finish_ko_kernel<3><<<np / 256, 256, 0, s>>>(mf, flut);
#ifdef MSDK_PROFILE
                    {
                        u64 h[10];
                        tc_d2h(ctx, h, ctx->d_scalars + 112, sizeof h);
                        TC_HIP(ctx, hipStreamSynchronize(s));
                        const double c = (double)(h[8] | 1);
                        fprintf(stderr, "finish (key-only): chunks %llu, keys per chunk %.0f | cycles per chunk: land %.0f B %.0f zero+B %.0f bins+B %.0f scan+B %.0f scatter+B %.0f walk+prefetch+rank+B %.0f copy-out %.0f\n",
                                (unsigned long long)h[8], h[9] / c, h[0] / c, h[1] / c, h[2] / c, h[3] / c, h[4] / c, h[5] / c, h[6] / c, h[7] / c);
                        tc_memset_async(ctx, ctx->d_scalars + 112, 0, sizeof h);
                    }
#endif
                } else if (keyonly) {
                    msd_finish_kernel<MSDF_KO_NT, MSDF_CAP_SMALL / MSDF_KO_NT, 4, 1, false, false><<<np / 256, MSDF_KO_NT, 0, s>>>(mf);
                } else {
                    msd_finish_kernel<256, 8, 4, 1, false><<<np / 256, 256, 0, s>>>(mf);
                }
                TC_LAUNCH_CHECK(ctx);
                npass_stat = MSD_LEVELS;
                st.msd_path = 1;
#ifdef MSD_PROFILE
                {   // cycles per phase of workgroup 0 / thread 0, per level (diagnostic build only)
                    u64 h[48];
                    tc_d2h(ctx, h, ctx->d_scalars + 64, sizeof h);
                    TC_HIP(ctx, hipStreamSynchronize(s));
                    for (int l = 0; l < MSD_LEVELS; l++)
                        fprintf(stderr, "msd level %d: tiles %llu | cursor+B0 %llu keygen/S1+B1 %llu S2 %llu S3 %llu land %llu B3 %llu S4 %llu (cycles per tile)\n", l + 1,
                                (unsigned long long)h[16 * l + 1], (unsigned long long)(h[16 * l] / (h[16 * l + 1] | 1)), (unsigned long long)(h[16 * l + 2] / (h[16 * l + 1] | 1)),
                                (unsigned long long)(h[16 * l + 3] / (h[16 * l + 1] | 1)), (unsigned long long)(h[16 * l + 4] / (h[16 * l + 1] | 1)), (unsigned long long)(h[16 * l + 5] / (h[16 * l + 1] | 1)),
                                (unsigned long long)(h[16 * l + 6] / (h[16 * l + 1] | 1)), (unsigned long long)(h[16 * l + 7] / (h[16 * l + 1] | 1)));
                    for (int l = 0; l < MSD_LEVELS; l++)
                        fprintf(stderr, "   level %d, S1 alone per tile: wave 0 %llu, last wave %llu cycles; slowest wave B0 -> before B1 %llu, B0 -> prefetch issued %llu\n", l + 1,
                                (unsigned long long)(h[16 * l + 8] / (h[16 * l + 1] | 1)), (unsigned long long)(h[16 * l + 9] / (h[16 * l + 1] | 1)),
                                (unsigned long long)(h[16 * l + 10] / (h[16 * l + 1] | 1)), (unsigned long long)(h[16 * l + 11] / (h[16 * l + 1] | 1)));
                    tc_memset_async(ctx, ctx->d_scalars + 64, 0, sizeof h);
                }
#endif
            } else {
            st.msd_path = 0;
            plan.add_range(64 - topbits, 64);
            build_keys_and_sort(plan, rb, /*sa_in_alt_at_end=*/true);
            npass_stat = plan.npass;
            fa.keys = rb.keys; fa.sa_in = rb.vals;
            const u32 waves = tc_cdiv(N, 64 * FIN_WPW);
            finish_kernel<<<tc_cdiv(waves, FIN_NT / 64), FIN_NT, 0, s>>>(fa);
            TC_LAUNCH_CHECK(ctx);
            }
            finish_regions_kernel<<<1, 64, 0, s>>>(fa.rcount, fa.rcap, roff, counters);
            TC_LAUNCH_CHECK(ctx);
            if (msd && keyonly)
                finish_compact_kernel<<<1024, 256, 0, s>>>(roff, fa.rcap, b.act[1][0], b.act[1][1], b.act[1][2],
                                                          b.act[0][0], b.act[0][1], b.act[0][2], b.v0, b.act[0][3], (u32)b.sparse_cap);
            else
                finish_compact_kernel<<<1024, 256, 0, s>>>(roff, fa.rcap, b.act[1][0], b.act[1][1], b.act[1][2],
                                                          b.act[0][0], b.act[0][1], b.act[0][2]);
            TC_LAUNCH_CHECK(ctx);
            fa.out_slot = b.act[0][0]; fa.out_idx = b.act[0][1]; fa.out_grp = b.act[0][2];
            tc_d2h(ctx, &ctx->h_scalars[12], ctx->d_scalars + 12, sizeof(u64));
            TC_HIP(ctx, hipStreamSynchronize(s));
            u32 fm = (u32)(ctx->h_scalars[12] & 0xffffffffu), over = (u32)(ctx->h_scalars[12] >> 32);
            if (env_int("TC_SA_TRACE", 0))
                fprintf(stderr, "textcomp: round 0 %s way%s: tied %u, flags 0x%x (1 over-long bucket left to the fix pass, 2 whole buckets tied, 4 bucket above the finish chunk, 8 joint counts off)\n",
                        msd ? "MSD" : "LSD", msd && msd_big ? " (big finish)" : "", fm, over);
            if (msd && (over & (4u | 8u))) continue;   // a level-3 bucket beyond the finish chunk (or counts that overflowed): the LSD way
            if (msd && keyonly) {
                // the tied members are known by slot, group and key: find their suffix starts again (tc_sa.hpp).  More
                // ties than the table is made for, or a pass that does not find exactly fm positions: the levels
                // are run once more with the suffix starts moving along.
                bool ok = fm <= TP_MAX_TIED && (u64)fm + 1024 <= b.sparse_cap && !(over & 2u);
                if (env_int("TC_SA_TRACE", 0) >= 2 && ok && fm > 0) {   // (order-free sums of the tied members' keys, slots and groups)
                    std::vector<u32> hk(fm), hh(fm), hs(fm), hg(fm);
                    tc_d2h(ctx, hk.data(), b.act[0][1], fm * sizeof(u32));
                    tc_d2h(ctx, hh.data(), b.act[0][3], fm * sizeof(u32));
                    tc_d2h(ctx, hs.data(), b.act[0][0], fm * sizeof(u32));
                    tc_d2h(ctx, hg.data(), b.act[0][2], fm * sizeof(u32));
                    TC_HIP(ctx, hipStreamSynchronize(s));
                    u64 a1 = 0, a2 = 0, a3 = 0, a4 = 0;
                    for (u32 i = 0; i < fm; i++) { a1 += hk[i]; a2 += hh[i]; a3 += hs[i]; a4 += hg[i]; }
                    fprintf(stderr, "textcomp: tied members: sum klo %llx khi %llx slot %llx grp %llx\n", (unsigned long long)a1, (unsigned long long)a2, (unsigned long long)a3, (unsigned long long)a4);
                    for (u32 i = 0; i < fm && i < 6; i++) fprintf(stderr, "textcomp:   member %u: klo %08x khi %06x slot %u grp %u\n", i, hk[i], hh[i], hs[i], hg[i]);
                }
                if (ok && fm > 0) {
                    tc_memset_async(ctx, b.tp.key, 0, sizeof(u64) << TP_SLOT_BITS);
                    tc_memset_async(ctx, b.tp.cnt, 0, sizeof(u32) << TP_SLOT_BITS);
                    tc_memset_async(ctx, b.tp.bloom, 0, ((size_t)1 << TP_BLOOM_LOG2) / 8);
                    u32 *d_total = reinterpret_cast<u32 *>(ctx->d_scalars + 11);
                    tc_memset_async(ctx, d_total, 0, sizeof(u64));
                    tied_table_kernel<<<tc_cdiv(fm, 256), 256, 0, s>>>(b.act[0][1], b.act[0][3], b.act[0][2], fm, cfg.B, cfg.s, cfg.P, b.tp);
                    TC_LAUNCH_CHECK(ctx);
                    RadixKeyGen kgp;
                    kgp.hash_ok = 0; kgp.hsh = 0; kgp.tlo = 0; kgp.thi = 0;
                    kgp.n_text = (u32)n; kgp.B = cfg.B; kgp.w = cfg.w; kgp.s = cfg.s; kgp.P = cfg.P;
                    memcpy(kgp.lut, cfg.lut, sizeof kgp.lut);
                    u32 pgrid = (u32)ctx->num_cus * 3;
                    if (pgrid > tc_cdiv(n, TPK_TILE)) pgrid = tc_cdiv(n, TPK_TILE);
                    switch (cfg.s) {   // (symbols per field: B^s <= 256)
#define TC_PROBE(S) case S: tied_probe_kernel<S><<<pgrid, TPK_NT, 0, s>>>(d_text, (u32)n, kgp, b.tp, b.act[0][0], b.act[0][1], b.act[0][2], d_total, fm); break;
                        TC_PROBE(1) TC_PROBE(2) TC_PROBE(3) TC_PROBE(4) TC_PROBE(5) TC_PROBE(6) TC_PROBE(7) TC_PROBE(8)
#undef TC_PROBE
                        default: TC_FAIL(ctx, TC_ERR_INTERNAL, "key-only levels: %u symbols per field", cfg.s);
                    }
                    TC_LAUNCH_CHECK(ctx);
                    tc_d2h(ctx, &ctx->h_scalars[11], ctx->d_scalars + 11, sizeof(u64));
                    TC_HIP(ctx, hipStreamSynchronize(s));
                    ok = (u32)ctx->h_scalars[11] == fm;
                }
                if (env_int("TC_SA_TRACE", 0))
                    fprintf(stderr, "textcomp: key-only levels: %u tied suffixes %s (the pass over the text met %u)\n", fm,
                            ok ? "found again in the text" : "-- NOT recoverable: the levels run again with suffix starts", (u32)ctx->h_scalars[11]);
                if (!ok) {
                    keyonly = false;
                    way = -1;      // (the loop's increment makes it the MSD way again)
                    continue;
                }
            }
            u32 slot_bits = (u32)rbits;
            if (!msd && (over & 1u) && fm <= fa.fix_cap && env_int("TC_SA_TIER2", 1) != 0) {
                // some buckets are longer than a wave window: the second pass turns them into tied
                // groups (the sorted keys are still in place) and voids what the first pass
                // emitted for their members
                const u32 fm_lean = fm;
                u32 *flagword = reinterpret_cast<u32 *>(ctx->d_scalars + 12) + 1;
                u32 *ndropped = reinterpret_cast<u32 *>(ctx->d_scalars + 13);
                tc_memset_async(ctx, flagword, 0, sizeof(u32));
                tc_memset_async(ctx, ndropped, 0, sizeof(u64));
                tc_memset_async(ctx, fa.ovbits, 0, ((size_t)tc_cdiv(N, 64) + 1) * sizeof(u64));
                const u32 fwaves = tc_cdiv(N, 64 * FIX_WIN);
                finish_fix_kernel<<<tc_cdiv(fwaves, FIN_NT / 64), FIN_NT, 0, s>>>(fa);
                TC_LAUNCH_CHECK(ctx);
                if (fm_lean) {
                    finish_filter_kernel<<<tc_cdiv(fm_lean, 256) < 4096u ? tc_cdiv(fm_lean, 256) : 4096u, 256, 0, s>>>(fa.out_slot, fm_lean, fa.ovbits, ndropped);
                    TC_LAUNCH_CHECK(ctx);
                }
                tc_d2h(ctx, &ctx->h_scalars[12], ctx->d_scalars + 12, 2 * sizeof(u64));
                TC_HIP(ctx, hipStreamSynchronize(s));
                fm = (u32)(ctx->h_scalars[12] & 0xffffffffu);
                over = (u32)(ctx->h_scalars[12] >> 32);
                fm_dropped = (u32)(ctx->h_scalars[13] & 0xffffffffu);
                slot_bits = (u32)rbits + 1;   // the void slot value must sort behind slot N - 1
            }
            // (MSD: rank lookups of untied suffixes count inside an UNSORTED level-3 bucket, ~550 keys
            // each: fine for the few ties of an iid text, hopeless for millions -- the LSD way then)
            if (msd && !msd_big && fm > (1u << 18)) continue;
            if (!(over & 1u) && fm <= b.sparse_cap - 1024) {
                m = fm - fm_dropped;
                // whole buckets were emitted as tied groups: they share only the globally sorted
                // symbols, so the doubling starts from those
                if (over & 2u) h_start = (u64)(tb / (int)cfg.w) * cfg.s;
                have_groups = true;
                if (msd && msd_big) {   // keys in final order: ranks of untied suffixes by binary search
                    skeys = kbuf_sorted;
                } else {
                    tkeys = rb.keys;
                    tkeys_shift = 64 - tb;
                }
                st.finish_pass = 1;
                st.rounds = 1;
                st.m[0] = N; st.key_bytes[0] = 8; st.passes[0] = (u32)npass_stat; st.h[0] = 0;
                const bool tiny_set = fm <= SEG_W && env_int("TC_SA_TINY", 1) != 0;
                if (m > 0 && tiny_set) {   // (a few thousand members: one workgroup, tc_seg.hpp)
                    tied_small_kernel<0><<<1, SEG_NT, 0, s>>>(b.act[0][0], b.act[0][1], b.act[0][2], fm, nullptr, nullptr, nullptr);
                    TC_LAUNCH_CHECK(ctx);
                } else if (m > 0) {  // bring the tied set into SA order (refine relies on it); void entries go last
                    u32 mm = fm;
                    pack_active_kernel<<<tc_cdiv(mm, 256), 256, 0, s>>>(b.act[0][0], b.act[0][2], mm, b.sk[0]);
                    TC_LAUNCH_CHECK(ctx);
                    TC_HIP(ctx, hipMemcpyAsync(b.sv[0], b.act[0][1], mm * sizeof(u32), hipMemcpyDeviceToDevice, s));
                    RadixPlan ps;
                    ps.add_range(32, 32 + (int)slot_bits);
                    RadixBuffers rs;
                    rs.keys = b.sk[0]; rs.keys_alt = b.sk[1]; rs.vals = b.sv[0]; rs.vals_alt = b.sv[1];
                    rs.hist = b.hist; rs.status = b.rstatus; rs.status_cap = radix_status_words(N);
                    radix_sort_pairs(ctx, rs, mm, ps, false, false);
                    unpack_active_kernel<<<tc_cdiv(mm, 256), 256, 0, s>>>(rs.keys, mm, b.act[0][0], b.act[0][2]);
                    TC_LAUNCH_CHECK(ctx);
                    TC_HIP(ctx, hipMemcpyAsync(b.act[0][1], rs.vals, mm * sizeof(u32), hipMemcpyDeviceToDevice, s));
                }
                // the key round (tc_sa.hpp): whole buckets were emitted as groups that share 9 symbols; the key holds 12 more
                if (msd && msd_big && (over & 2u) && keyround && m >= (u64)env_int("TC_SA_SEG_MIN", 1 << 16) && env_int("TC_SA_SEG", 1) != 0) {
                    const u32 mm = (u32)m;
                    u64 *kall = const_cast<u64 *>(kbuf_sorted);
                    key_round_kernel<<<tc_cdiv(mm, 256), 256, 0, s>>>(b.act[0][0], b.act[0][2], kall, mm, b.sk[0], b.sv[0]);
                    TC_LAUNCH_CHECK(ctx);
                    seg_sort_pairs(ctx, b.seg, b.sk[0], b.sv[0], b.sk[1], b.sv[1], mm, 32);
                    key_round_store_kernel<<<tc_cdiv(mm, 256), 256, 0, s>>>(b.sk[0], b.act[0][0], mm, kall);
                    TC_LAUNCH_CHECK(ctx);
                    GroupArgs gk = {};
                    gk.keys = b.sk[0]; gk.count = mm; gk.vals = b.sv[0]; gk.vals_are_idx = 0; gk.norank = 1;
                    gk.in_slot = b.act[0][0]; gk.in_idx = b.act[0][1];
                    gk.out_slot = b.act[1][0]; gk.out_idx = b.act[1][1]; gk.out_grp = b.act[1][2];
                    run_group(false, gk, sa);
                    const u64 m2 = fetch_m();
                    for (int q = 0; q < 3; q++)
                        if (m2) TC_HIP(ctx, hipMemcpyAsync(b.act[0][q], b.act[1][q], m2 * sizeof(u32), hipMemcpyDeviceToDevice, s));
                    st.m[st.rounds] = m; st.key_bytes[st.rounds] = 8; st.passes[st.rounds] = 1; st.h[st.rounds] = (u32)h_start;
                    st.rounds++;
                    st.seg_rounds++;
                    if (env_int("TC_SA_TRACE", 0))
                        fprintf(stderr, "textcomp: key round: %u members of whole buckets ordered by the key's remaining 32 bits, %llu stay tied\n", mm, (unsigned long long)m2);
                    m = m2;
                    h_start = cfg.h0;   // every tie now shares the whole key
                }
            } else if (!msd && fm > b.sparse_cap - 1024) {
                many_ties = true;
            }
            }   // way
        }
    }
    if (!have_groups) {
        cfg.P = P_full;
        // the sample (or a finish pass that drowned in ties) says the entropy estimate behind P_full does
        // not hold -- natural language, runs: take every field the key has room for; one more pass of
        // the first sort, but the doubling starts deeper and usually saves a round (Zipf text, 256 MiB:
        // 84.6 -> 76.0 ms, five rounds -> four)
        if ((hopeless || st.finish_pass == 0) && env_int("TC_SA_FIELDS", 0) == 0 && env_int("TC_SA_DEEP", 1) != 0 &&
            env_int("TC_SA_FINISH", 1) != 0)
            cfg.P = 56 / cfg.w;
        cfg.h0 = cfg.P * cfg.s;
        keybits = (int)(cfg.P * cfg.w);
        h_start = cfg.h0;
        RadixPlan plan;
        plan.add_range(64 - keybits, 64);
        RadixBuffers rb;
        build_keys_and_sort(plan, rb, /*sa_in_alt_at_end=*/false);
        skeys = rb.keys;
        GroupArgs g0 = {};
        g0.keys = skeys; g0.count = (u32)N; g0.vals = sa;
        g0.out_slot = b.act[0][0]; g0.out_idx = b.act[0][1]; g0.out_grp = b.act[0][2]; g0.out_tpos = b.act[0][3];
        // many ties expected (the sample, or a finish pass that just met them -- a text of a period longer than the sample sees:
        // one group pass less, 11 ms per GiB): ranks in the same pass
        if (hopeless || many_ties) { g0.isa = b.isa; isa_ready = true; }
        const bool g0_pairs = (hopeless || many_ties) && N >= bin_min && part_act1_ok;
        if (g0_pairs) g0.pairs = rb.keys_alt;
        run_group(true, g0, sa);
        if (g0_pairs) apply_pairs(rb.keys_alt, (u32)N, part_act1);
        m = fetch_m();
        st.rounds = 1;
        st.m[0] = N; st.key_bytes[0] = 8; st.passes[0] = (u32)plan.npass; st.h[0] = 0;
    }

    // (TC_SA_TRACE: wall time per step, each closed by a stream sync -- experiments only)
    const bool trace_on = env_int("TC_SA_TRACE", 0) != 0;
    auto trace_t0 = std::chrono::steady_clock::now();
    auto trace = [&](const char *what, u64 count) {
        if (!trace_on) return;
        (void)hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "textcomp:   %-28s %10llu  %8.3f ms\n", what, (unsigned long long)count,
                std::chrono::duration<double, std::milli>(t1 - trace_t0).count());
        trace_t0 = t1;
    };
    trace("(sync before the rank table)", m);
    // 5. ranks: dense ISA when many suffixes are tied, else a sparse table of the tied
    //    positions + a search for everything else (sorted round-0 keys, or the SA itself)
    const bool dense = m > b.sparse_cap - 1024 || env_int("TC_SA_DENSE", 0) != 0;
    RankLookup rl = {};
    rl.text = d_text; rl.n = (u32)n; rl.N = (u32)N;
    rl.B = cfg.B; rl.w = cfg.w; rl.s = cfg.s; rl.P = cfg.P; rl.h0 = cfg.h0;
    memcpy(rl.lut, cfg.lut, sizeof rl.lut);
    if (dense && !skeys) TC_FAIL(ctx, TC_ERR_INTERNAL, "dense mode needs the sorted keys");
    if (dense) {
        if (!isa_ready) {  // (also with m == 0: the primary index is read from the ranks)
            GroupArgs gi = {};
            gi.keys = skeys; gi.count = (u32)N; gi.vals = sa;
            gi.isa = b.isa; gi.isa_only = 1;
            // (scratch: whichever round-0 key buffer does not hold the sorted keys; the second active set)
            u64 *kfree = skeys == b.k0 ? b.k1 : b.k0;
            const bool gi_pairs = N >= bin_min && part_act1_ok && (skeys == b.k0 || skeys == b.k1);
            if (gi_pairs) gi.pairs = kfree;
            run_group(true, gi, sa);
            if (gi_pairs) apply_pairs(kfree, (u32)N, part_act1);
        }
        rl.isa = b.isa;
    } else {
        rl.skeys = skeys; rl.tkeys = tkeys; rl.tshift = tkeys_shift; rl.sa = sa; rl.t_idx = b.t_idx; rl.t_rank = b.t_rank; rl.t_n = (u32)m;
        if (m > 0 && m <= SEG_W && env_int("TC_SA_TINY", 1) != 0) {
            tied_small_kernel<1><<<1, SEG_NT, 0, s>>>(b.act[0][0], b.act[0][1], b.act[0][2], (u32)m, b.t_idx, b.t_rank, b.act[0][3]);
            TC_LAUNCH_CHECK(ctx);
        } else if (m > 0) {
            u32 mm = (u32)m;
            widen_u32_kernel<<<tc_cdiv(mm, 256), 256, 0, s>>>(b.act[0][1], b.sk[0], mm);
            TC_LAUNCH_CHECK(ctx);
            RadixPlan pt;
            pt.add_range(0, rbits);
            RadixBuffers rt;
            rt.keys = b.sk[0]; rt.keys_alt = b.sk[1]; rt.vals = b.sv[0]; rt.vals_alt = b.sv[1];
            rt.hist = b.hist; rt.status = b.rstatus; rt.status_cap = radix_status_words(N);
            radix_sort_pairs(ctx, rt, mm, pt, true, false);
            // large tied sets: bitmap + popcount directory instead of a binary search per lookup,
            // and a directory into the sorted keys for the ranks of untied suffixes
            const bool accel = m >= (u64)env_int("TC_SA_ACCEL_MIN", 1 << 20);
            const u32 nwords = (u32)(N / 64 + 1);
            if (accel) tc_memset_async(ctx, b.t_bits, 0, (size_t)nwords * sizeof(u64));
            table_build_kernel<<<tc_cdiv(mm, 256), 256, 0, s>>>(rt.keys, rt.vals, b.act[0][2], mm,
                                                               b.t_idx, b.t_rank, b.act[0][3],
                                                               accel ? b.t_bits : nullptr);
            TC_LAUNCH_CHECK(ctx);
            if (accel) {
                const u32 nb = tc_cdiv(nwords, BDIR_TILE);
                bitdir_sum_kernel<<<nb, 256, 0, s>>>(b.t_bits, nwords, b.t_bsum);
                TC_LAUNCH_CHECK(ctx);
                bitdir_spine_kernel<<<1, 1024, 0, s>>>(b.t_bsum, nb);
                TC_LAUNCH_CHECK(ctx);
                bitdir_down_kernel<<<nb, 256, 0, s>>>(b.t_bits, nwords, b.t_bsum, b.t_dir);
                TC_LAUNCH_CHECK(ctx);
                rl.t_bits = b.t_bits; rl.t_dir = b.t_dir;
                const u64 *dkeys = tkeys ? tkeys : ((skeys == b.k0 || skeys == b.k1) ? skeys : nullptr);   // (sorted keys: a directory serves them too)
                if (dkeys) {
                    int kb = tkeys ? (64 - tkeys_shift < SA_KDIR_BITS ? 64 - tkeys_shift : SA_KDIR_BITS) : SA_KDIR_BITS;
                    // (a directory fine enough to leave ~16 keys per entry: more bits than log2 N - 4 only make it sparser)
                    while (kb > 16 && (1ull << kb) > N / 16) kb--;
                    if (env_int("TC_SA_KDIR_SEARCH", 0) != 0) {
                        kdir_build_kernel<<<tc_cdiv((1ull << kb) + 1, 256), 256, 0, s>>>(dkeys, (u32)N, kb, b.kdir);
                        TC_LAUNCH_CHECK(ctx);
                    } else {
                        const u64 entries = (1ull << kb) + 1;
                        const u32 nb = tc_cdiv(entries, KDF_CHUNK);
                        u32 *bmin = b.kdir + ((size_t)1 << SA_KDIR_BITS) + 2;
                        tc_memset_async(ctx, b.kdir, 0xff, entries * sizeof(u32));
                        kdir_mark_kernel<<<tc_cdiv(N, 256), 256, 0, s>>>(dkeys, (u32)N, kb, b.kdir);
                        TC_LAUNCH_CHECK(ctx);
                        kdir_fill_min_kernel<<<nb, 256, 0, s>>>(b.kdir, entries, bmin);
                        TC_LAUNCH_CHECK(ctx);
                        kdir_fill_spine_kernel<<<1, 1024, 0, s>>>(bmin, nb);
                        TC_LAUNCH_CHECK(ctx);
                        kdir_fill_apply_kernel<<<nb, 256, 0, s>>>(b.kdir, entries, bmin);
                        TC_LAUNCH_CHECK(ctx);
                    }
                    rl.kdir = b.kdir; rl.kdir_bits = kb;
                }
            }
        }
    }

    trace(dense ? "ranks: dense ISA" : "ranks: sparse table", m);
    // 6. prefix doubling on the tied suffixes
    int cur = 0;
    if (env_int("TC_SA_H_START", 0) > 0) h_start = (u64)env_int("TC_SA_H_START", 0);  // experiments: any h <= sorted depth is valid
    u64 h = h_start;
    // chain rounds (tc_chain.hpp): when a round sheds next to nothing (periodic text, a long run of one symbol) the next one orders every group
    // by how long its members keep seeing the same thing at + h, + 2 h, .. -- two passes of the same sort at one h.
    // TC_SA_CHAIN: 0 never, 1 (default) after a PLAIN round that resolved < 1/256 of a set of >= 2^20 members, 2 every round.
    // (Not in the very first doubling round, however few suffixes round 0 resolved: a chain is cut wherever two residue classes
    // of the period share their h symbols -- members of the merged group see two different ranks at + h, one of them is not the
    // reference -- and the cut repeats with the period, so all members of a class get the SAME k.  One such coincidence in a
    // 1 MiB period at h = 21 left 97 % of a 1 GiB record tied after the chain round; a plain round first splits the merged
    // groups, and 2 h symbols rarely coincide: chain round at 42 -> everything resolved.)
    const int chain_env = env_int("TC_SA_CHAIN", 1);
    int keymode = 0;      // 0: key2 = rank[i + h]; 1: the chain code; 2: the rank the member's terminal sees
    u64 prev_mm = 0;      // members of the last plain doubling round (0: none yet, or a chain round came since)
    // back-off: text that is repetitive without being periodic (a Fibonacci or Thue-Morse word: every round keeps nearly all of it
    // tied, but its chains are short) would pay a chain round -- two passes -- at every other doubling for nothing (2^28 bytes:
    // 762 instead of 538 ms).  A chain round that resolved less than an eighth of its members makes the next attempt wait
    // 2, 4 plain rounds; after three such rounds there are no more (forced rounds, TC_SA_CHAIN=2, ignore this)
    int chain_fail = 0, chain_wait = 0;
    u64 chain_m0 = 0;
    while (m > 0) {
        if (st.rounds >= TC_MAX_ROUNDS) TC_FAIL(ctx, TC_ERR_INTERNAL, "suffix sort did not converge");
        u32 mm = (u32)m;
        u32 hh = h > N ? (u32)N : (u32)h;
        if (trace_on) {   // members per group-size class (k0 is free at this point in both modes' first use)
            u64 *gh = reinterpret_cast<u64 *>(b.hist);
            tc_memset_async(ctx, gh, 0, 32 * sizeof(u64));
            group_size_hist_kernel<<<tc_cdiv(mm, 256), 256, 0, s>>>(b.act[cur][0], b.act[cur][2], mm, gh);
            u64 hh32[32];
            tc_d2h(ctx, hh32, gh, sizeof hh32);
            (void)hipStreamSynchronize(s);
            fprintf(stderr, "textcomp:   members by group size 2^c:");
            for (int c = 0; c < 32; c++) if (hh32[c]) fprintf(stderr, " %d:%.1f%%", c, 100.0 * (double)hh32[c] / (double)mm);
            fprintf(stderr, "\n");
            trace_t0 = std::chrono::steady_clock::now();
        }
        // dense: the round-0 key buffers are dead; sparse: they hold the sorted keys
        u64 *k2 = dense ? b.k0 : b.sk[0], *k2alt = dense ? b.k1 : b.sk[1];
        u32 *kv = dense ? b.v0 : b.sv[0], *kvalt = dense ? b.v2 : b.sv[1];
        RadixPlan p2;
        p2.add_range(0, rbits);
        p2.add_range(32, 32 + rbits);
        RadixPlanDev pd2;
        pd2.npass = p2.npass;
        for (int p = 0; p < p2.npass; p++) { pd2.shift[p] = p2.shift[p]; pd2.mask[p] = p2.mask[p]; }
        // large rounds: digit histograms on the way; dense: the suffix starts are sorted along
        // (no gather through the active set afterwards)
        const bool seg_round = env_int("TC_SA_SEG", 1) != 0 && mm >= (u32)env_int("TC_SA_SEG_MIN", 1 << 16);
        const bool fuse_hist = mm >= (1u << 20) && !seg_round && keymode == 0;
        const bool vals_idx = dense;
        if (fuse_hist) tc_memset_async(ctx, b.hist, 0, sizeof(u32) * RDX_MAX_PASSES * RDX_BINS);
        // (both bitmaps are copied into the reference table's memory once the flags are made -- the table is dead by then --
        // so that they survive the sort of pass 1: pass 2 asks again which members were on path)
        const u32 chain_words = (u32)(N / 64 + 1);
        u64 *chain_path = reinterpret_cast<u64 *>(b.chain_ref), *chain_sign = chain_path + chain_words;
        if (keymode == 0 && seg_round && hh >= 4 && h < N && chain_env != 0 &&
            (chain_env == 2 || (mm >= (1u << 20) && prev_mm > 0 && (prev_mm - m) * 256 < prev_mm && chain_wait == 0 && chain_fail < 3))) {
            // reference ranks, on-path / sign bits of every tied position, their scan along stride h -> a code per position
            const ChainDims cd = chain_dims(N, hh);
            u64 *pathbits = b.seg.segbits, *signbits = b.seg.ybits;   // (free here: the sort writes them anew)
            u32 *any = b.chain_summ + chain_any_offset();
            tc_memset_async(ctx, b.chain_ref, 0xff, (size_t)N * sizeof(u32));
            chain_ref_kernel<<<tc_cdiv(mm, 256), 256, 0, s>>>(b.act[cur][0], b.act[cur][1], b.act[cur][2], rl, mm, hh, b.chain_ref);
            TC_LAUNCH_CHECK(ctx);
            if (dense) {
                u32 fgrid = tc_cdiv(chain_words, 4);
                if (fgrid > 16384) fgrid = 16384;
                chain_flags_kernel<<<fgrid, 256, 0, s>>>(b.isa, b.chain_ref, (u32)N, hh, pathbits, signbits, chain_words);
            } else {
                tc_memset_async(ctx, pathbits, 0, (size_t)chain_words * sizeof(u64));
                tc_memset_async(ctx, signbits, 0, (size_t)chain_words * sizeof(u64));
                u32 fgrid = tc_cdiv(mm, 256);
                if (fgrid > 16384) fgrid = 16384;
                chain_flags_members_kernel<<<fgrid, 256, 0, s>>>(b.act[cur][1], b.act[cur][2], rl, mm, hh, b.chain_ref, pathbits, signbits);
            }
            TC_LAUNCH_CHECK(ctx);
            TC_HIP(ctx, hipMemcpyAsync(chain_path, pathbits, (size_t)chain_words * sizeof(u64), hipMemcpyDeviceToDevice, s));
            TC_HIP(ctx, hipMemcpyAsync(chain_sign, signbits, (size_t)chain_words * sizeof(u64), hipMemcpyDeviceToDevice, s));
            {   // (row blocks that hold a position on path; a block's words are shared by up to 1024 workgroups)
                tc_memset_async(ctx, any, 0, (size_t)cd.nb * sizeof(u32));
                u64 parts = ((u64)cd.bk * cd.h / 64) / 4096 + 1;
                if (parts > 1024) parts = 1024;
                chain_blockany_kernel<<<dim3(cd.nb, (u32)parts), 256, 0, s>>>(chain_path, cd, any);
                TC_LAUNCH_CHECK(ctx);
            }
            const u32 cgrid = (u32)tc_cdiv((u64)cd.nb * cd.h, 256);
            if (cd.nb > 1) {
                chain_scan_a_kernel<<<cgrid, 256, 0, s>>>(chain_path, chain_sign, cd, any, b.chain_summ);
                TC_LAUNCH_CHECK(ctx);
                chain_scan_b_kernel<<<tc_cdiv(cd.h, 256), 256, 0, s>>>(b.chain_summ, cd);
                TC_LAUNCH_CHECK(ctx);
            }
            chain_scan_c_kernel<<<cgrid, 256, 0, s>>>(chain_path, chain_sign, cd, any, b.chain_summ, b.chain_code);
            TC_LAUNCH_CHECK(ctx);
            keymode = 1;
            chain_m0 = m;
            st.chain_rounds++;
            trace("chain round: codes", N);
            if (trace_on) {
                unsigned long long *dg = reinterpret_cast<unsigned long long *>(b.hist);
                tc_memset_async(ctx, dg, 0, 8 * sizeof(u64));
                chain_diag_kernel<<<4096, 256, 0, s>>>(b.act[cur][1], b.act[cur][2], mm, chain_path, chain_sign, b.chain_code, dg);
                u64 hd[8];
                tc_d2h(ctx, hd, dg, sizeof hd);
                (void)hipStreamSynchronize(s);
                fprintf(stderr, "textcomp:   chain tables (h = %u, %u x %u cells of %u rows, %s ranks): members %u, on path %llu, k = 0: %llu, largest k %llu\n",
                        hh, cd.nb, cd.h, cd.bk, dense ? "dense" : "sparse", mm, (unsigned long long)hd[0], (unsigned long long)hd[1], (unsigned long long)hd[2]);
                trace_t0 = std::chrono::steady_clock::now();
            }
        }
        if (keymode == 1) {
            chain_key1_kernel<<<tc_cdiv(mm, 256), 256, 0, s>>>(b.act[cur][1], b.act[cur][2], chain_path, chain_sign, b.chain_code, mm, k2, vals_idx ? kv : nullptr);
        } else if (keymode == 2) {
            u32 kgrid = tc_cdiv(mm, 256);
            if (kgrid > 65536) kgrid = 65536;
            chain_key2_kernel<<<kgrid, 256, 0, s>>>(b.act[cur][1], b.act[cur][2], chain_path, chain_sign, b.chain_code, rl, mm, hh, k2, vals_idx ? kv : nullptr);
        } else {
            // one lookup per thread for small sets (latency-bound); coarser when histograms are kept
            u32 kgrid = fuse_hist ? tc_cdiv(mm, 256 * 8) : tc_cdiv(mm, 256);
            if (fuse_hist && kgrid > 8192) kgrid = 8192;
            // (a few thousand members looked up by counts inside unsorted buckets: the kernel takes a wave per member)
            if (!fuse_hist && !vals_idx && !rl.isa && !rl.skeys && rl.tkeys && mm <= 65536u) kgrid = tc_cdiv(mm, 4);
            if (fuse_hist) key2_kernel<true><<<kgrid, 256, 0, s>>>(b.act[cur][1], b.act[cur][2], rl, mm, hh, k2, vals_idx ? kv : nullptr, pd2, b.hist);
            else key2_kernel<false><<<kgrid, 256, 0, s>>>(b.act[cur][1], b.act[cur][2], rl, mm, hh, k2, vals_idx ? kv : nullptr, pd2, b.hist);
        }
        TC_LAUNCH_CHECK(ctx);
        trace("round: keys (rank lookups)", mm);
        RadixBuffers r2;
        r2.keys = k2; r2.keys_alt = k2alt; r2.vals = kv; r2.vals_alt = kvalt;
        r2.hist = b.hist; r2.status = b.rstatus; r2.status_cap = radix_status_words(N);
        if (seg_round) {
            // the members are in SA order, so every group is a run of equal top key halves: a sort inside the runs
            // (tc_seg.hpp) instead of eight stable passes over the whole set
            if (!vals_idx) {
                seg_iota_kernel<<<tc_cdiv(mm, 256), 256, 0, s>>>(kv, mm);
                TC_LAUNCH_CHECK(ctx);
            }
            seg_sort_pairs(ctx, b.seg, k2, kv, k2alt, kvalt, mm, keymode == 1 ? 32 : rbits);
            st.seg_rounds++;
        } else {
            radix_sort_pairs(ctx, r2, mm, p2, /*gen_idx=*/!vals_idx, /*hist_ready=*/fuse_hist);
        }
        trace(seg_round ? "round: segmented sort" : "round: radix passes", mm);
        GroupArgs gr = {};
        gr.keys = r2.keys; gr.count = mm; gr.vals = r2.vals; gr.vals_are_idx = vals_idx ? 1 : 0;
        gr.in_slot = b.act[cur][0]; gr.in_idx = b.act[cur][1]; gr.in_tpos = b.act[cur][3];
        gr.isa = dense ? b.isa : nullptr; gr.t_rank = dense ? nullptr : b.t_rank;
        gr.out_slot = b.act[cur ^ 1][0]; gr.out_idx = b.act[cur ^ 1][1];
        gr.out_grp = b.act[cur ^ 1][2]; gr.out_tpos = b.act[cur ^ 1][3];
        // dense, large round: ranks by regions (pairs into the scratch key buffer; the sorted keys are dead
        // once the groups are made, so the partitioned pairs go there)
        const bool gr_pairs = dense && mm >= bin_min;
        if (gr_pairs) gr.pairs = r2.keys_alt;
        run_group(false, gr, sa);
        if (gr_pairs) apply_pairs(r2.keys_alt, mm, r2.keys);
        if (keymode != 2) {   // (a chain round is ONE entry -- its first pass's: every entry is a doubling of h, so the rounds stay <= 32)
            st.m[st.rounds] = m; st.key_bytes[st.rounds] = 8; st.passes[st.rounds] = seg_round ? 1u : (u32)p2.npass;
            st.h[st.rounds] = hh;
            st.rounds++;
        } else {
            st.passes[st.rounds - 1]++;
        }
        m = fetch_m();
        trace("round: groups", mm);
        cur ^= 1;
        if (keymode == 1) keymode = 2;   // (the second pass of a chain round: same h)
        else {
            if (keymode == 2) {
                if ((chain_m0 - m) * 8 < chain_m0) chain_wait = 1 << ++chain_fail;
            } else if (chain_wait > 0) chain_wait--;
            prev_mm = keymode == 2 ? 0 : mm;
            keymode = 0;
            h *= 2;
        }
    }
    primary_kernel<<<1, 64, 0, s>>>(rl, ctx->d_scalars);
    TC_LAUNCH_CHECK(ctx);
    tc_d2h(ctx, ctx->h_scalars, ctx->d_scalars, sizeof(u64));
    TC_HIP(ctx, hipStreamSynchronize(s));
    *primary = ctx->h_scalars[0];
    st.sigma = cfg.sigma_text + 1;
    st.radix_launches = 0;
    st.ms_radix = 0;
    for (int i = 0; i < ctx->pev_used; i++) {  // stream is idle here (last group sync)
        float ms = 0;
        if (hipEventElapsedTime(&ms, ctx->pev[2 * i], ctx->pev[2 * i + 1]) == hipSuccess) {
            st.ms_radix += ms;
            st.radix_launches++;
        }
    }
}

// ------------------------------------------------------------------- accessors
template <class Acc>
static Acc make_acc(const void *d_src, i64 primary);
template <>
BwtAcc make_acc<BwtAcc>(const void *d_src, i64 primary) {
    return BwtAcc{reinterpret_cast<const u8 *>(d_src), primary};
}
template <>
SymAcc make_acc<SymAcc>(const void *d_src, i64) {
    return SymAcc{reinterpret_cast<const i16 *>(d_src)};
}
template <>
U16Acc make_acc<U16Acc>(const void *d_src, i64) {
    return U16Acc{reinterpret_cast<const u16 *>(d_src)};
}

// Symbol histogram of an accessor stream -> host counts257.
template <class Acc>
static void sym_hist_host(tc_ctx *ctx, Acc acc, u64 N, u32 *d_counts, u32 *counts257) {
    tc_memset_async(ctx, d_counts, 0, 260 * sizeof(u32));
    u32 grid = tc_cdiv(N, 256 * 32);
    if (grid > 2048) grid = 2048;
    sym_hist_kernel<Acc><<<grid, 256, 0, ctx->stream>>>(acc, N, d_counts);
    TC_LAUNCH_CHECK(ctx);
    tc_d2h(ctx, counts257, d_counts, 257 * sizeof(u32));
    TC_HIP(ctx, hipStreamSynchronize(ctx->stream));
}

// ------------------------------------------------------------------------- MTF
template <class Acc, int ROWS>
static void mtf_general_launch(tc_ctx *ctx, Acc acc, u64 N, const Lut16 &lut, u16 *lists,
                               u32 *seen, u32 chunks, u16 *d_idx) {
    hipStream_t s = ctx->stream;
    mtf_gen_summary_kernel<Acc, ROWS><<<chunks, 64, 0, s>>>(acc, N, lut, lists, seen);
    TC_LAUNCH_CHECK(ctx);
    mtf_gen_scan_kernel<ROWS><<<1, 64 * MTFG_SCAN_WAVES, 0, s>>>(lists, seen, chunks);
    TC_LAUNCH_CHECK(ctx);
    mtf_gen_apply_kernel<Acc, ROWS><<<chunks, 64, 0, s>>>(acc, N, lut, lists, d_idx);
    TC_LAUNCH_CHECK(ctx);
}

// any sigma > 16: timestamps (tc_mtf.hpp, "general path, timestamps"); the final list lands in `flist`
template <class Acc, int ROWS>
static void mtf_ts_launch(tc_ctx *ctx, Acc acc, u64 N, const Lut16 &lut, u32 sigma, u32 *ts, u32 *seg, u16 *flist,
                          u16 *d_idx) {
    hipStream_t s = ctx->stream;
    const u32 chunks = tc_cdiv(N, TS_CH), nseg = tc_cdiv(chunks, TS_SEG);
    mtf_ts_last_kernel<Acc><<<chunks, 256, 0, s>>>(acc, N, lut, ts);
    TC_LAUNCH_CHECK(ctx);
    mtf_ts_scan_kernel<0><<<nseg, TS_STRIDE, 0, s>>>(ts, chunks, seg, nseg, sigma);
    TC_LAUNCH_CHECK(ctx);
    mtf_ts_scan_kernel<1><<<1, TS_STRIDE, 0, s>>>(ts, chunks, seg, nseg, sigma);
    TC_LAUNCH_CHECK(ctx);
    mtf_ts_scan_kernel<2><<<nseg, TS_STRIDE, 0, s>>>(ts, chunks, seg, nseg, sigma);
    TC_LAUNCH_CHECK(ctx);
    mtf_ts_final_kernel<<<1, TS_STRIDE, 0, s>>>(seg + (size_t)nseg * TS_STRIDE, sigma, flist);
    TC_LAUNCH_CHECK(ctx);
    mtf_ts_apply_kernel<Acc, ROWS><<<tc_cdiv(chunks, TS_WPB * TS_ILP), 64 * TS_WPB, 0, s>>>(acc, N, lut, ts, d_idx, chunks);
    TC_LAUNCH_CHECK(ctx);
}

// sigma <= 256: one chunk per lane (tc_mtf.hpp, "general path, lane chunks")
template <class Acc, int ROWS>
static void mtf_lane_launch(tc_ctx *ctx, Acc acc, u64 N, const Alphabet &al, u16 *lists, u32 *seen,
                            u16 *d_idx) {
    hipStream_t s = ctx->stream;
    GmArgs a;
    a.N = N; a.sigma = al.sigma; a.ls = ((al.sigma + 3) / 4) | 1u;
    for (int v = 0; v < 257; v++) a.lut.v[v] = (u8)al.code_of_sym[v];
    a.lists = lists; a.seen = seen; a.idx = d_idx;
    const u32 tiles = tc_cdiv(N, GM_TILE);
    const size_t lds = gm_lds_bytes(a.ls);
    TC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(mtf_gm_kernel<Acc, ROWS, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    TC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(mtf_gm_kernel<Acc, ROWS, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    mtf_gm_kernel<Acc, ROWS, false><<<tiles, GM_NT, lds, s>>>(acc, a);
    TC_LAUNCH_CHECK(ctx);
    mtf_gen_scan_kernel<ROWS><<<1, 64 * MTFG_SCAN_WAVES, 0, s>>>(lists, seen, tiles);
    TC_LAUNCH_CHECK(ctx);
    mtf_gm_kernel<Acc, ROWS, true><<<tiles, GM_NT, lds, s>>>(acc, a);
    TC_LAUNCH_CHECK(ctx);
}

// seqToMTF on the device.  counts257 (host) may be null: then it is measured here.
// d_idx8 (optional): for sigma <= 16 the indices are written THERE, one byte each (*used8 = true)
template <class Acc>
static void mtf_encode_device(tc_ctx *ctx, Arena &A, Acc acc, u64 N, const u32 *counts257,
                              u16 *d_idx, i16 *final_list, u32 *sigma, bool dry, u8 *d_idx8 = nullptr,
                              bool *used8 = nullptr) {
    const u32 tiles = tc_cdiv(N, MTF_TILE);
    const u32 chunks = tc_cdiv(N, MTFG_CH);
    u32 *d_counts = A.get<u32>(260);
    u64 *t_perm = A.get<u64>(tiles + 1);
    u32 *t_mask = A.get<u32>(tiles + 1 > 512 ? tiles + 1 : 512);
    u16 *lists = A.get<u16>(((size_t)chunks + 4) * 320);   // (also: [N / TS_CH + 1][TS_STRIDE] u32 timestamps)
    u32 *seen = A.get<u32>(chunks + 1);
    u32 *ts_seg = A.get<u32>(((size_t)tc_cdiv(tc_cdiv(N, TS_CH), TS_SEG) + 2) * TS_STRIDE);
    u16 *ts_flist = A.get<u16>(TS_STRIDE);
    if (dry) return;
    hipStream_t s = ctx->stream;
    u32 local[257];
    if (!counts257) {
        sym_hist_host<Acc>(ctx, acc, N, d_counts, local);
        counts257 = local;
    }
    Alphabet al;
    al.build(counts257);
    *sigma = al.sigma;
    const bool force_general = env_int("TC_MTF_FORCE_GENERAL", 0) != 0;
    if (al.sigma <= 16 && !force_general) {
        Lut8 lut;
        for (int v = 0; v < 257; v++) lut.v[v] = (u8)al.code_of_sym[v];
        // fast path: every tile recovers its incoming list by a short backward scan
        bool fast_ok = false;
        const bool scan_failed = ctx->mtf_fastin_failed != 0;   // (this record's one-kernel attempt: long runs in the column)
        ctx->mtf_fastin_failed = 0;
        if (env_int("TC_MTF_FASTIN", 1) != 0 && !scan_failed) {
            u32 *flag = reinterpret_cast<u32 *>(ctx->d_scalars + 15);
            tc_memset_async(ctx, flag, 0, sizeof(u64));
            if (d_idx8 && al.sigma <= 8 && env_int("TC_MTF_SMALL", 1) != 0)   // (a DNA record: the list in 32 bits)
                mtf_nib_apply_kernel<Acc, true, u8, true><<<tiles, MTF_NT, 0, s>>>(acc, N, lut, t_perm, d_idx8, al.sigma, flag);
            else if (d_idx8) mtf_nib_apply_kernel<Acc, true, u8><<<tiles, MTF_NT, 0, s>>>(acc, N, lut, t_perm, d_idx8, al.sigma, flag);
            else mtf_nib_apply_kernel<Acc, true><<<tiles, MTF_NT, 0, s>>>(acc, N, lut, t_perm, d_idx, al.sigma, flag);
            TC_LAUNCH_CHECK(ctx);
            mtf_nib_final_kernel<Acc><<<1, 64, 0, s>>>(acc, N, lut, al.sigma, t_perm + tiles, flag);
            TC_LAUNCH_CHECK(ctx);
            tc_d2h(ctx, &ctx->h_scalars[15], ctx->d_scalars + 15, sizeof(u64));
            tc_d2h(ctx, &ctx->h_scalars[8], t_perm + tiles, sizeof(u64));
            TC_HIP(ctx, hipStreamSynchronize(s));
            fast_ok = ((u32)ctx->h_scalars[15]) == 0;
        }
        if (!fast_ok) {
            mtf_nib_summary_kernel<Acc><<<tiles, MTF_NT, 0, s>>>(acc, N, lut, t_perm, t_mask);
            TC_LAUNCH_CHECK(ctx);
            mtf_nib_scan_kernel<<<1, MTF_NT, 0, s>>>(t_perm, t_mask, tiles);
            TC_LAUNCH_CHECK(ctx);
            if (d_idx8) mtf_nib_apply_kernel<Acc, false, u8><<<tiles, MTF_NT, 0, s>>>(acc, N, lut, t_perm, d_idx8, al.sigma, nullptr);
            else mtf_nib_apply_kernel<Acc, false><<<tiles, MTF_NT, 0, s>>>(acc, N, lut, t_perm, d_idx, al.sigma, nullptr);
            TC_LAUNCH_CHECK(ctx);
            tc_d2h(ctx, &ctx->h_scalars[8], t_perm + tiles, sizeof(u64));
            TC_HIP(ctx, hipStreamSynchronize(s));
        }
        u64 perm = ctx->h_scalars[8];
        for (u32 i = 0; i < al.sigma; i++) final_list[i] = al.sym_of_code[(perm >> (4 * i)) & 15];
        if (used8) *used8 = d_idx8 != nullptr;
    } else {
        Lut16 lut;
        for (int v = 0; v < 257; v++) lut.v[v] = al.code_of_sym[v];
        int rows;
        u32 last;  // slot of the final list
        // timestamps: the default beyond 64 symbols (1 GiB: uniform bytes 258 -> 37 ms, ASCII 81 -> 26 ms,
        // Zipf words over all byte values 126 -> ~40 ms; up to 64 symbols the lane chunks are ahead on
        // BWT-like streams: Zipf text, sigma 28, 12 against 23 ms).  TC_MTF_TS=0: never, 2: whenever sigma > 16.
        const int ts_mode = env_int("TC_MTF_TS", 1);
        if (ts_mode != 0 && (al.sigma > 64 || ts_mode == 2) && N + 300 < (1ull << 32)) {
            rows = (int)((al.sigma + 63) / 64);
            u32 *ts = reinterpret_cast<u32 *>(lists);
            if (rows == 1) mtf_ts_launch<Acc, 1>(ctx, acc, N, lut, al.sigma, ts, ts_seg, ts_flist, d_idx);
            else if (rows == 2) mtf_ts_launch<Acc, 2>(ctx, acc, N, lut, al.sigma, ts, ts_seg, ts_flist, d_idx);
            else if (rows == 3) mtf_ts_launch<Acc, 3>(ctx, acc, N, lut, al.sigma, ts, ts_seg, ts_flist, d_idx);
            else if (rows == 4) mtf_ts_launch<Acc, 4>(ctx, acc, N, lut, al.sigma, ts, ts_seg, ts_flist, d_idx);
            else mtf_ts_launch<Acc, 5>(ctx, acc, N, lut, al.sigma, ts, ts_seg, ts_flist, d_idx);
            std::vector<u16> fl(al.sigma);
            tc_d2h(ctx, fl.data(), ts_flist, fl.size() * sizeof(u16));
            TC_HIP(ctx, hipStreamSynchronize(s));
            for (u32 i = 0; i < al.sigma; i++) final_list[i] = al.sym_of_code[fl[i]];
            return;
        }
        // large alphabets: lane chunks unless the sampled average rank says the symbols are spread
        // uniformly (tc_mtf.hpp, "which general path?")
        auto prefers_wave = [&](const Alphabet &ax) {
            if (ax.sigma <= 128 || N < (u64)MRS_BLOCKS * 256 * MRS_WIN || env_int("TC_MTF_RANK_SAMPLE", 1) == 0)
                return false;
            u64 *d_sum = ctx->d_scalars + 23;
            tc_memset_async(ctx, d_sum, 0, sizeof(u64));
            mtf_rank_sample_kernel<Acc><<<MRS_BLOCKS, 256, 0, s>>>(acc, N, d_sum);
            TC_LAUNCH_CHECK(ctx);
            tc_d2h(ctx, &ctx->h_scalars[23], d_sum, sizeof(u64));
            TC_HIP(ctx, hipStreamSynchronize(s));
            const u64 avg = ctx->h_scalars[23] / ((u64)MRS_BLOCKS * 256);   // distinct symbols per window
            return avg >= (u64)env_int("TC_MTF_WAVE_MIN_DISTINCT", 128);
        };
        if constexpr (std::is_same<Acc, BwtAcc>::value) {
            // sigma = 257 with the one sentinel of a BWT: 256-symbol lane chunks + fix-ups (tc_mtf.hpp)
            u32 bytes_only[257];
            memcpy(bytes_only, counts257, sizeof bytes_only);
            bytes_only[0] = 0;
            Alphabet ab;
            ab.build(bytes_only);   // 256 symbols, code = byte value
            if (al.sigma == 257 && acc.primary > 0 && (u64)acc.primary < N &&
                env_int("TC_MTF_WAVE_CHUNKS", 0) == 0 && env_int("TC_MTF_SENTINEL_SPLIT", 1) != 0 &&
                !prefers_wave(ab)) {
                BwtAcc dup = acc;
                dup.dup = 1;
                mtf_lane_launch<BwtAcc, 4>(ctx, dup, N, ab, lists, seen, d_idx);
                u32 *first = t_mask;    // 512 words (tiles + 1 >= 1: sized below)
                tc_memset_async(ctx, first, 0xff, 512 * sizeof(u32));
                u32 grid = tc_cdiv(N, 256 * 64);
                if (grid > 4096) grid = 4096;
                mtf257_first_kernel<<<grid, 256, 0, s>>>(acc.L, N, (u64)acc.primary, first);
                TC_LAUNCH_CHECK(ctx);
                mtf257_fix_kernel<<<1, 512, 0, s>>>(d_idx, first, (u64)acc.primary, ctx->d_scalars + 20);
                TC_LAUNCH_CHECK(ctx);
                std::vector<u16> fl(256);
                tc_d2h(ctx, fl.data(), lists + (size_t)tc_cdiv(N, GM_TILE) * 256, 256 * sizeof(u16));
                tc_d2h(ctx, &ctx->h_scalars[20], ctx->d_scalars + 20, 2 * sizeof(u64));
                TC_HIP(ctx, hipStreamSynchronize(s));
                const u32 after = (u32)ctx->h_scalars[21];   // distinct values met after the sentinel
                for (u32 i = 0, k = 0; i < 257; i++)
                    final_list[i] = i == after ? (i16)-1 : ab.sym_of_code[fl[k++]];
                return;
            }
        }
        if (al.sigma <= 256 && env_int("TC_MTF_WAVE_CHUNKS", 0) == 0 && !prefers_wave(al)) {
            rows = (int)((al.sigma + 63) / 64);
            last = tc_cdiv(N, GM_TILE);
            if (rows == 1) mtf_lane_launch<Acc, 1>(ctx, acc, N, al, lists, seen, d_idx);
            else if (rows == 2) mtf_lane_launch<Acc, 2>(ctx, acc, N, al, lists, seen, d_idx);
            else if (rows == 3) mtf_lane_launch<Acc, 3>(ctx, acc, N, al, lists, seen, d_idx);
            else mtf_lane_launch<Acc, 4>(ctx, acc, N, al, lists, seen, d_idx);
        } else {  // sigma = 257 (nine-bit codes): one chunk per wave
            rows = al.sigma <= 64 ? 1 : (al.sigma <= 128 ? 2 : 5);
            last = chunks;
            if (rows == 1) mtf_general_launch<Acc, 1>(ctx, acc, N, lut, lists, seen, chunks, d_idx);
            else if (rows == 2) mtf_general_launch<Acc, 2>(ctx, acc, N, lut, lists, seen, chunks, d_idx);
            else mtf_general_launch<Acc, 5>(ctx, acc, N, lut, lists, seen, chunks, d_idx);
        }
        std::vector<u16> fl(rows * 64);
        tc_d2h(ctx, fl.data(), lists + (size_t)last * (rows * 64), fl.size() * sizeof(u16));
        TC_HIP(ctx, hipStreamSynchronize(s));
        for (u32 i = 0; i < al.sigma; i++) final_list[i] = al.sym_of_code[fl[i]];
    }
}

// ------------------------------------------------------------------------- RLE
template <class Acc, class SymT>
static void rle_encode_device(tc_ctx *ctx, Arena &A, Acc acc, u64 N, u32 *d_counts, SymT *d_syms,
                              u64 cap, u64 *total, bool dry, bool small16 = false /* values < 16 (byte-wide stream) */) {
    const bool idx_stream = std::is_same<Acc, U16Acc>::value || std::is_same<Acc, U8Acc>::value;
    const u32 tiles = tc_cdiv(N, idx_stream ? RLE16_TILE : RLE_TILE);
    const u32 btiles = tc_cdiv(N, RN_TILE);                     // tiles of the blocked kernel (byte-wide index stream)
    const u32 stiles = tiles > btiles ? tiles : btiles;
    u64 *status = A.get<u64>(2 * (size_t)stiles + 4);
    if (dry) return;
    tc_memset_async(ctx, status, 0, (2 * (size_t)stiles + 4) * sizeof(u64));
    if constexpr (std::is_same<Acc, U8Acc>::value) {
        // byte-wide index stream (the fused encode): the blocked kernel (tc_pack.hpp); TC_RLE_BLOCKED=0: the striped one
        if ((((uintptr_t)acc.v) & 15) == 0 && env_int("TC_RLE_BLOCKED", 1) != 0) {
            RleBlkArgs b;
            b.src = acc.v; b.N = N; b.counts = d_counts; b.vals = reinterpret_cast<u16 *>(d_syms); b.cap = cap;
            b.status_a = status; b.status_b = status + btiles;
            b.ticket = reinterpret_cast<u32 *>(status + 2 * (size_t)btiles);
            b.scalars = ctx->d_scalars; b.err = ctx->d_err; b.ntiles = btiles;
            b.wide = ((((uintptr_t)d_counts) | ((uintptr_t)d_syms)) & 15) == 0 ? 1u : 0u;
            if (small16) {    // (values < 16: one staged byte per run)
                u32 grid = tc_persistent_grid_for(ctx, rle_blk_kernel<true>, RN_NT, 8);
                if (grid > btiles) grid = btiles;
                rle_blk_kernel<true><<<grid, RN_NT, 0, ctx->stream>>>(b);
            } else {
                u32 grid = tc_persistent_grid_for(ctx, rle_blk_kernel<false>, RN_NT, 8);
                if (grid > btiles) grid = btiles;
                rle_blk_kernel<false><<<grid, RN_NT, 0, ctx->stream>>>(b);
            }
            TC_LAUNCH_CHECK(ctx);
            tc_d2h(ctx, &ctx->h_scalars[2], ctx->d_scalars + 2, sizeof(u64));
            TC_HIP(ctx, hipStreamSynchronize(ctx->stream));
            *total = ctx->h_scalars[2];
            return;
        }
    }
    RleArgs a;
    a.N = N; a.counts = d_counts; a.syms = d_syms; a.cap = cap;
    a.status_pair = status; a.status_sum = status + tiles;
    a.ticket = reinterpret_cast<u32 *>(status + 2 * (size_t)tiles);
    a.scalars = ctx->d_scalars; a.err = ctx->d_err;
    a.diag = env_int("TC_RLE_DIAG", 0);
    if constexpr (std::is_same<Acc, U16Acc>::value) {
        u32 grid = tc_persistent_grid_for(ctx, rle_encode_idx_kernel<u16>, RLE_NT, 2);
        if (grid > tiles) grid = tiles;
        rle_encode_idx_kernel<u16><<<grid, RLE_NT, 0, ctx->stream>>>(acc.v, a);
    } else if constexpr (std::is_same<Acc, U8Acc>::value) {
        u32 grid = tc_persistent_grid_for(ctx, rle_encode_idx_kernel<u8>, RLE_NT, 2);
        if (grid > tiles) grid = tiles;
        rle_encode_idx_kernel<u8><<<grid, RLE_NT, 0, ctx->stream>>>(acc.v, a);
    } else {
        u32 grid = tc_persistent_grid_for(ctx, rle_encode_kernel<Acc, SymT>, RLE_NT, 2);
        if (grid > tiles) grid = tiles;
        rle_encode_kernel<Acc, SymT><<<grid, RLE_NT, 0, ctx->stream>>>(acc, a);
    }
    TC_LAUNCH_CHECK(ctx);
    tc_d2h(ctx, &ctx->h_scalars[2], ctx->d_scalars + 2, sizeof(u64));
    TC_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *total = ctx->h_scalars[2];
}

// seqToMTF and seqToRLE of a last column over at most 8 symbols in one kernel (tc_pack.hpp, mtf_rle_kernel): the
// index stream is never written.  false (nothing written that matters): another alphabet, unaligned run arrays, or a
// tile whose incoming list the backward scan did not recover -- the caller runs the two stages.  TC_MTF_RLE=0: never.
static bool mtf_rle_device(tc_ctx *ctx, Arena &A, BwtAcc acc, u64 N, const u32 *counts257, tc_block *out, u64 cap,
                           u64 *total, u32 *sigma, bool dry) {
    const u32 tiles = tc_cdiv(N, MTF_TILE);
    u64 *status = A.get<u64>(2 * (size_t)tiles + 8);
    if (dry || env_int("TC_MTF_RLE", 1) == 0 || env_int("TC_MTF_FORCE_GENERAL", 0) != 0) return false;
    Alphabet al;
    al.build(counts257);
    if (al.sigma > 8 || N + 64 >= (1ull << 32)) return false;
    hipStream_t s = ctx->stream;
    MtfRleArgs a;
    for (int v = 0; v < 257; v++) a.lut.v[v] = (u8)al.code_of_sym[v];
    tc_memset_async(ctx, status, 0, (2 * (size_t)tiles + 8) * sizeof(u64));
    a.acc = acc; a.N = N; a.sigma = al.sigma;
    a.flag = reinterpret_cast<u32 *>(status + 2 * (size_t)tiles + 1);
    a.counts = out->run_count; a.vals = reinterpret_cast<u16 *>(out->run_value); a.cap = cap;
    a.status_a = status; a.status_b = status + tiles;
    a.ticket = reinterpret_cast<u32 *>(status + 2 * (size_t)tiles);
    a.scalars = ctx->d_scalars; a.err = ctx->d_err; a.ntiles = tiles;
    a.wide = ((((uintptr_t)out->run_count) | ((uintptr_t)out->run_value)) & 15) == 0 ? 1u : 0u;
    mtf_rle_kernel<false><<<tiles, MTF_NT, 0, s>>>(a);
    TC_LAUNCH_CHECK(ctx);
    u64 *d_final = status + 2 * (size_t)tiles + 2;
    mtf_nib_final_kernel<BwtAcc><<<1, 64, 0, s>>>(acc, N, a.lut, al.sigma, d_final, a.flag);
    TC_LAUNCH_CHECK(ctx);
    tc_d2h(ctx, &ctx->h_scalars[15], a.flag, sizeof(u32));
    tc_d2h(ctx, &ctx->h_scalars[8], d_final, sizeof(u64));
    tc_d2h(ctx, &ctx->h_scalars[2], ctx->d_scalars + 2, sizeof(u64));
    TC_HIP(ctx, hipStreamSynchronize(s));
#ifdef MTFRLE_PROFILE
    {
        u64 h[9];
        tc_d2h(ctx, h, ctx->d_scalars + 112, sizeof h);
        TC_HIP(ctx, hipStreamSynchronize(s));
        const double c = (double)(h[8] | 1);
        fprintf(stderr, "mtf_rle: tiles %llu | cycles per tile: stage+B %.0f pass %.0f scan+replay+B %.0f ends+B %.0f lookback A/counts+B %.0f lookback B+B %.0f emit+B %.0f copy-out %.0f\n",
                (unsigned long long)h[8], h[0] / c, h[1] / c, h[2] / c, h[3] / c, h[4] / c, h[5] / c, h[6] / c, h[7] / c);
        tc_memset_async(ctx, ctx->d_scalars + 112, 0, sizeof h);
    }
#endif
    if ((u32)ctx->h_scalars[15] != 0) {
        ctx->mtf_fastin_failed = 1;
        return false;
    }
    const u64 perm = ctx->h_scalars[8];
    for (u32 i = 0; i < al.sigma; i++) out->final_list[i] = al.sym_of_code[(perm >> (4 * i)) & 15];
    *sigma = al.sigma;
    *total = ctx->h_scalars[2];
    return true;
}

// --------------------------------------------------------------- fused pipeline
// bytestringToBWT -> bytestringBWTToMTFB -> runs of the index stream.
static void encode_device(tc_ctx *ctx, const u8 *d_text, u64 n, tc_block *out, u64 cap) {
    const u64 N = n + 1;
    ctx->stats = tc_stats{};
    ctx->stats.n = n; ctx->stats.N = N;
    u8 *d_L = nullptr;
    u16 *d_idx = nullptr;
    u64 primary = 0, total = 0;
    u32 counts[256], counts257[257];
    u32 sigma = 0;
    hipStream_t s = ctx->stream;
    auto plan = [&](Arena &A, bool dry) {
        d_L = A.get<u8>(N + 16);
        d_idx = A.get<u16>(N);
        size_t mark = A.off;
        if (!dry) TC_HIP(ctx, hipEventRecord(ctx->ev[0], s));
        sa_build(ctx, A, d_text, n, nullptr, d_L, &primary, counts, dry);
        size_t end_sa = A.off;
        A.off = mark;  // the suffix-sort buffers are dead: MTF / RLE scratch overlays them
        if (!dry) {
            TC_HIP(ctx, hipEventRecord(ctx->ev[1], s));
            counts257[0] = 1;
            for (int b = 0; b < 256; b++) counts257[1 + b] = counts[b];
        }
        BwtAcc acc{d_L, (i64)primary};
        // small alphabets: the index stream between the two stages is one byte per symbol (the
        // same buffer, half used)
        bool idx8 = false;
        if (mtf_rle_device(ctx, A, acc, N, counts257, out, cap, &total, &sigma, dry)) {   // (sigma <= 8: one kernel)
            TC_HIP(ctx, hipEventRecord(ctx->ev[2], s));
            TC_HIP(ctx, hipEventRecord(ctx->ev[3], s));
            if (A.off < end_sa) A.off = end_sa;
            return;
        }
        mtf_encode_device<BwtAcc>(ctx, A, acc, N, dry ? nullptr : counts257, d_idx,
                                  out->final_list, &sigma, dry, reinterpret_cast<u8 *>(d_idx), &idx8);
        if (!dry) TC_HIP(ctx, hipEventRecord(ctx->ev[2], s));
        if (idx8) {
            U8Acc iacc{reinterpret_cast<const u8 *>(d_idx)};
            rle_encode_device<U8Acc, u16>(ctx, A, iacc, N, out->run_count, out->run_value, cap, &total, dry, sigma <= 16);
        } else {
            U16Acc iacc{d_idx};
            rle_encode_device<U16Acc, u16>(ctx, A, iacc, N, out->run_count, out->run_value, cap, &total, dry);
        }
        if (!dry) TC_HIP(ctx, hipEventRecord(ctx->ev[3], s));
        if (A.off < end_sa) A.off = end_sa;
    };
    Arena dry(nullptr);
    plan(dry, true);
    tc_ws_reserve(ctx, dry.off);
    Arena A(ctx->ws);
    plan(A, false);
    tc_sync_check(ctx);
    out->n = n; out->primary = primary; out->sigma = sigma; out->nruns = total;
    tc_stats &st = ctx->stats;
    st.runs = total;
    (void)hipEventElapsedTime(&st.ms_sa, ctx->ev[0], ctx->ev[1]);
    (void)hipEventElapsedTime(&st.ms_mtf, ctx->ev[1], ctx->ev[2]);
    (void)hipEventElapsedTime(&st.ms_rle, ctx->ev[2], ctx->ev[3]);
    (void)hipEventElapsedTime(&st.ms_total, ctx->ev[0], ctx->ev[3]);
    st.ms_bwt = 0;  // the last column is produced inside the suffix-sort kernels
    if (total > cap) TC_FAIL(ctx, TC_ERR_CAPACITY, "need %llu run slots, have %llu",
                             (unsigned long long)total, (unsigned long long)cap);
}
