// tc_radix_host.hpp -- host driver of the device radix sort (see tc_radix.hpp).
#pragma once
#include "tc_radix.hpp"

// The pass kernel kept after the round-1 experiments (profiles/r01_radix_ablation.txt):
// one 4096-pair tile per block, tile ids from an atomic ticket, per-digit look-back one
// status word per round trip.  Batched look-back, persistent blocks with prefetch, a split
// histogram/scan/scatter pass and sharded tickets were measured and were not faster.
#ifndef RDX_LBB
#define RDX_LBB 1   // status words fetched per look-back round trip
#endif
template <bool GEN>
static void radix_launch_pass(tc_ctx *ctx, RadixBuffers &b, u32 n, int shift, u32 mask,
                              const u32 *bucket_base, u32 tiles, u32 *ticket,
                              const u8 *text, const RadixKeyGen *kg, bool xcd_group) {
    hipStream_t s = ctx->stream;
#ifdef TC_RADIX_DIAG
    if (const char *dg = getenv("TC_DIAG")) shift |= (atoi(dg) & 0xfff) << 8;  // timing-only ablations
#endif
    shift |= xcd_group ? 0x200000 : 0x100000;  // XCD-grouped tile order, or the single safe counter
    if (kg) {
        radix_pass_kernel<true, false, RDX_LBB, false, true><<<tiles, RDX_NT, 0, s>>>(
            b.keys, b.vals, b.keys_alt, b.vals_alt, n, shift, mask, bucket_base, b.status, ticket,
            ctx->d_err, nullptr, text, *kg);
    } else {
        RadixKeyGen none = {};
        radix_pass_kernel<GEN, false, RDX_LBB, false, false><<<tiles, RDX_NT, 0, s>>>(
            b.keys, b.vals, b.keys_alt, b.vals_alt, n, shift, mask, bucket_base, b.status, ticket,
            ctx->d_err, nullptr, nullptr, none);
    }
    TC_LAUNCH_CHECK(ctx);
}

// keygen != null: the FIRST pass builds its keys from `text` (b.keys is not read).
void radix_sort_pairs(tc_ctx *ctx, RadixBuffers &b, u32 n, const RadixPlan &plan, bool gen_idx,
                      bool hist_ready, bool timed, const u8 *text, const RadixKeyGen *keygen,
                      bool xcd_group) {
    if (n == 0 || plan.npass == 0) return;
    RadixPlanDev pd;
    pd.npass = plan.npass;
    for (int p = 0; p < plan.npass; p++) {
        pd.shift[p] = plan.shift[p];
        pd.mask[p] = plan.mask[p];
    }
    hipStream_t s = ctx->stream;
    if (!hist_ready) {
        TC_HIP(ctx, hipMemsetAsync(b.hist, 0, sizeof(u32) * RDX_MAX_PASSES * RDX_BINS, s));
        u32 grid = tc_cdiv(n, 256 * 16);
        if (grid > 2048) grid = 2048;
        radix_hist_kernel<<<grid, 256, 0, s>>>(b.keys, n, pd, b.hist);
        TC_LAUNCH_CHECK(ctx);
    }
    radix_scan_hist_kernel<<<plan.npass, 256, 0, s>>>(b.hist);
    TC_LAUNCH_CHECK(ctx);
    const u32 tiles = tc_cdiv(n, RDX_TILE);
    const size_t words = (size_t)tiles * RDX_BINS + 130;
    // (a sort of a few thousand pairs is mostly launches: one zeroing for all its passes where the buffer has the room)
    const bool once = n <= (1u << 20) && b.status_cap >= (size_t)plan.npass * words;
    u64 *const status0 = b.status;
    if (once) TC_HIP(ctx, hipMemsetAsync(status0, 0, (size_t)plan.npass * words * sizeof(u64), s));
    for (int p = 0; p < plan.npass; p++) {
        if (once) b.status = status0 + (size_t)p * words;
        u32 *ticket = reinterpret_cast<u32 *>(b.status + (size_t)tiles * RDX_BINS);
        if (!once) TC_HIP(ctx, hipMemsetAsync(b.status, 0, words * sizeof(u64), s));
        const bool ev = timed && ctx->profile && ctx->pev_used < 16;
        if (ev) TC_HIP(ctx, hipEventRecord(ctx->pev[2 * ctx->pev_used], s));
        if (p == 0 && keygen)
            radix_launch_pass<true>(ctx, b, n, plan.shift[p], plan.mask[p], b.hist + p * RDX_BINS, tiles,
                                    ticket, text, keygen, xcd_group);
        else if (gen_idx && p == 0)
            radix_launch_pass<true>(ctx, b, n, plan.shift[p], plan.mask[p], b.hist + p * RDX_BINS, tiles,
                                    ticket, nullptr, nullptr, xcd_group);
        else
            radix_launch_pass<false>(ctx, b, n, plan.shift[p], plan.mask[p], b.hist + p * RDX_BINS, tiles,
                                     ticket, nullptr, nullptr, xcd_group);
        if (ev) {
            TC_HIP(ctx, hipEventRecord(ctx->pev[2 * ctx->pev_used + 1], s));
            ctx->pev_used++;
        }
        u64 *tk = b.keys; b.keys = b.keys_alt; b.keys_alt = tk;
        u32 *tv = b.vals; b.vals = b.vals_alt; b.vals_alt = tv;
    }
    b.status = status0;
}
