// tc_radix_host.hpp -- host driver of the device radix sort (see tc_radix.hpp).
#pragma once
#include "tc_radix.hpp"

// Pass-kernel variants (TC_RADIX_VARIANT): how a tile learns its bucket offsets.
//   0 ticket per tile, look-back 1 status word per round trip
//   1 ticket per tile, look-back 8 per round trip
//   2 persistent blocks (one ticket per block, prefetch), look-back 8
//   3 split: tile histograms + scan + scatter (no look-back; one extra key read)
//   4 ticket per tile, look-back 4
static int radix_variant() {
    const char *e = getenv("TC_RADIX_VARIANT");
    return e && *e ? atoi(e) : 0;
}

template <bool GEN>
static void radix_launch_pass(tc_ctx *ctx, int variant, RadixBuffers &b, u32 n, int shift, u32 mask,
                              const u32 *bucket_base, u32 tiles, u32 *ticket) {
    hipStream_t s = ctx->stream;
    if (const char *dg = getenv("TC_DIAG")) shift |= (atoi(dg) & 0xfff) << 8;
    if (ctx->safe_tickets || !getenv("TC_SHARDED_TICKETS")) shift |= 0x100000;  // single counter (sharding measured no gain)  // timing-only diagnostics, wrong output
    if (variant == 3) {
        u32 *matrix = reinterpret_cast<u32 *>(b.status);
        u32 *tile_offs = matrix + (size_t)tiles * RDX_BINS;
        u32 *part = reinterpret_cast<u32 *>(b.status + (size_t)tiles * RDX_BINS + 8);
        const u64 len = (u64)tiles * RDX_BINS;
        const u32 nparts = tc_cdiv(len, 4096);
        radix_tile_hist_kernel<<<tiles, 256, 0, s>>>(b.keys, n, shift, mask, tiles, matrix);
        TC_LAUNCH_CHECK(ctx);
        scan32_reduce_kernel<<<nparts, 256, 0, s>>>(matrix, len, part);
        TC_LAUNCH_CHECK(ctx);
        scan32_spine_kernel<<<1, 1024, 0, s>>>(part, nparts);
        TC_LAUNCH_CHECK(ctx);
        scan32_down_kernel<<<nparts, 256, 0, s>>>(matrix, len, part, tiles, tile_offs);
        TC_LAUNCH_CHECK(ctx);
        radix_pass_kernel<GEN, false, 1, true><<<tiles, RDX_NT, 0, s>>>(
            b.keys, b.vals, b.keys_alt, b.vals_alt, n, shift, mask, bucket_base, b.status, ticket,
            ctx->d_err, tile_offs);
    } else if (variant == 2) {
        u32 grid = tc_persistent_grid_for(ctx, radix_pass_kernel<GEN, true, 8, false>, RDX_NT, 2);
        if (grid > tiles) grid = tiles;
        radix_pass_kernel<GEN, true, 8, false><<<grid, RDX_NT, 0, s>>>(
            b.keys, b.vals, b.keys_alt, b.vals_alt, n, shift, mask, bucket_base, b.status, ticket,
            ctx->d_err, nullptr);
    } else if (variant == 0) {
        radix_pass_kernel<GEN, false, 1, false><<<tiles, RDX_NT, 0, s>>>(
            b.keys, b.vals, b.keys_alt, b.vals_alt, n, shift, mask, bucket_base, b.status, ticket,
            ctx->d_err, nullptr);
    } else if (variant == 4) {
        radix_pass_kernel<GEN, false, 4, false><<<tiles, RDX_NT, 0, s>>>(
            b.keys, b.vals, b.keys_alt, b.vals_alt, n, shift, mask, bucket_base, b.status, ticket,
            ctx->d_err, nullptr);
    } else {
        radix_pass_kernel<GEN, false, 8, false><<<tiles, RDX_NT, 0, s>>>(
            b.keys, b.vals, b.keys_alt, b.vals_alt, n, shift, mask, bucket_base, b.status, ticket,
            ctx->d_err, nullptr);
    }
    TC_LAUNCH_CHECK(ctx);
}

void radix_sort_pairs(tc_ctx *ctx, RadixBuffers &b, u32 n, const RadixPlan &plan, bool gen_idx,
                      bool hist_ready, bool timed) {
    if (n == 0 || plan.npass == 0) return;
    RadixPlanDev pd;
    pd.npass = plan.npass;
    for (int p = 0; p < plan.npass; p++) {
        pd.shift[p] = plan.shift[p];
        pd.mask[p] = plan.mask[p];
    }
    hipStream_t s = ctx->stream;
    const int variant = radix_variant();
    if (!hist_ready && variant != 3) {
        TC_HIP(ctx, hipMemsetAsync(b.hist, 0, sizeof(u32) * RDX_MAX_PASSES * RDX_BINS, s));
        u32 grid = tc_cdiv(n, 256 * 16);
        if (grid > 2048) grid = 2048;
        radix_hist_kernel<<<grid, 256, 0, s>>>(b.keys, n, pd, b.hist);
        TC_LAUNCH_CHECK(ctx);
    }
    if (variant != 3) {
        radix_scan_hist_kernel<<<plan.npass, 256, 0, s>>>(b.hist);
        TC_LAUNCH_CHECK(ctx);
    }
    const u32 tiles = tc_cdiv(n, RDX_TILE);
    const size_t words = (size_t)tiles * RDX_BINS + 130;  // granules + 8 ticket counters, 128 B apart
    for (int p = 0; p < plan.npass; p++) {
        u32 *ticket = reinterpret_cast<u32 *>(b.status + (size_t)tiles * RDX_BINS);
        if (variant != 3) TC_HIP(ctx, hipMemsetAsync(b.status, 0, words * sizeof(u64), s));
        const bool ev = timed && ctx->profile && ctx->pev_used < 16;
        if (ev) TC_HIP(ctx, hipEventRecord(ctx->pev[2 * ctx->pev_used], s));
        if (gen_idx && p == 0)
            radix_launch_pass<true>(ctx, variant, b, n, plan.shift[p], plan.mask[p],
                                    b.hist + p * RDX_BINS, tiles, ticket);
        else
            radix_launch_pass<false>(ctx, variant, b, n, plan.shift[p], plan.mask[p],
                                     b.hist + p * RDX_BINS, tiles, ticket);
        if (ev) {
            TC_HIP(ctx, hipEventRecord(ctx->pev[2 * ctx->pev_used + 1], s));
            ctx->pev_used++;
        }
        u64 *tk = b.keys; b.keys = b.keys_alt; b.keys_alt = tk;
        u32 *tv = b.vals; b.vals = b.vals_alt; b.vals_alt = tv;
    }
}
