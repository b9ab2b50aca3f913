// tc_radix_host.hpp -- host driver of the device radix sort (see tc_radix.hpp).
#pragma once
#include "tc_radix.hpp"

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
    const size_t words = (size_t)tiles * RDX_BINS + 2;
    for (int p = 0; p < plan.npass; p++) {
        TC_HIP(ctx, hipMemsetAsync(b.status, 0, words * sizeof(u64), s));
        u32 *ticket = reinterpret_cast<u32 *>(b.status + (size_t)tiles * RDX_BINS);
        const bool ev = timed && ctx->profile && ctx->pev_used < 16;
        if (ev) TC_HIP(ctx, hipEventRecord(ctx->pev[2 * ctx->pev_used], s));
        if (gen_idx && p == 0)
            radix_pass_kernel<true><<<tiles, RDX_NT, 0, s>>>(b.keys, b.vals, b.keys_alt,
                                                             b.vals_alt, n, plan.shift[p],
                                                             plan.mask[p], b.hist + p * RDX_BINS,
                                                             b.status, ticket, ctx->d_err);
        else
            radix_pass_kernel<false><<<tiles, RDX_NT, 0, s>>>(b.keys, b.vals, b.keys_alt,
                                                              b.vals_alt, n, plan.shift[p],
                                                              plan.mask[p], b.hist + p * RDX_BINS,
                                                              b.status, ticket, ctx->d_err);
        TC_LAUNCH_CHECK(ctx);
        if (ev) {
            TC_HIP(ctx, hipEventRecord(ctx->pev[2 * ctx->pev_used + 1], s));
            ctx->pev_used++;
        }
        u64 *tk = b.keys; b.keys = b.keys_alt; b.keys_alt = tk;
        u32 *tv = b.vals; b.vals = b.vals_alt; b.vals_alt = tv;
    }
}
