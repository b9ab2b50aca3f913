// tc_radix.hpp -- device LSD radix sort of (u64 key, u32 value) pairs.
//
// One "onesweep"-style pass = ONE read and ONE write of every pair: global digit
// histograms for all passes are taken up front (or fused into the producer of the
// keys), and each pass resolves a tile's position inside its digit bucket with a
// decoupled look-back over per-tile digit counts while the tile sits in LDS.
// Ranking inside the tile is stable: wave-striped items, match-any by ballots.
//
// Replaces (together with tc_sa.hip) `DS.unstableSortOn snd` over the suffixes,
// reference BWT/Internal.hs:130.
#pragma once
#include "tc_common.hpp"

#define RDX_BITS 8
#define RDX_BINS 256
#define RDX_MAX_PASSES 12
#define RDX_NT 512
#define RDX_ITEMS 8
#define RDX_TILE (RDX_NT * RDX_ITEMS)

struct RadixPlan {
    int npass = 0;
    int shift[RDX_MAX_PASSES];
    u32 mask[RDX_MAX_PASSES];
    void add_range(int lo_bit, int hi_bit) {  // passes over bits [lo_bit, hi_bit), LSD order
        for (int b = lo_bit; b < hi_bit; b += RDX_BITS) {
            int w = hi_bit - b < RDX_BITS ? hi_bit - b : RDX_BITS;
            shift[npass] = b;
            mask[npass] = (1u << w) - 1u;
            npass++;
        }
    }
};

struct RadixPlanDev {
    int npass;
    int shift[RDX_MAX_PASSES];
    u32 mask[RDX_MAX_PASSES];
};

#ifdef __HIPCC__

// ---- up-front digit histograms for every pass (one read of the keys) --------
__global__ __launch_bounds__(256) void radix_hist_kernel(const u64 *__restrict__ keys, u32 n,
                                                         RadixPlanDev plan,
                                                         u32 *__restrict__ hist) {
    __shared__ u32 s_h[RDX_MAX_PASSES * RDX_BINS];
    for (int i = threadIdx.x; i < plan.npass * RDX_BINS; i += 256) s_h[i] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        u64 k = keys[i];
        for (int p = 0; p < plan.npass; p++)
            atomicAdd(&s_h[p * RDX_BINS + (u32)((k >> plan.shift[p]) & plan.mask[p])], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < plan.npass * RDX_BINS; i += 256) {
        u32 c = s_h[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

// counts[npass][256] -> exclusive bucket bases, in place.  grid = npass blocks.
__global__ __launch_bounds__(256) void radix_scan_hist_kernel(u32 *hist) {
    __shared__ u32 s[8];
    u32 *h = hist + blockIdx.x * RDX_BINS;
    u32 v = h[threadIdx.x], tot;
    u32 e = block_excl_sum<256>(v, s, &tot);
    h[threadIdx.x] = e;
}

// ---- one pass ---------------------------------------------------------------
template <bool GEN_IDX>
__global__ __launch_bounds__(RDX_NT) void radix_pass_kernel(
    const u64 *__restrict__ kin, const u32 *__restrict__ vin, u64 *__restrict__ kout,
    u32 *__restrict__ vout, u32 n, int shift, u32 mask, const u32 *__restrict__ bucket_base,
    u64 *status, u32 *ticket, u32 *err) {
    constexpr int NW = RDX_NT / 64;
    __shared__ u64 s_keys[RDX_TILE];
    __shared__ u32 s_vals[RDX_TILE];
    __shared__ u32 s_hist[NW * RDX_BINS];
    __shared__ u32 s_dbase[RDX_BINS];
    __shared__ u32 s_gbase[RDX_BINS];
    __shared__ u32 s_scan[8];
    __shared__ u32 s_tile;

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = tid; i < NW * RDX_BINS; i += RDX_NT) s_hist[i] = 0;
    __syncthreads();
    const u32 tile = s_tile;
    const u64 base = (u64)tile * RDX_TILE;
    const u32 valid = (n - base) < (u64)RDX_TILE ? (u32)(n - base) : (u32)RDX_TILE;

    u64 key[RDX_ITEMS];
    u32 val[RDX_ITEMS];
    u32 rnk[RDX_ITEMS];
    const u32 wofs = w * 64 * RDX_ITEMS;
#pragma unroll
    for (int k = 0; k < RDX_ITEMS; k++) {
        u32 p = wofs + k * 64 + l;
        if (p < valid) {
            key[k] = kin[base + p];
            val[k] = GEN_IDX ? (u32)(base + p) : vin[base + p];
        } else {
            key[k] = ~0ull;
            val[k] = 0;
        }
    }
    // stable ranking inside the wave, item by item
    u32 *wh = s_hist + w * RDX_BINS;
#pragma unroll
    for (int k = 0; k < RDX_ITEMS; k++) {
        u32 p = wofs + k * 64 + l;
        u32 d = p < valid ? (u32)((key[k] >> shift) & mask) : 255u;  // pads: last digit, last
        u64 m = ~0ull;
#pragma unroll
        for (int b = 0; b < RDX_BITS; b++) {
            u64 bal = __ballot((d >> b) & 1u);
            m &= ((d >> b) & 1u) ? bal : ~bal;
        }
        u32 old = wh[d];
        u32 prior = __popcll(m & lanemask_lt());
        __builtin_amdgcn_wave_barrier();
        if (prior == 0) wh[d] = old + __popcll(m);
        __builtin_amdgcn_wave_barrier();
        rnk[k] = old + prior;
    }
    __syncthreads();
    // digit totals, exclusive over waves; one owner thread per digit
    u32 tot = 0;
    if (tid < RDX_BINS) {
#pragma unroll
        for (int i = 0; i < NW; i++) {
            u32 c = s_hist[i * RDX_BINS + tid];
            s_hist[i * RDX_BINS + tid] = tot;
            tot += c;
        }
    }
    u32 tot_real = tot;
    if (tid == 255) tot_real = tot - (RDX_TILE - valid);
    u64 *st = status + (u64)tile * RDX_BINS + tid;
    if (tid < RDX_BINS) lb_store(st, (tile == 0 ? LB_FLAG_INC : LB_FLAG_AGG) | (u64)tot_real);
    u32 dtot;
    u32 dbase = block_excl_sum<RDX_NT>(tid < RDX_BINS ? tot : 0u, s_scan, &dtot);
    if (tid < RDX_BINS) {
        u32 excl = 0;
        if (tile > 0) {
            i64 t = (i64)tile - 1;
            u32 spins = 0;
            while (true) {
                u64 s = lb_load(status + (u64)t * RDX_BINS + tid);
                u32 f = (u32)(s >> 62);
                if (f == 0) {
                    if (++spins > LB_SPIN_LIMIT) {
                        atomicOr(err, 2u);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                excl += (u32)LB_VALUE(s);
                if (f == 2) break;
                t--;
            }
            lb_store(st, LB_FLAG_INC | (u64)(excl + tot_real));
        }
        s_dbase[tid] = dbase;
        s_gbase[tid] = bucket_base[tid] + excl - dbase;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RDX_ITEMS; k++) {
        u32 p = wofs + k * 64 + l;
        u32 d = p < valid ? (u32)((key[k] >> shift) & mask) : 255u;
        u32 pos = s_dbase[d] + wh[d] + rnk[k];
        s_keys[pos] = key[k];
        s_vals[pos] = val[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RDX_ITEMS; k++) {
        u32 p = tid + k * RDX_NT;
        if (p < valid) {
            u64 kk = s_keys[p];
            u32 d = (u32)((kk >> shift) & mask);
            u32 g = s_gbase[d] + p;
            kout[g] = kk;
            vout[g] = s_vals[p];
        }
    }
}

#endif  // __HIPCC__

// Host driver.  Sorts n pairs by the plan's passes (stable, LSD).  Buffers ping-
// pong; on return keys/vals point at the sorted data and *_alt at scratch.
// `hist` (device, [npass][256] u32): if hist_ready the caller already filled it
// with digit COUNTS for every pass (fused into the key producer); else it is
// computed here.  `status` must hold tiles*256 u64 + 2 words.
struct RadixBuffers {
    u64 *keys, *keys_alt;
    u32 *vals, *vals_alt;
    u32 *hist;    // [RDX_MAX_PASSES][256]
    u64 *status;  // [tiles*256 + 2]
};
static inline size_t radix_status_words(u64 n) { return (size_t)tc_cdiv(n, RDX_TILE) * RDX_BINS + 2; }

void radix_sort_pairs(tc_ctx *ctx, RadixBuffers &b, u32 n, const RadixPlan &plan, bool gen_idx,
                      bool hist_ready, bool timed = false);
