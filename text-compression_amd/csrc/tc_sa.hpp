// tc_sa.hpp -- suffix array + BWT last column on the device.
//
// Replaces createSuffixArray (reference BWT/Internal.hs:110-134: comparison sort
// of all n+1 suffixes incl. the empty one) and saToBWT (:98-106).  Algorithm:
//   round 0  k-mer sort: every suffix gets a 64-bit key = its first h0 symbols in
//            a dense base-(sigma+1) code ('$' = 0 < every symbol, so a proper
//            prefix sorts first, Q1) plus, in the low byte, the byte that precedes
//            it in the text; one LSD radix sort over the key's used bits.  The
//            low byte makes the last column fall out of the sorted keys with no
//            gather: L[j] = low byte of key[j].
//   round r  prefix doubling on the still-tied suffixes only: sort the active set
//            by (group, rank[i + h]), split groups, h doubles.
// All distinct suffixes have distinct (infinite) keys, so the result is the unique
// order the reference's sort produces, whatever its algorithm.
#pragma once
#include <type_traits>
#include "tc_radix.hpp"

#define SA_NT 256
#define SA_ITEMS 16
#define SA_TILE (SA_NT * SA_ITEMS)
#define KB_HALO 72  // >= P*s + s for every configuration (<= 7*8 + 8)

struct SaConfig {
    u32 sigma_text;  // distinct byte values present
    u32 B;           // sigma_text + 1 ('$' = code 0)
    u32 w;           // bits per field
    u32 s;           // symbols per field
    u32 P;           // fields sorted in round 0
    u32 h0;          // P * s
    double entropy;  // bits per symbol of the byte histogram
    u16 lut[256];    // byte -> code (1..sigma_text), 0 if absent
};

#ifdef __HIPCC__

// ---- byte histogram -----------------------------------------------------------
// One LDS counter per (byte value, lane): lanes never share an address, and the bank is
// the lane id, so a 5-letter text does not serialise 64 ways on 5 hot counters.
__global__ __launch_bounds__(256) void hist256_kernel(const u8 *__restrict__ text, u64 n,
                                                      u32 *__restrict__ counts) {
    __shared__ u32 s_h[256 * 64];
    for (int i = threadIdx.x; i < 256 * 64; i += 256) s_h[i] = 0;
    __syncthreads();
    u32 *h = s_h + (threadIdx.x & 63);
    const u64 nvec = ((uintptr_t)text & 15) ? 0 : n / 16;
    const uint4 *tv = reinterpret_cast<const uint4 *>(text);
    auto tally = [&](const uint4 v) {
        const u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            atomicAdd(&h[(x[q] & 255) * 64], 1u);
            atomicAdd(&h[((x[q] >> 8) & 255) * 64], 1u);
            atomicAdd(&h[((x[q] >> 16) & 255) * 64], 1u);
            atomicAdd(&h[(x[q] >> 24) * 64], 1u);
        }
    };
    // four loads in flight per thread (8 waves per CU, one load each, left the memory latency exposed)
    const u64 stride = (u64)gridDim.x * 256;
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < nvec; i += 4 * stride) {
        const uint4 v0 = tv[i], v1 = tv[i + stride], v2 = tv[i + 2 * stride], v3 = tv[i + 3 * stride];
        tally(v0); tally(v1); tally(v2); tally(v3);
    }
    for (; i < nvec; i += stride) tally(tv[i]);
    for (u64 i = nvec * 16 + (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        atomicAdd(&h[(u32)text[i] * 64], 1u);
    __syncthreads();
    u32 c = 0;
    for (int l = 0; l < 64; l++) c += s_h[threadIdx.x * 64 + ((l + threadIdx.x) & 63)];
    if (c) atomicAdd(&counts[threadIdx.x], c);
}

struct KeyBuildParams {
    u32 B, w, s, P;
    u16 lut[256];
    RadixPlanDev plan;
};

// ---- round-0 keys ---------------------------------------------------------------
// key(i) = fields G(i), G(i+s), ..  (G(p) = s symbols from p, base B) from bit 63
// down, w bits each; low byte = text[i-1] (0 for i == 0).  Text comes in through
// 16-byte loads into an LDS image of codes; G is built once per position.
// Digit histograms for the sort passes (saves one read of the keys):
//   ONEHIST (w == 8, every pass = one whole field): every pass sees the same multiset of
//   G values up to boundary terms, so ONE histogram H[v] = #{j < N : G(j) = v} is taken
//   here and keyhist_fix_kernel derives each pass's counts from it;
//   otherwise one LDS histogram per pass.
#define KB_PRE 16  // LDS slot 0 = text position base - KB_PRE (keeps 16-byte units aligned)
#define KB_SLOTS (SA_TILE + KB_PRE + KB_HALO + 8)

template <bool ONEHIST, int CS, int CP>  // CS/CP: compile-time s / P (0 = take them from kp)
__global__ __launch_bounds__(SA_NT) void keybuild_kernel(const u8 *__restrict__ text, u32 n,
                                                         KeyBuildParams kp,
                                                         u64 *__restrict__ keys,
                                                         u32 *__restrict__ hist) {
    __shared__ __attribute__((aligned(16))) u16 s_c[KB_SLOTS];  // codes
    __shared__ __attribute__((aligned(16))) u16 s_g[KB_SLOTS];  // G values
    __shared__ __attribute__((aligned(16))) u8 s_raw[KB_SLOTS]; // raw bytes
    __shared__ u32 s_h[ONEHIST ? RDX_BINS : RDX_MAX_PASSES * RDX_BINS];
    __shared__ u16 s_lut[256];
    const int tid = threadIdx.x;
    const u32 N = n + 1;
    const u64 base = (u64)blockIdx.x * SA_TILE;
    const u32 KS = CS ? (u32)CS : kp.s, KP = CP ? (u32)CP : kp.P;
    const int nh = ONEHIST ? RDX_BINS : kp.plan.npass * RDX_BINS;
    for (int i = tid; i < nh; i += SA_NT) s_h[i] = 0;
    s_lut[tid] = kp.lut[tid];
    __syncthreads();
    // phase A: text -> codes + raw, 16 bytes per thread per step
    const u32 span = SA_TILE + KP * KS + KS;                  // symbols needed from `base`
    const u32 units = (KB_PRE + span + 15) / 16;
    const bool aligned = (((uintptr_t)text) & 15) == 0;
    for (u32 u = tid; u < units; u += SA_NT) {
        const i64 p0 = (i64)base - KB_PRE + (i64)u * 16;       // text position of the unit's first byte
        u8 raw[16];
        if (aligned && p0 >= 0 && p0 + 16 <= (i64)n) {
            uint4 v = *reinterpret_cast<const uint4 *>(text + p0);
            u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 16; q++) raw[q] = (u8)(x[q >> 2] >> (8 * (q & 3)));
        } else {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                i64 p = p0 + q;
                raw[q] = (p >= 0 && p < (i64)n) ? text[p] : (u8)0;
            }
        }
        u16 code[16];
#pragma unroll
        for (int q = 0; q < 16; q++) {
            i64 p = p0 + q;
            code[q] = (p >= 0 && p < (i64)n) ? s_lut[raw[q]] : (u16)0;
        }
        uint4 *dc = reinterpret_cast<uint4 *>(s_c + u * 16);
        dc[0] = make_uint4(code[0] | (code[1] << 16), code[2] | (code[3] << 16), code[4] | (code[5] << 16), code[6] | (code[7] << 16));
        dc[1] = make_uint4(code[8] | (code[9] << 16), code[10] | (code[11] << 16), code[12] | (code[13] << 16), code[14] | (code[15] << 16));
        *reinterpret_cast<uint4 *>(s_raw + u * 16) =
            make_uint4(raw[0] | (raw[1] << 8) | (raw[2] << 16) | ((u32)raw[3] << 24),
                       raw[4] | (raw[5] << 8) | (raw[6] << 16) | ((u32)raw[7] << 24),
                       raw[8] | (raw[9] << 8) | (raw[10] << 16) | ((u32)raw[11] << 24),
                       raw[12] | (raw[13] << 8) | (raw[14] << 16) | ((u32)raw[15] << 24));
    }
    __syncthreads();
    // phase B: G(p) for every slot that a key of this tile can touch
    const u32 gslots = KB_PRE + SA_TILE + KP * KS;
    for (u32 q = tid; q < gslots; q += SA_NT) {
        u32 g = 0;
#pragma unroll
        for (u32 j = 0; j < KS; j++) g = g * kp.B + s_c[q + j];
        s_g[q] = (u16)g;
    }
    __syncthreads();
    // phase C: keys (striped => coalesced 8-byte stores) + histogram(s)
#pragma unroll 4
    for (int k = 0; k < SA_ITEMS; k++) {
        const u32 p = tid + k * SA_NT;
        const u64 i = base + p;
        if (i < N) {
            const u32 q = KB_PRE + p;
            u64 key = 0;
            int sh = 64;
#pragma unroll
            for (u32 f = 0; f < KP; f++) {
                sh -= kp.w;
                key |= (u64)s_g[q + f * KS] << sh;
            }
            key |= (u64)s_raw[q - 1];
            keys[i] = key;
            if (ONEHIST) {
                atomicAdd(&s_h[s_g[q]], 1u);
            } else {
                for (int pq = 0; pq < kp.plan.npass; pq++)
                    atomicAdd(&s_h[pq * RDX_BINS + (u32)((key >> kp.plan.shift[pq]) & kp.plan.mask[pq])], 1u);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nh; i += SA_NT) {
        u32 c = s_h[i];
        if (c) atomicAdd(&hist[ONEHIST ? (RDX_MAX_PASSES - 1) * RDX_BINS + i : i], c);
    }
}

// Histogram of G over all N positions, straight from the text (used when the first radix
// pass generates its keys itself, so no keybuild launch exists to take it).  Same
// conflict-free [value][lane] LDS layout as hist256_kernel.  Result in the LAST hist row.
template <int CS>  // CS: compile-time s (0 = runtime kp.s)
__global__ __launch_bounds__(256) void ghist_kernel(const u8 *__restrict__ text, u32 n,
                                                    KeyBuildParams kp, u32 *__restrict__ hist) {
    const u32 KS = CS ? (u32)CS : kp.s;
    __shared__ u32 s_h[256 * 64];
    __shared__ u16 s_lut[256];
    for (int i = threadIdx.x; i < 256 * 64; i += 256) s_h[i] = 0;
    s_lut[threadIdx.x] = kp.lut[threadIdx.x];
    __syncthreads();
    u32 *h = s_h + (threadIdx.x & 63);
    const u64 N = (u64)n + 1;
    const bool aligned = (((uintptr_t)text) & 15) == 0;
    const u64 units = (N + 15) / 16;
    for (u64 u = (u64)blockIdx.x * 256 + threadIdx.x; u < units; u += (u64)gridDim.x * 256) {
        const u64 p0 = u * 16;
        constexpr int NC = CS ? 16 + CS - 1 : 24;   // codes needed: 16 positions + (s - 1) look-ahead
        u32 c[NC];
        if (aligned && p0 + 32 <= n) {
            const uint4 *tv = reinterpret_cast<const uint4 *>(text + p0);
            uint4 a = tv[0];
            uint2 b = *reinterpret_cast<const uint2 *>(text + p0 + 16);
            u32 x[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
#pragma unroll
            for (int q = 0; q < NC; q++) c[q] = s_lut[(x[q >> 2] >> (8 * (q & 3))) & 255];
        } else {
#pragma unroll
            for (int q = 0; q < NC; q++) c[q] = (p0 + q < n) ? (u32)s_lut[text[p0 + q]] : 0u;
        }
#pragma unroll
        for (int j = 0; j < 16; j++) {
            if (p0 + j < N) {
                u32 g = 0;
#pragma unroll
                for (u32 t = 0; t < KS; t++) g = g * kp.B + c[j + t];   // s <= 8: j + t < 24
                atomicAdd(&h[g * 64], 1u);
            }
        }
    }
    __syncthreads();
    u32 cnt = 0;
    for (int l = 0; l < 64; l++) cnt += s_h[threadIdx.x * 64 + ((l + threadIdx.x) & 63)];
    if (cnt) atomicAdd(&hist[(RDX_MAX_PASSES - 1) * RDX_BINS + threadIdx.x], cnt);
}

// ONEHIST fix-up (one block): H sits in the LAST histogram row; pass q sorts field f_q,
// whose digit for suffix i is G(i + f_q*s), so
//   hist_q[v] = H[v] - #{j < f_q*s : G(j) = v} + #{N <= j < N + f_q*s : G(j) = v}.
__global__ __launch_bounds__(256) void keyhist_fix_kernel(const u8 *__restrict__ text, u32 n,
                                                          KeyBuildParams kp, u32 *hist) {
    __shared__ u32 s_g[128];
    const u32 N = n + 1;
    const int tid = threadIdx.x;
    // G at the head [0, maxoff) and just past the end [N, N + maxoff)
    if (tid < 128) {
        u32 j = tid < 64 ? (u32)tid : N + (u32)(tid - 64);
        u32 g = 0;
        for (u32 t = 0; t < kp.s; t++) {
            u64 p = (u64)j + t;
            g = g * kp.B + (p < n ? (u32)kp.lut[text[p]] : 0u);
        }
        s_g[tid] = g;
    }
    __syncthreads();
    const u32 H = hist[(RDX_MAX_PASSES - 1) * RDX_BINS + tid];
    for (int q = 0; q < kp.plan.npass; q++) {
        const u32 f = (u32)((64 - kp.plan.shift[q]) / 8 - 1);  // field sorted by pass q
        const u32 off = f * kp.s;
        u32 sub = 0, add = 0;
        for (u32 j = 0; j < off && j < 64; j++) {
            sub += s_g[j] == (u32)tid;
            add += s_g[64 + j] == (u32)tid;
        }
        hist[q * RDX_BINS + tid] = H - sub + add;
    }
}

// ---- group detection / re-ranking ---------------------------------------------
// One kernel, three uses (all over elements in sorted order):
//   INIT        element j sits at SA position j; tie <=> equal key bits above the
//               payload byte; writes L[j] = key low byte; compacts the members of
//               groups of size >= 2 into the first active set (slot, idx, group).
//   INIT + isa_only   (dense mode, second pass) rank[sa[j]] = group start, for all j.
//   INIT + isa        the same ranks written in the first pass (when dense mode is expected).
//   REFINE      element kk of the sorted active set goes to SA position slot[kk]; tie
//               <=> equal (group, rank[i+h]); writes SA, L (gathered), the new rank of
//               every member (dense: isa[idx]; sparse: t_rank[tpos]) and the next set.
// Flags are wave ballots, so every scan is bit arithmetic on wave-uniform 64-bit
// masks with scalar carries; only the cross-wave / cross-tile prefixes use LDS and a
// decoupled look-back (max of "last head", sum of actives).
#define GRP_NT 512
#ifndef GRP_ITEMS
#define GRP_ITEMS 8
#endif
#define GRP_TILE (GRP_NT * GRP_ITEMS)

struct GroupArgs {
    const u64 *keys;    // sorted keys
    u32 count;
    const u32 *vals;    // INIT: SA.  REFINE: for sorted position kk, index k0 into the in_* arrays
                        // (vals_are_idx: the suffix start itself, sorted along with the keys)
    int vals_are_idx;
    const u32 *in_slot; // REFINE: SA position of the kk-th active element (by sorted position)
    const u32 *in_idx;  // REFINE: suffix start, by k0
    const u32 *in_tpos; // REFINE sparse: rank-table position, by k0
    const u8 *text;
    u32 *sa;
    u32 *isa;           // dense rank array or null
    u64 *pairs;         // dense, large sets: (suffix start << 32 | rank) by sorted position instead of
                        // isa[start] = rank; rank_bin_kernel + rank_scatter_kernel apply them by regions
    u32 *t_rank;        // sparse rank table or null
    u8 *L;
    u32 *out_slot, *out_idx, *out_grp, *out_tpos;  // next active set
    int isa_only;
    int norank;         // REFINE, the key round (tc_encode_host.hpp): no rank array / table exists yet -- nothing to store, no table rows
    u64 *status_max, *status_sum;  // look-back granules, [tiles] each
    u32 *ticket;
    u64 *scalars;  // [1] active count
    u32 *err;
};

template <bool INIT>
__global__ __launch_bounds__(GRP_NT, 4) void group_kernel(GroupArgs a) {
    constexpr int NW = GRP_NT / 64;
    __shared__ u32 s_wmax[NW], s_wsum[NW];
    __shared__ u64 s_pref[2];
    __shared__ u32 s_tile;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const u64 KMASK = INIT ? ~0xffull : ~0ull;
    const u64 count = a.count;
    const u32 ntiles = (u32)((count + GRP_TILE - 1) / GRP_TILE);
    // persistent blocks, but every tile is drawn from the ticket counter when a block is
    // ready for it: a tile only ever waits on tiles already claimed by running blocks, so
    // no co-residency of the whole grid is assumed (other kernels may share the device)
    for (;;) {
    __syncthreads();
    if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
    __syncthreads();
    const u32 tile = s_tile;
    if (tile >= ntiles) break;
    const u64 base = (u64)tile * GRP_TILE + (u64)w * 64 * GRP_ITEMS;

    // ---- phase 1: head ballots, packed low bytes, wave aggregates -----------------
    u64 hb[GRP_ITEMS], ab[GRP_ITEMS];
    u32 lowb[GRP_ITEMS / 4];
#pragma unroll
    for (int q = 0; q < GRP_ITEMS / 4; q++) lowb[q] = 0;
    u64 carry = 0;
    bool has_prev = false;
    if (base > 0 && base < count) {
        carry = a.keys[base - 1] & KMASK;
        has_prev = true;
    }
    u64 raws[GRP_ITEMS];   // (all of a lane's keys first: one load per turn of the loop below was eight latencies in a row)
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; k++) {
        const u64 j = base + k * 64 + l;
        raws[k] = j < count ? a.keys[j] : 0;
    }
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; k++) {
        u64 j = base + k * 64 + l;
        bool in = j < count;
        u64 raw = raws[k];
        u64 key = raw & KMASK;
        if (INIT) lowb[k >> 2] |= (u32)(raw & 0xff) << (8 * (k & 3));
        u64 up = __shfl_up(key, 1, 64);
        u64 pk = (l == 0) ? carry : up;
        bool hp = (l == 0) ? has_prev : true;
        hb[k] = __ballot(in && (!hp || pk != key));
        carry = __shfl(key, 63, 64);
        has_prev = true;
    }
    u64 tail_next = 1;  // head flag of the element just past this wave's segment
    {
        u64 jn = base + (u64)64 * GRP_ITEMS;
        if (jn < count) tail_next = ((a.keys[jn] & KMASK) != carry) ? 1 : 0;
    }
    u32 wcnt = 0, wlast = 0;  // actives in this wave; 1 + value of the last head
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; k++) {
        u64 jb = base + (u64)k * 64;
        u64 inb = jb >= count ? 0ull : (count - jb >= 64 ? ~0ull : ((1ull << (count - jb)) - 1ull));
        u64 nxt0 = (k + 1 < GRP_ITEMS) ? (hb[(k + 1) % GRP_ITEMS] & 1ull) : tail_next;
        u64 nh = (hb[k] >> 1) | (nxt0 << 63);
        // the element after the last valid one is a (virtual) head
        u64 in_next = (inb >> 1) | ((jb + 64 < count ? 1ull : 0ull) << 63);
        nh |= ~in_next;
        ab[k] = inb & ~(hb[k] & nh);
        wcnt += (u32)__popcll(ab[k]);
        if (hb[k]) {
            u32 hl = 63u - (u32)__builtin_clzll(hb[k]);
            u64 jh = jb + hl;
            wlast = INIT ? (u32)jh + 1u : a.in_slot[jh] + 1u;
        }
    }
    if (l == 0) {
        s_wmax[w] = wlast;
        s_wsum[w] = wcnt;
    }
    __syncthreads();
    u32 pmax = 0, psum = 0, bmax = 0, bsum = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        u32 m = s_wmax[i], c = s_wsum[i];
        if (i < w) {
            pmax = pmax > m ? pmax : m;
            psum += c;
        }
        bmax = bmax > m ? bmax : m;
        bsum += c;
    }
    if (w == 0) {
        u64 e = lb_exclusive<OpMax>(a.status_max, tile, bmax, a.err);
        if (l == 0) s_pref[0] = e;
    } else if (w == 1) {
        u64 e = lb_exclusive<OpSum>(a.status_sum, tile, bsum, a.err);
        if (l == 0) {
            s_pref[1] = e;
            if ((u64)(tile + 1) * GRP_TILE >= count) a.scalars[1] = e + bsum;  // last tile
        }
    }
    __syncthreads();
    u32 cur_head = (u32)s_pref[0] > pmax ? (u32)s_pref[0] : pmax;  // 1 + value
    u32 cur_cnt = (u32)s_pref[1] + psum;

    // ---- phase 2: ranks, outputs, compaction -------------------------------------
    u32 k0s[INIT ? 1 : GRP_ITEMS], slots[INIT ? 1 : GRP_ITEMS];
    if (!INIT) {
#pragma unroll
        for (int k = 0; k < GRP_ITEMS; k++) {
            const u64 j = base + (u64)k * 64 + l;
            k0s[k] = j < count ? a.vals[j] : 0u;
            slots[k] = j < count ? a.in_slot[j] : 0u;
        }
    }
#pragma unroll
    for (int k = 0; k < GRP_ITEMS; k++) {
        const u64 jb = base + (u64)k * 64;
        if (jb >= count) break;
        const u64 j = jb + l;
        const bool in = j < count;
        const bool act = (ab[k] >> l) & 1ull;
        u32 k0 = 0, slot = 0;
        if (!INIT && in) {
            k0 = k0s[k];
            slot = slots[k];
        }
        // group start for this lane: last head at or before it (every lane executes
        // the shuffles; lanes without a head in range take the running scalar)
        const u64 mle = hb[k] & ((2ull << l) - 1ull);
        const u32 hl = mle ? 63u - (u32)__builtin_clzll(mle) : 0u;
        const u32 sv = INIT ? 0u : (u32)__shfl((int)slot, (int)hl, 64);
        const u32 g = mle ? (INIT ? (u32)(jb + hl) : sv) : cur_head - 1u;
        if (hb[k]) {
            u32 hlast = 63u - (u32)__builtin_clzll(hb[k]);
            cur_head = (INIT ? (u32)(jb + hlast) : (u32)__shfl((int)slot, (int)hlast, 64)) + 1u;
        }
        const u32 o = cur_cnt + (u32)__popcll(ab[k] & lanemask_lt());
        cur_cnt += (u32)__popcll(ab[k]);
        if (!in) continue;
        if (INIT) {
            if (a.isa_only) {
                if (a.pairs) a.pairs[j] = ((u64)a.vals[j] << 32) | g;
                else a.isa[a.vals[j]] = g;
            } else {
                a.L[j] = (u8)(lowb[k >> 2] >> (8 * (k & 3)));
                const u32 v = a.vals[j];
                if (a.pairs) a.pairs[j] = ((u64)v << 32) | g;
                else if (a.isa) a.isa[v] = g;   // dense mode expected: ranks in the same pass
                if (act) {
                    a.out_slot[o] = (u32)j;
                    a.out_idx[o] = v;
                    a.out_grp[o] = g;
                }
            }
        } else {
            const u32 i = a.vals_are_idx ? k0 : a.in_idx[k0];
            // nothing reads the SA during the rounds (ranks come from the rank array / the rank table and the sorted
            // keys): a member that stays tied is placed (SA, last column -- a random text byte) in the round that
            // resolves it, not in every round it lives through (round 4: also with the sparse rank table)
            if (!act) {
                a.sa[slot] = i;
                a.L[slot] = i ? a.text[i - 1] : (u8)0;
            }
            // a member of its old group's first subgroup keeps its rank (the rank of a tied member is the slot of its
            // group's head): nothing to store -- on text with long repeats that is almost every member, every round
            const bool moved = g != reinterpret_cast<const u32 *>(a.keys)[2 * j + 1];
            u32 tp = 0;
            if (a.pairs) a.pairs[j] = moved ? ((u64)i << 32) | g : ~0ull;
            else if (a.isa) { if (moved) a.isa[i] = g; }
            else if (!a.norank) {
                tp = a.in_tpos[k0];
                if (moved) a.t_rank[tp] = g;
            }
            if (act) {
                a.out_slot[o] = slot;
                a.out_idx[o] = i;
                a.out_grp[o] = g;
                if (!a.isa && !a.norank) a.out_tpos[o] = tp;
            }
        }
    }
    __syncthreads();  // LDS prefix slots are reused by the next tile
    }
}

// ---- dense ranks by regions ----------------------------------------------------------
// isa[start] = rank over a large set is one random 4-byte store per member: at 2^30 members ~48 ms,
// every store a 64-byte line of HBM traffic.  Instead group_kernel writes (start << 32 | rank) pairs in
// stream order; rank_bin_kernel partitions them by the top 8 bits of `start` (unstable -- order inside a
// region is irrelevant, the starts are distinct); rank_scatter_kernel then stores region after region,
// the workgroups running together covering a few regions of <= 16 MiB of ranks that stay in MALL until
// their lines are complete.  Measured (scripts/dbg/local_scatter_probe.py, 2^30): 48 ms -> 17-19 ms for
// the stores, plus the partition.
// Region d of the partitioned array starts at slot d << shift and never overflows: the starts are
// distinct, so at most 2^shift of them fall into one region (no counting pass).
#define RBIN_NT 1024
#define RBIN_ITEMS 16
#define RBIN_TILE (RBIN_NT * RBIN_ITEMS)

__global__ __launch_bounds__(256) void rank_cursor_kernel(u32 *cursor, int shift) {
    cursor[threadIdx.x] = (u32)threadIdx.x << shift;
}

__global__ __launch_bounds__(RBIN_NT) void rank_bin_kernel(const u64 *__restrict__ in, u32 m, int shift,
                                                           u32 *__restrict__ cursor, u64 *__restrict__ out, u64 nslots) {
    __shared__ u64 s_stage[RBIN_TILE];
    __shared__ u32 s_cnt[256], s_lb[256], s_gb[256];
    __shared__ u32 s_scan[RBIN_NT / 64 + 1];
    const u32 tid = threadIdx.x;
    const u64 base = (u64)blockIdx.x * RBIN_TILE;
    const u32 valid = (u64)m - base < (u64)RBIN_TILE ? (u32)((u64)m - base) : (u32)RBIN_TILE;
    if (tid < 256) s_cnt[tid] = 0;
    __syncthreads();
    u64 v[RBIN_ITEMS];
    u32 r[RBIN_ITEMS];
#pragma unroll
    for (int k = 0; k < RBIN_ITEMS; k++) {
        const u32 p = k * RBIN_NT + tid;
        v[k] = p < valid ? in[base + p] : 0ull;
    }
    // (a pair of all ones: a member whose rank did not change -- group_kernel<REFINE> -- is dropped here)
#pragma unroll
    for (int k = 0; k < RBIN_ITEMS; k++) {
        const u32 p = k * RBIN_NT + tid;
        if (p >= valid) v[k] = ~0ull;
        if (v[k] != ~0ull) r[k] = atomicAdd(&s_cnt[(u32)(v[k] >> 32) >> shift], 1u);
    }
    __syncthreads();
    const u32 c = tid < 256 ? s_cnt[tid] : 0u;
    u32 tot;
    const u32 lb = block_excl_sum<RBIN_NT>(c, s_scan, &tot);
    if (tid < 256) {
        s_lb[tid] = lb;
        s_gb[tid] = c ? atomicAdd(&cursor[tid], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RBIN_ITEMS; k++) {
        const u32 p = k * RBIN_NT + tid;
        if (v[k] != ~0ull) s_stage[s_lb[(u32)(v[k] >> 32) >> shift] + r[k]] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < RBIN_ITEMS; k++) {
        const u32 p = k * RBIN_NT + tid;
        if (p < tot) {
            const u64 x = s_stage[p];
            const u32 d = (u32)(x >> 32) >> shift;
            const u64 g = (u64)s_gb[d] + (p - s_lb[d]);
            if (g < nslots) out[g] = x;   // (always, while the starts are distinct; a guard against a caller's mistake)
        }
    }
}

// slots [0, nslots) of the partitioned array: slot p belongs to region p >> shift and is filled iff
// p < cursor[region]
#define RSCAT_NT 256
#define RSCAT_ITEMS 8
__global__ __launch_bounds__(RSCAT_NT) void rank_scatter_kernel(const u64 *__restrict__ part, u64 nslots, int shift,
                                                                 const u32 *__restrict__ cursor, u32 *__restrict__ isa) {
    const u64 base = (u64)blockIdx.x * (RSCAT_NT * RSCAT_ITEMS);
    u64 x[RSCAT_ITEMS];
    u32 cur[RSCAT_ITEMS];
    bool ok[RSCAT_ITEMS];
    // (the fill marks first, then the pairs: as one loop it was a mark load, a wait, a pair load, a wait -- eight times)
#pragma unroll
    for (int k = 0; k < RSCAT_ITEMS; k++) {
        const u64 p = base + (u64)k * RSCAT_NT + threadIdx.x;
        cur[k] = p < nslots ? cursor[p >> shift] : 0u;
    }
#pragma unroll
    for (int k = 0; k < RSCAT_ITEMS; k++) {
        const u64 p = base + (u64)k * RSCAT_NT + threadIdx.x;
        ok[k] = p < nslots && (u32)p < cur[k];
        x[k] = ok[k] ? part[p] : 0ull;
    }
#pragma unroll
    for (int k = 0; k < RSCAT_ITEMS; k++)
        if (ok[k]) isa[(u32)(x[k] >> 32)] = (u32)x[k];
}

// ---- finish: order the small buckets left by a partial (top-bits) sort -----------
// Input: pairs sorted (stably) by the key bits >= tshift only.  Elements that agree on
// those bits form a bucket; on high-entropy text buckets are tiny (1 GiB ACGTN, 32 top
// bits = 12 symbols: ~4 suffixes on average).  Each wave owns the buckets that START in
// its 64-position windows, ranks every member inside its bucket by the remaining key
// bits (all-pairs within the bucket, through a small per-wave LDS image), and writes
// SA / last column at the final positions.  Members whose whole key ties go to the
// active set (unordered; the host sorts it by slot).  A bucket that does not end within
// the next 64 positions raises `oversize`: the host then redoes the sort the long way.
#ifndef FIN_WPW
#define FIN_WPW 8      // windows per wave; measured at 1 GiB: 2: 7.75, 4: 6.96, 8: 6.59, 12: 7.99, 16: 7.89 ms
#endif
#ifndef FIN_NT
#define FIN_NT 256
#endif

// SA and last column of a text made of one repeated byte: SA = n, n-1, .., 0
__global__ __launch_bounds__(256) void unary_sa_kernel(const u8 *__restrict__ text, u32 n, u32 *__restrict__ sa,
                                                        u8 *__restrict__ L) {
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    if (j > n) return;
    sa[j] = n - (u32)j;
    L[j] = j < n ? text[0] : (u8)0;   // row n holds suffix 0: the sentinel slot (byte 0 by convention)
}

// Will the finish pass pay off?  SAMP_N pseudo-random suffixes, their top `topbits` key bits into an
// LDS hash set: *out = number of samples whose prefix was already present.  On text with heavy
// repeated contexts (natural language, long runs) a large share of the samples collide and the
// caller goes straight to the full path; random-like text gives ~0.
#define SAMP_N 8192
#define SAMP_SLOTS 16384
__global__ __launch_bounds__(1024) void sample_dup_kernel(const u8 *__restrict__ text, u32 n, RadixKeyGen kg,
                                                           int topbits, u32 *out) {
    __shared__ u32 table[SAMP_SLOTS];
    __shared__ u16 s_lut[256];
    __shared__ u32 s_dups;
    for (int i = threadIdx.x; i < SAMP_SLOTS; i += 1024) table[i] = 0xFFFFFFFFu;
    if (threadIdx.x < 256) s_lut[threadIdx.x] = kg.lut[threadIdx.x];
    if (threadIdx.x == 0) s_dups = 0;
    __syncthreads();
    u32 dups = 0;
    const u32 nf = ((u32)topbits + kg.w - 1) / kg.w;  // fields that reach into the top bits
    // a thread's eight samples: their text first -- four (unaligned) 32-bit loads each, all 32 issued together -- then the
    // keys (a byte load per symbol, each waited for, was ~100 latencies in a row: 120 us of a launch that does nothing else)
    constexpr int SPT = SAMP_N / 1024;
    const bool words = nf * kg.s <= 16;
    u32 tw[SPT][4];
    u64 qs[SPT];
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        const u32 k = threadIdx.x + 1024u * (u32)j;
        u64 z = (u64)(k + 1) * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 29)) * 0xBF58476D1CE4E5B9ull;
        qs[j] = (u64)(((z >> 32) * (u64)n) >> 32);
        const bool in = words && qs[j] + 16 <= n;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            u32 v = 0;
            if (in) __builtin_memcpy(&v, text + qs[j] + 4 * x, 4);
            tw[j][x] = v;
        }
    }
#pragma unroll
    for (int j = 0; j < SPT; j++) {
        u64 q = qs[j];
        const bool in = words && q + 16 <= n;
        u64 key = 0;
        int sh = 64;
        u32 pos = 0;
        for (u32 f = 0; f < nf; f++) {
            u32 g = 0;
            for (u32 t = 0; t < kg.s; t++, q++, pos++) {
                u32 c;
                if (in) {
                    u32 wsel = tw[j][0];   // (a select, not an indexed read: the words stay in registers)
                    wsel = (pos >> 2) == 1 ? tw[j][1] : wsel;
                    wsel = (pos >> 2) == 2 ? tw[j][2] : wsel;
                    wsel = (pos >> 2) == 3 ? tw[j][3] : wsel;
                    c = (u32)s_lut[(wsel >> (8 * (pos & 3))) & 255u];
                } else {
                    c = q < n ? (u32)s_lut[text[q]] : 0u;
                }
                g = g * kg.B + c;
            }
            sh -= kg.w;
            key |= (u64)g << sh;
        }
        u64 kt = key >> (64 - topbits);
        if (topbits > 32) kt = (kt ^ (kt >> 31)) * 0xD6E8FEB86659FD93ull >> 32;   // fold to 32 bits
        const u32 v = (u32)kt;
        if (v == 0xFFFFFFFFu) continue;
        u32 slot = (v * 0x9E3779B1u) >> 18;  // 14 bits
        for (int probe = 0; probe < SAMP_SLOTS; probe++) {
            const u32 old = atomicCAS(&table[slot], 0xFFFFFFFFu, v);
            if (old == 0xFFFFFFFFu) break;
            if (old == v) { dups++; break; }
            slot = (slot + 1) & (SAMP_SLOTS - 1);
        }
    }
    dups = wave_sum(dups);
    if (lane_id() == 0 && dups) atomicAdd(&s_dups, dups);
    __syncthreads();
    if (threadIdx.x == 0) *out = s_dups;
}

struct FinishArgs {
    const u64 *keys;
    const u32 *sa_in;
    u32 N;
    int tshift;        // bucket id = key >> tshift
    int lshift, lbits; // remaining key bits = (key >> lshift) & ((1 << lbits) - 1), lbits <= 32
    u32 *sa_out;
    u8 *L;
    u32 *out_slot, *out_idx, *out_grp;
    u32 act_cap;
    u32 *counters;     // [0] active count, [1] bit 0: a bucket longer than the window logic handles was
                       //     met (what was written for its members is void), bit 1: such buckets were emitted as
                       //     tied groups
    u32 *rcount;       // finish_kernel: FIN_REGIONS counters, FIN_RSTRIDE words apart; block b appends to region
                       //     b % FIN_REGIONS of out_* (rcap entries each) -- one counter for the whole grid is a
                       //     12 ns serial step per block on repeat-rich input
    u32 rcap;
    u32 fix_cap;       // finish_fix_kernel stops writing once the active count exceeds this
    u32 *ovbits;       // finish_fix_kernel: one bit per position, set for members of over-long buckets
};

// tied members are staged per wave and leave with ONE counter update per flush: on inputs with ties in
// most windows (repeats) a device-scope atomic per window on one address serialises the grid
// (genome-like 256 MiB: 6.6 ms, of which ~5 ms waiting for the counter)
#define FIN_REGIONS 64
#define FIN_RSTRIDE 64
#ifndef FIN_STAGE
#define FIN_STAGE 192   // staged entries per wave (a window adds at most 128)
#endif
__global__ __launch_bounds__(FIN_NT) void finish_kernel(FinishArgs a) {
    __shared__ u32 s_low[FIN_NT / 64][128];
#if FIN_STAGE
    __shared__ u32 s_stg[FIN_NT / 64][3][FIN_STAGE + 128];
#endif
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    u32 *low = s_low[w];
#if FIN_STAGE
    __shared__ u32 s_nstg[FIN_NT / 64];
    __shared__ u32 s_base;
    u32 nstg = 0;   // wave-uniform
    // have_base: the slots were obtained for the whole block (end of the kernel)
    const u32 region = blockIdx.x % FIN_REGIONS;
    u32 *rctr = a.rcount + region * FIN_RSTRIDE;
    const u32 rbase = region * a.rcap;
    auto flush = [&](bool have_base, u32 base) {   // base: offset inside the block's region
        if (!have_base) {
            if (l == 0) base = atomicAdd(rctr, nstg);
            base = __shfl(base, 0, 64);
        }
        __builtin_amdgcn_wave_barrier();
        for (u32 e = l; e < nstg; e += 64) {
            const u32 o = rbase + base + e;
            if (base + e < a.rcap) {
                a.out_slot[o] = s_stg[w][0][e];
                a.out_idx[o] = s_stg[w][1][e];
                a.out_grp[o] = s_stg[w][2][e];
            }
        }
        __builtin_amdgcn_wave_barrier();
        nstg = 0;
    };
#endif
    const u64 N = a.N;
    const u64 wave = (u64)blockIdx.x * (FIN_NT / 64) + w;
    const u64 ws0 = wave * 64 * FIN_WPW;
#if !FIN_STAGE
#error "finish_kernel needs the staged hand-over (FIN_STAGE > 0)"
#endif
    // (staged: a wave past the end runs through -- every window returns at once -- to reach the
    // block-wide hand-over of the staged entries)
    const u64 lowmask = a.lbits >= 64 ? ~0ull : ((1ull << a.lbits) - 1ull);  // remaining key bits

    // all chunks of this wave are requested up front (one round trip instead of one per window)
    u64 kk[FIN_WPW + 1];
    u32 vv[FIN_WPW + 1];
    // interior waves (all but the last few): no range tests anywhere below
    const bool interior = ws0 + (u64)(FIN_WPW + 1) * 64 <= N;
    if (interior) {
#pragma unroll
        for (int c = 0; c <= FIN_WPW; c++) {
            const u64 pos = ws0 + (u64)c * 64 + l;
            kk[c] = a.keys[pos];
            vv[c] = a.sa_in[pos];
        }
    } else {
#pragma unroll
        for (int c = 0; c <= FIN_WPW; c++) {
            const u64 pos = ws0 + (u64)c * 64 + l;
            kk[c] = pos < N ? a.keys[pos] : ~0ull;
            vv[c] = pos < N ? a.sa_in[pos] : 0u;
        }
    }
    u64 tprev = (ws0 > 0 && ws0 <= N) ? (a.keys[ws0 - 1] >> a.tshift) : 0;  // wave-uniform
    bool hasprev = ws0 > 0;

    auto window = [&](auto full_tag, const int win) {
        constexpr bool FULL = decltype(full_tag)::value;
        const u64 ws = ws0 + (u64)win * 64;
        if (!FULL && ws >= N) return;
        const u64 kA = kk[win], kB = kk[win + 1];
        const u32 vA = vv[win], vB = vv[win + 1];
        const bool inA = FULL || ws + l < N, inB = FULL || ws + 64 + l < N;
        const u64 tA = kA >> a.tshift, tB = kB >> a.tshift;
        // head flags
        u64 upA = __shfl_up(tA, 1, 64);
        u64 lastA = __shfl(tA, 63, 64);
        u64 upB = __shfl_up(tB, 1, 64);
        bool hA = inA && ((l == 0) ? (!hasprev || tA != tprev) : (tA != upA));
        bool hB = inB && ((l == 0) ? (tB != lastA) : (tB != upB));
        const u64 hbA = __ballot(hA), hbB = __ballot(hB);
        const u64 inbA = __ballot(inA), inbB = __ballot(inB);
        // ownership: A-lanes at or after the first head of A; B-lanes before the first head of B
        // (they continue A's last bucket) -- provided A has a head at all
        const int a0 = hbA ? __builtin_ctzll(hbA) : 64;
        const int b0 = hbB ? __builtin_ctzll(hbB) : 64;
        // the bucket running out of A must end inside B (a head in B, or the array ends there)
        if (hbA && !hbB && (~inbB) == 0ull) {
            // no head in a full B: the last bucket of A is longer than this kernel handles.  What
            // is written for its members below is meaningless; finish_fix_kernel redoes them
            // (and finish_filter_kernel drops their entries from the tied list).
            if (l == 0 && !(__hip_atomic_load(&a.counters[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1u))
                atomicOr(&a.counters[1], 1u);
        }
        const bool ownA = inA && l >= a0;
        const bool ownB = inB && hbA && l < b0;
        // bucket [s, t) in combined coordinates c = 0..127
        u64 mleA = hbA & ((2ull << l) - 1ull);
        int sA = mleA ? 63 - __builtin_clzll(mleA) : 0;            // start of my bucket (A lanes)
        u64 mgtA = (l == 63) ? 0ull : (hbA & ~((2ull << l) - 1ull)); // heads after me in A
        int endA_in_A = mgtA ? __builtin_ctzll(mgtA) : -1;
        int lastHeadA = hbA ? 63 - __builtin_clzll(hbA) : 0;
        // end of A's last bucket: first head of B, else the end of the array (valid
        // positions are a prefix of the 128)
        int endB = b0 < 64 ? 64 + b0 : (int)(__popcll(inbA) + __popcll(inbB));
        int tAend = endA_in_A >= 0 ? endA_in_A : endB;
        // A lanes: bucket [sA, tAend); B lanes (owned): bucket [lastHeadA, endB)
        const u32 lowA = (u32)((kA >> a.lshift) & lowmask), lowB = (u32)((kB >> a.lshift) & lowmask);
        low[l] = lowA;
        low[64 + l] = lowB;
        __builtin_amdgcn_wave_barrier();
        // rank inside the bucket: walk the bucket's members once (stable: ties broken by
        // position).  A-lanes walk their own bucket; the B-lanes that continue A's last
        // bucket all share ONE bucket and are handled by a second, usually short, walk.
        const int cA = l, cB = 64 + l;
        const int sB = lastHeadA, tBend = endB;
        int lenA = ownA ? tAend - sA : 0;
        int maxlen = lenA;
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            int o = __shfl_xor(maxlen, d, 64);
            maxlen = maxlen > o ? maxlen : o;
        }
        u32 rankA = 0, ltA = 0, leA = 0;
        for (int t = 0; t < maxlen; t++) {
            const int u = sA + t;
            if (ownA && u < tAend) {
                const u32 y = low[u];
                ltA += y < lowA;
                leA += y <= lowA;
                rankA += (y < lowA) | ((y == lowA) & (u < cA));
            }
        }
        const u32 eqA = ownA ? leA - ltA - 1u : 0u;  // other members with the same remaining bits
        u32 rankB = 0, ltB = 0, leB = 0;
        if (hbA && b0 > 0) {  // wave-uniform: some B-lanes continue A's last bucket
            for (int u = sB; u < tBend; u++) {
                const u32 y = low[u];
                ltB += y < lowB;
                leB += y <= lowB;
                rankB += (y < lowB) | ((y == lowB) & (u < cB));
            }
        }
        const u32 eqB = ownB ? leB - ltB - 1u : 0u;
        __builtin_amdgcn_wave_barrier();
        // outputs
        const bool actA = ownA && eqA > 0, actB = ownB && eqB > 0;
        const u64 abA = __ballot(actA), abB = __ballot(actB);
        const u32 nact = (u32)__popcll(abA) + (u32)__popcll(abB);
#if FIN_STAGE
        if (nact) {
            if (nstg > FIN_STAGE) flush(false, 0u);
            if (actA) {
                const u32 e = nstg + (u32)__popcll(abA & lanemask_lt());
                s_stg[w][0][e] = (u32)(ws + sA) + rankA;
                s_stg[w][1][e] = vA;
                s_stg[w][2][e] = (u32)(ws + sA) + ltA;
            }
            if (actB) {
                const u32 e = nstg + (u32)__popcll(abA) + (u32)__popcll(abB & lanemask_lt());
                s_stg[w][0][e] = (u32)(ws + sB) + rankB;
                s_stg[w][1][e] = vB;
                s_stg[w][2][e] = (u32)(ws + sB) + ltB;
            }
            nstg += nact;
        }
        if (ownA) {
            u32 j = (u32)(ws + sA) + rankA;
            a.sa_out[j] = vA;
            a.L[j] = (u8)(kA & 0xff);
        }
        if (ownB) {
            u32 j = (u32)(ws + sB) + rankB;
            a.sa_out[j] = vB;
            a.L[j] = (u8)(kB & 0xff);
        }
#else
        u32 abase = 0;
        if (nact) {
            if (l == 0) abase = atomicAdd(&a.counters[0], nact);
            abase = __shfl(abase, 0, 64);
        }
        if (ownA) {
            u32 j = (u32)(ws + sA) + rankA;
            a.sa_out[j] = vA;
            a.L[j] = (u8)(kA & 0xff);
            if (actA) {
                u32 o = abase + (u32)__popcll(abA & lanemask_lt());
                if (o < a.act_cap) {
                    a.out_slot[o] = j;
                    a.out_idx[o] = vA;
                    a.out_grp[o] = (u32)(ws + sA) + ltA;
                }
            }
        }
        if (ownB) {
            u32 j = (u32)(ws + sB) + rankB;
            a.sa_out[j] = vB;
            a.L[j] = (u8)(kB & 0xff);
            if (actB) {
                u32 o = abase + (u32)__popcll(abA) + (u32)__popcll(abB & lanemask_lt());
                if (o < a.act_cap) {
                    a.out_slot[o] = j;
                    a.out_idx[o] = vB;
                    a.out_grp[o] = (u32)(ws + sB) + ltB;
                }
            }
        }
#endif
        tprev = lastA;
        hasprev = true;
    };
    if (interior) {
#pragma unroll
        for (int win = 0; win < FIN_WPW; win++) window(std::true_type{}, win);
    } else {
#pragma unroll
        for (int win = 0; win < FIN_WPW; win++) window(std::false_type{}, win);
    }
#if FIN_STAGE
    // one counter update per BLOCK: a device-scope atomic on one address is the slowest thing here
    // when most waves hold ties (genome-like input)
    if (l == 0) s_nstg[w] = nstg;
    __syncthreads();
    u32 tot = 0, before = 0;
#pragma unroll
    for (int i = 0; i < FIN_NT / 64; i++) {
        if (i < w) before += s_nstg[i];
        tot += s_nstg[i];
    }
    if (tot == 0) return;
    if (threadIdx.x == 0) s_base = atomicAdd(rctr, tot);
    __syncthreads();
    if (nstg) flush(true, s_base + before);
#endif
}

// Second pass, launched only when the lean pass met a bucket longer than it handles.  Members of
// such a bucket -- the last bucket of a window when it does not end in the next one, and the
// leading lanes of a window when the previous window holds no head of their bucket (this covers
// windows lying entirely inside a bucket) -- become ONE tied group per bucket: group = bucket
// start, members stay in place; the doubling rounds order them.  Their slots are marked in
// `ovbits` (one bit per position) so that the entries the lean pass produced for them can be
// dropped.  A wave covers FIX_WIN windows: it first counts its members (one atomic per wave;
// nothing is written once the total exceeds fix_cap -- the caller then takes the full path),
// then emits.
#define FIX_WIN 16
__global__ __launch_bounds__(FIN_NT) void finish_fix_kernel(FinishArgs a) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const u64 N = a.N;
    const u64 wave = (u64)blockIdx.x * (FIN_NT / 64) + w;
    const u64 ws0 = wave * 64 * FIX_WIN;
    if (ws0 >= N) return;
    // the wave's FIX_WIN + 1 windows of keys by loads issued together, kept in LDS for both phases (one load per window and
    // loop turn, each waited for, twice: the pass ran at 1.4 TB/s -- 6.1 ms per GiB of repeat-rich DNA)
    __shared__ u64 s_k[FIN_NT / 64][(FIX_WIN + 1) * 64];
    u64 tprev0, tprev65;
    {
        u64 kw[FIX_WIN + 1];
#pragma unroll
        for (int i = 0; i <= FIX_WIN; i++) {
            const u64 p = ws0 + (u64)i * 64 + l;
            kw[i] = p < N ? a.keys[p] : ~0ull;
        }
        tprev0 = ws0 > 0 ? a.keys[ws0 - 1] : 0;
        tprev65 = ws0 >= 65 ? a.keys[ws0 - 65] : 0;
#pragma unroll
        for (int i = 0; i <= FIX_WIN; i++) s_k[w][i * 64 + l] = kw[i];
    }
    wave_fence();
    u32 abase = 0;
    for (int phase = 0; phase < 2; phase++) {
        u32 total = 0;
        u64 tprev = ws0 > 0 ? (tprev0 >> a.tshift) : 0;
        // the previous window holds a bucket head unless its 64 keys and the key before them all
        // share their top bits (sorted keys)
        bool prev_head = ws0 < 65 || (tprev65 >> a.tshift) != tprev;
        u64 kA = s_k[w][l];
        for (int win = 0; win < FIX_WIN; win++) {
            const u64 ws = ws0 + (u64)win * 64;
            if (ws >= N) break;
            const bool inA = ws + l < N, inB = ws + 64 + l < N;
            const u64 kB = s_k[w][(win + 1) * 64 + l];
            const u64 tA = kA >> a.tshift, tB = kB >> a.tshift;
            u64 upA = __shfl_up(tA, 1, 64);
            u64 lastA = __shfl(tA, 63, 64);
            u64 upB = __shfl_up(tB, 1, 64);
            bool hA = inA && ((l == 0) ? (ws == 0 || tA != tprev) : (tA != upA));
            bool hB = inB && ((l == 0) ? (tB != lastA) : (tB != upB));
            const u64 hbA = __ballot(hA), hbB = __ballot(hB);
            const u64 inbA = __ballot(inA), inbB = __ballot(inB);
            const int a0 = hbA ? __builtin_ctzll(hbA) : 64;
            const int lastHeadA = hbA ? 63 - __builtin_clzll(hbA) : 0;
            const bool lastOver = hbA && !hbB && (~inbB) == 0ull;
            // leading lanes were ranked by the previous window iff it holds their bucket's head
            // and the bucket ends in here
            const bool lead_ok = prev_head && (hbA != 0ull || (~inbA) != 0ull);
            const bool ovLead = inA && l < a0 && !lead_ok;
            const bool ovLast = lastOver && inA && l >= lastHeadA;
            const u64 abO = __ballot(ovLead || ovLast);
            if (abO && phase == 1) {
                if (l == 0) reinterpret_cast<u64 *>(a.ovbits)[ws >> 6] = abO;
                u32 bstart = 0;  // start of the bucket the leading lanes continue
                if (__ballot(ovLead)) {
                    // first index whose top bits are >= those of lane 0's key: 64-ary search below ws
                    const u64 t0 = __shfl(tA, 0, 64);
                    u64 lo = 0, hi = ws;
                    while (hi - lo >= 64) {  // the last step needs a lane past hi (answer == hi)
                        const u64 step = (hi - lo) / 64;
                        const bool ge = (a.keys[lo + step * l] >> a.tshift) >= t0;
                        const u64 mge = __ballot(ge);
                        if (mge == 0) {
                            lo = lo + step * 63 + 1;
                        } else {
                            const int f = __builtin_ctzll(mge);
                            hi = lo + step * f;
                            if (f > 0) lo = lo + step * (f - 1) + 1;
                        }
                    }
                    const bool ge = lo + l < hi ? (a.keys[lo + l] >> a.tshift) >= t0 : true;
                    bstart = (u32)(lo + __builtin_ctzll(__ballot(ge)));
                }
                if (ovLead || ovLast) {
                    const u32 j = (u32)ws + (u32)l;
                    const u32 v = a.sa_in[j];
                    a.sa_out[j] = v;
                    a.L[j] = (u8)(kA & 0xff);
                    const u32 o = abase + total + (u32)__popcll(abO & lanemask_lt());
                    if (o < a.act_cap) {
                        a.out_slot[o] = j;
                        a.out_idx[o] = v;
                        a.out_grp[o] = ovLead ? bstart : (u32)ws + (u32)lastHeadA;
                    }
                }
            }
            total += (u32)__popcll(abO);
            tprev = lastA;
            prev_head = hbA != 0ull;
            kA = kB;
        }
        if (phase == 0) {
            if (total == 0) return;
            if (l == 0) {
                abase = atomicAdd(&a.counters[0], total);
                atomicOr(&a.counters[1], 2u);
            }
            abase = __shfl(abase, 0, 64);
            if ((u64)abase + total > (u64)a.fix_cap) return;  // too many: the caller falls back
        }
    }
}

// entries of the lean pass whose slot belongs to an over-long bucket are void: slot := ~0 (they
// sort behind every real entry); *ndropped counts them
__global__ __launch_bounds__(256) void finish_filter_kernel(u32 *__restrict__ slot, u32 count,
                                                            const u32 *__restrict__ ovbits, u32 *ndropped) {
    // grid-stride, one counter update per block (an update per wave on the one address took longer
    // than the pass itself on repeat-rich input)
    __shared__ u32 s_n;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    u32 mine = 0;
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < count; k += (u64)gridDim.x * 256) {
        const u32 j = slot[k];
        if ((ovbits[j >> 5] >> (j & 31)) & 1u) {
            slot[k] = 0xFFFFFFFFu;
            mine++;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) mine += (u32)__shfl_xor((int)mine, d, 64);
    if (lane_id() == 0 && mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(ndropped, s_n);
}

// the regions of finish_kernel -> one contiguous list.  roff[r] = entries before region r; counters[0]
// = their total (what the host reads); a region that overflowed makes the total exceed every capacity.
__global__ __launch_bounds__(64) void finish_regions_kernel(const u32 *__restrict__ rcount, u32 rcap,
                                                            u32 *__restrict__ roff, u32 *__restrict__ counters) {
    const u32 l = threadIdx.x;
    const u32 c = rcount[l * FIN_RSTRIDE];
    const bool over = c > rcap;
    u32 inc = over ? rcap : c;
    const u32 mine = inc;
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = (u32)__shfl_up((int)inc, d, 64);
        if ((int)l >= d) inc += t;
    }
    roff[l] = inc - mine;
    const u64 anyover = __ballot(over);
    if (l == 63) {
        roff[64] = inc;
        counters[0] = anyover ? 0xFFFFFFF0u : inc;
    }
}
__global__ __launch_bounds__(256) void finish_compact_kernel(const u32 *__restrict__ roff, u32 rcap,
                                                             const u32 *__restrict__ in_slot,
                                                             const u32 *__restrict__ in_idx,
                                                             const u32 *__restrict__ in_grp,
                                                             u32 *__restrict__ out_slot, u32 *__restrict__ out_idx,
                                                             u32 *__restrict__ out_grp,
                                                             const u32 *__restrict__ in_x = nullptr, u32 *__restrict__ out_x = nullptr,
                                                             u32 x_cap = 0) {
    const u32 total = roff[64];
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < total; k += (u64)gridDim.x * 256) {
        u32 lo = 0, hi = FIN_REGIONS;   // the region whose range holds entry k
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (roff[mid] <= (u32)k) lo = mid; else hi = mid;
        }
        const u64 src = (u64)lo * rcap + ((u32)k - roff[lo]);
        out_slot[k] = in_slot[src];
        out_idx[k] = in_idx[src];
        out_grp[k] = in_grp[src];
        if (in_x && k < x_cap) out_x[k] = in_x[src];   // (key-only levels: the upper key bits of a tied member)
    }
}

// ---- key-only MSD levels (tc_msd.hpp, VALS = false): the suffix starts of the tied members, found again ------
// The levels moved keys only, so a member of the tied set is known by its slot, its group and its KEY -- the P * s
// symbols it shares with the other members of its group.  Its suffix start is some position of the text where
// those symbols occur, and ANY assignment of a group's positions to the group's slots will do: the members are
// tied, the doubling rounds order them.  So: the keys go into a hash table (value = the group's first slot, a
// counter), one pass over the text computes every position's P * s-symbol value by a rolling base-B number and
// probes -- through a Bloom filter in LDS, so that a position that is not tied (all but a few thousand of 2^30)
// costs one LDS bit test -- and every hit takes the next slot of its group.  A group of c members occurs exactly
// c times in the text, so the pass ends with exactly m entries (the caller checks that).
#define TP_SLOT_BITS 17
#define TP_BLOOM_LOG2 18
#define TP_MAX_TIED (1u << 15)
struct TiedTable {
    u64 *key;      // [2^TP_SLOT_BITS] dense value of the key's symbols (never 0 for a tied suffix); 0 = empty
    u32 *grp;      // first slot of the group
    u32 *cnt;      // members found so far
    u32 *bloom;    // [2^TP_BLOOM_LOG2 / 32]
};
__device__ __forceinline__ u64 tp_mix(u64 v) { return v * 0x9E3779B97F4A7C15ull; }
// The filter's hash is a cyclic polynomial over the symbol codes (rotate-and-xor: three cheap 32-bit operations
// per position of the text to roll it), the table's key is the exact base-B value of the h symbols, computed
// only for the positions that pass the filter.
__device__ __forceinline__ u32 tp_sym(u32 c) { return __umul24(c, 0x9E3779u); }   // (one full-rate multiply; code 0 -> 0)
__device__ __forceinline__ u32 tp_rotl(u32 x, u32 r) { return __builtin_amdgcn_alignbit(x, x, (32u - r) & 31u); }

// members (klo = key bits 8..39, khi = key bits 40..63, grp) -> table.  A field is an s-digit base-B number.
__global__ __launch_bounds__(256) void tied_table_kernel(const u32 *__restrict__ klo, const u32 *__restrict__ khi,
                                                         const u32 *__restrict__ grp, u32 m, u32 B, u32 sdig, u32 P,
                                                         TiedTable T) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    const u64 key56 = ((u64)khi[k] << 32) | klo[k];
    u64 v = 0;
    u32 H = 0;
    for (u32 f = 0; f < P; f++) {
        const u32 g = (u32)(key56 >> (48 - 8 * f)) & 255u;
        u32 div = 1;
        for (u32 j = 1; j < sdig; j++) div *= B;
        for (u32 j = 0; j < sdig; j++, div /= B) {
            const u32 c = (g / div) % B;
            v = v * B + c;
            H = tp_rotl(H, 1) ^ tp_sym(c);
        }
    }
    const u32 hb = H >> (32 - TP_BLOOM_LOG2);
    atomicOr(&T.bloom[hb >> 5], 1u << (hb & 31));
    u32 slot = (u32)(tp_mix(v) >> (64 - TP_SLOT_BITS));
    for (u32 probe = 0; probe < (1u << TP_SLOT_BITS); probe++) {
        const u64 old = atomicCAS((unsigned long long *)&T.key[slot], 0ull, (unsigned long long)v);
        if (old == 0ull || old == v) {
            T.grp[slot] = grp[k];   // (the same value from every member of the group)
            return;
        }
        slot = (slot + 1) & ((1u << TP_SLOT_BITS) - 1u);
    }
}

#define TPK_NT 256
#define TPK_PER 64
#define TPK_TILE (TPK_NT * TPK_PER)
#define TPK_PAD(p) ((p) + ((p) >> 6) * 4u)     // a thread's 64 codes start 68 bytes apart: conflict-free reads
// S = symbols per field (the key has P = 7 fields): the window of 64 + 7 S codes a thread rolls over sits in
// registers, every byte selection is a constant -- ~12 VALU and one LDS bit test per position of the text
template <int S>
__global__ __launch_bounds__(TPK_NT) void tied_probe_kernel(const u8 *__restrict__ text, u32 n, RadixKeyGen kg,
                                                            TiedTable T, u32 *__restrict__ out_slot,
                                                            u32 *__restrict__ out_idx, u32 *__restrict__ out_grp,
                                                            u32 *total, u32 cap) {
    constexpr u32 h = 7 * S;                       // symbols per key
    constexpr int WD = (64 + h + 3) / 4;           // dwords of the window
    static_assert(h <= 56 && WD <= 30, "the window is 16 own dwords + at most 14 of the next thread's");
    __shared__ u32 s_bloom[(1u << TP_BLOOM_LOG2) / 32];
    __shared__ __attribute__((aligned(16))) u8 s_code[TPK_PAD(TPK_TILE + 64) + 16];
    __shared__ u8 s_lut[256];
    const u32 tid = threadIdx.x;
    const u32 B = kg.B;
    for (u32 i = tid; i < (1u << TP_BLOOM_LOG2) / 32; i += TPK_NT) s_bloom[i] = T.bloom[i];
    s_lut[tid] = (u8)kg.lut[tid];
    const u32 ntiles = (n + TPK_TILE - 1) / TPK_TILE;
    // a tile's text is fetched while the tile before it is probed: all of a thread's 16-byte units by loads issued
    // together (one per loop turn, each waited for before the next, was 4-5 memory latencies in a row per tile),
    // behind an LDS-only barrier so that nothing waits for them before their conversion
    constexpr int NU = (TPK_TILE + 64 + TPK_NT * 16 - 1) / (TPK_NT * 16);
    uint4 raw[NU];
    auto unit_ok = [&](u32 base, int i) -> bool {
        const u32 p = ((u32)i * TPK_NT + tid) * 16;
        const u64 g = (u64)base + p;
        return p < TPK_TILE + 64 && g + 16 <= n && ((((uintptr_t)text) + g) & 15) == 0;
    };
    auto fetch = [&](u32 base) {
#pragma unroll
        for (int i = 0; i < NU; i++) {
            raw[i] = make_uint4(0, 0, 0, 0);
            if (unit_ok(base, i)) raw[i] = *reinterpret_cast<const uint4 *>(text + (u64)base + ((u32)i * TPK_NT + tid) * 16);
        }
    };
    if (blockIdx.x < ntiles) fetch(blockIdx.x * TPK_TILE);
    for (u32 tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const u32 base = tile * TPK_TILE;
        __syncthreads();   // (the previous tile's codes have been read; first tile: the tables are in place)
#pragma unroll
        for (int i = 0; i < NU; i++) {
            const u32 p = ((u32)i * TPK_NT + tid) * 16;
            if (p >= TPK_TILE + 64) continue;
            const u64 g = (u64)base + p;
            if (unit_ok(base, i)) {
                const u32 x[4] = {raw[i].x, raw[i].y, raw[i].z, raw[i].w};
#pragma unroll
                for (int j = 0; j < 16; j++) s_code[TPK_PAD(p + j)] = s_lut[(x[j >> 2] >> (8 * (j & 3))) & 255u];
            } else {
                for (int j = 0; j < 16; j++) s_code[TPK_PAD(p + j)] = g + j < n ? s_lut[text[g + j]] : (u8)0;
            }
        }
        if (tile + gridDim.x < ntiles) fetch((tile + gridDim.x) * TPK_TILE);
        lds_barrier();
        const u32 p0 = tid * TPK_PER;
        if (base + p0 < n) {
            u32 W[WD];
            {
                const u32 *own = reinterpret_cast<const u32 *>(s_code + 68u * tid);        // = TPK_PAD(p0)
                const u32 *nxt = reinterpret_cast<const u32 *>(s_code + 68u * (tid + 1));
#pragma unroll
                for (int i = 0; i < WD; i++) W[i] = i < 16 ? own[i] : nxt[i - 16];
            }
            auto code = [&](int x) -> u32 { return (W[x >> 2] >> (8 * (x & 3))) & 255u; };
            u32 H = 0;
#pragma unroll
            for (int j = 0; j < (int)h; j++) H = tp_rotl(H, 1) ^ tp_sym(code(j));
            u64 hits = 0;
#pragma unroll
            for (int j = 0; j < TPK_PER; j++) {
                const u32 hb = H >> (32 - TP_BLOOM_LOG2);
                hits |= (u64)((s_bloom[hb >> 5] >> (hb & 31)) & 1u) << j;
                H = tp_rotl(H, 1) ^ tp_rotl(tp_sym(code(j)), h & 31u) ^ tp_sym(code(j + (int)h));
            }
            while (hits) {      // (rare) the exact value of the h symbols at such a position, and the table
                const u32 j = (u32)__builtin_ctzll(hits);
                hits &= hits - 1;
                const u32 pos = base + p0 + j;
                if (pos >= n) break;
                u64 v = 0;
                for (u32 x = 0; x < h; x++) v = v * B + s_code[TPK_PAD(p0 + j + x)];
                u32 slot = (u32)(tp_mix(v) >> (64 - TP_SLOT_BITS));
                for (u32 probe = 0; probe < (1u << TP_SLOT_BITS); probe++) {
                    const u64 kk = T.key[slot];
                    if (kk == v) {
                        const u32 jj = atomicAdd(&T.cnt[slot], 1u);
                        const u32 o = atomicAdd(total, 1u);
                        if (o < cap) {
                            const u32 g0 = T.grp[slot];
                            out_slot[o] = g0 + jj;
                            out_idx[o] = pos;
                            out_grp[o] = g0;
                        }
                        break;
                    }
                    if (kk == 0ull) break;
                    slot = (slot + 1) & ((1u << TP_SLOT_BITS) - 1u);
                }
            }
        }
    }
}

// active set arrives unordered from finish_kernel; refine needs it in SA order:
// key = slot << 32 | grp, value = idx, sorted by the slot bits
__global__ __launch_bounds__(256) void pack_active_kernel(const u32 *__restrict__ slot,
                                                          const u32 *__restrict__ grp, u32 m,
                                                          u64 *__restrict__ keys) {
    u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k < m) keys[k] = ((u64)slot[k] << 32) | grp[k];
}
__global__ __launch_bounds__(256) void unpack_active_kernel(const u64 *__restrict__ keys, u32 m,
                                                            u32 *__restrict__ slot,
                                                            u32 *__restrict__ grp) {
    u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k < m) {
        slot[k] = (u32)(keys[k] >> 32);
        grp[k] = (u32)keys[k];
    }
}

// ---- rank of an arbitrary text position ----------------------------------------
// dense: isa[p].  sparse: positions that were ever tied live in a table sorted by
// position (t_idx -> t_rank); every other suffix was unique after round 0, so its
// rank is the position of its round-0 key in the sorted key array.
struct RankLookup {
    const u32 *isa;      // dense, or null
    const u32 *t_idx;    // sparse table
    const u32 *t_rank;
    u32 t_n;
    const u64 *skeys;    // sorted round-0 keys (low byte = payload), or null
    const u64 *tkeys;    // keys sorted by their bits >= tshift only (fast path), or null
    int tshift;
    const u32 *sa;       // the suffix array after round 0 (used when skeys is null)
    // optional accelerators (large tied sets): one bit per text position that is in the table +
    // prefix popcounts per 64-bit word (table index in two loads instead of a binary search),
    // and a directory over the top kdir_bits key bits into tkeys (first index per prefix)
    const u64 *t_bits;
    const u32 *t_dir;
    const u32 *kdir;
    int kdir_bits;
    const u8 *text;
    u32 n, N;
    u32 B, w, s, P;
    u32 h0;              // symbols covered by the round-0 key
    u16 lut[256];
};

// -1 / 0 / +1: first `h` symbols of suffix x vs suffix y (end of text sorts first)
__device__ __forceinline__ int suffix_cmp(const u8 *text, u32 n, u64 x, u64 y, u32 h) {
    for (u32 t = 0; t < h; t++, x++, y++) {
        int cx = x < n ? (int)text[x] : -1, cy = y < n ? (int)text[y] : -1;
        if (cx != cy) return cx < cy ? -1 : 1;
        if (cx < 0) return 0;
    }
    return 0;
}

// WAVE: every lane of the wave asks for the SAME p (a few thousand lookups in all: one wave each); the count inside
// the unsorted bucket is then taken 64 keys per step instead of 8 by one lane
template <bool WAVE = false>
__device__ __forceinline__ u32 rank_of(const RankLookup &r, const u16 *s_lut, u64 p) {
    if (p >= r.N) return 0u;
    if (r.isa) return r.isa[p];
    if (r.t_bits) {
        const u64 wbits = r.t_bits[p >> 6];
        if ((wbits >> (p & 63)) & 1ull)
            return r.t_rank[r.t_dir[p >> 6] + (u32)__popcll(wbits & ((1ull << (p & 63)) - 1ull))];
    } else {   // binary search in the table of ever-tied positions
        u32 lo = 0, hi = r.t_n;
        while (lo < hi) {
            u32 mid = (lo + hi) >> 1;
            if (r.t_idx[mid] < (u32)p) lo = mid + 1; else hi = mid;
        }
        if (lo < r.t_n && r.t_idx[lo] == (u32)p) return r.t_rank[lo];
    }
    // round-0 key of suffix p, straight from the text
    u64 key = 0;
    if (p + 64 <= r.n && r.P * r.s <= 56) {
        // (the window as seven unaligned 8-byte loads issued together, the symbols picked out of registers: a byte load per
        // symbol, each waited for, was up to 56 load latencies in a row per lookup -- most of the time of a sparse lookup)
        u64 wv[7];
#pragma unroll
        for (int x = 0; x < 7; x++) __builtin_memcpy(&wv[x], r.text + p + 8 * x, 8);
        const u32 len = r.P * r.s;
        int sh = 64;
        u32 g = 0, t = 0;
#pragma unroll
        for (int k = 0; k < 56; k++) {
            if ((u32)k < len) {
                g = g * r.B + (u32)s_lut[(u32)(wv[k >> 3] >> (8 * (k & 7))) & 255u];
                if (++t == r.s) {
                    sh -= (int)r.w;
                    key |= (u64)g << sh;
                    g = 0; t = 0;
                }
            }
        }
    } else {
        int sh = 64;
        u64 q = p;
        for (u32 f = 0; f < r.P; f++) {
            u32 g = 0;
            for (u32 t = 0; t < r.s; t++, q++) g = g * r.B + (q < r.n ? (u32)s_lut[r.text[q]] : 0u);
            sh -= r.w;
            key |= (u64)g << sh;
        }
    }
    if (!r.skeys) {
        // suffix p was unique after round 0: its rank is its position in the SA.  tkeys is sorted
        // by the top bits only; inside p's (small, fully ordered) top-bits bucket the first h0
        // symbols are compared against the suffixes the SA points at.  One lower bound over the
        // composite order (top bits, then suffix).
        u64 lo = 0, hi = r.N;
        if (r.tkeys) {
            const u64 tk = key >> r.tshift;
            if (r.kdir) {
                const u64 v = key >> (64 - r.kdir_bits);
                lo = r.kdir[v];
                hi = r.kdir[v + 1];
            }
            if (!WAVE && r.kdir && hi - lo <= 64) {
                // a fine directory leaves a few keys: the rank is lo + the keys of the range below p's key -- keys of
                // other buckets of the range compare by their top bits alone, keys of p's own bucket (in pass order,
                // unsorted) by all bits, and p's key is unique.  No search, no dependent loads.
                u32 below = 0;
                for (u64 j = lo; j < hi; j += 8) {
                    u64 t[8];
#pragma unroll
                    for (int x = 0; x < 8; x++) t[x] = j + x < hi ? r.tkeys[j + x] : ~0ull;
#pragma unroll
                    for (int x = 0; x < 8; x++) below += (t[x] & ~0xffull) < key ? 1u : 0u;
                }
                return (u32)lo + below;
            }
            while (lo < hi) {   // first element of p's top-bits bucket
                const u64 mid = (lo + hi) >> 1;
                if ((r.tkeys[mid] >> r.tshift) < tk) lo = mid + 1; else hi = mid;
            }
            // the bucket occupies the same range in tkeys (in pass order) and in the SA (in suffix
            // order); p's whole key is unique, so its place among the members is the number of
            // members with a smaller key -- no look at the text or the SA (an untied suffix never
            // sits in an over-long bucket: those go to the table whole)
            // (eight keys per step: the loads of a step do not wait for each other -- a level-3 bucket of
            // the MSD way holds ~550 keys, and a walk of dependent loads took 0.3 ms per round)
            u32 below = 0;
            u64 j = lo;
            if (WAVE) {
                const u32 l = lane_id();
                for (;; j += 64) {
                    const u64 jj = j + l;
                    const u64 t = jj < r.N ? r.tkeys[jj] : ~0ull;
                    const bool out = jj >= r.N || (t >> r.tshift) != tk;
                    const u64 ob = __ballot(out);
                    const u64 inside = ob ? ((1ull << __builtin_ctzll(ob)) - 1ull) : ~0ull;   // lanes before the bucket's end
                    below += (u32)__popcll(__ballot(!out && (t & ~0xffull) < key) & inside);
                    if (ob) return (u32)lo + below;
                }
            }
            for (; j + 8 <= r.N; j += 8) {
                u64 t[8];
#pragma unroll
                for (int x = 0; x < 8; x++) t[x] = r.tkeys[j + x];
                bool end = false;
#pragma unroll
                for (int x = 0; x < 8; x++) {
                    end = end || (t[x] >> r.tshift) != tk;   // keys of the bucket are contiguous
                    below += (!end && (t[x] & ~0xffull) < key) ? 1u : 0u;
                }
                if (end) return (u32)lo + below;
            }
            for (; j < r.N; j++) {
                const u64 t = r.tkeys[j];
                if ((t >> r.tshift) != tk) break;
                below += (t & ~0xffull) < key ? 1u : 0u;
            }
            return (u32)lo + below;
        }
        while (lo < hi) {
            u64 mid = (lo + hi) >> 1;
            if (suffix_cmp(r.text, r.n, r.sa[mid], p, r.h0) < 0) lo = mid + 1; else hi = mid;
        }
        return (u32)lo;
    }
    u64 lo = 0, hi = r.N;  // lower_bound over the key bits above the payload byte
    if (r.kdir) {          // (the directory over the sorted keys: a range of ~16 of them, counted instead of searched)
        const u64 v = key >> (64 - r.kdir_bits);
        lo = r.kdir[v];
        hi = r.kdir[v + 1];
        if (hi - lo <= 64) {
            u32 below = 0;
            for (u64 j = lo; j < hi; j += 8) {
                u64 t[8];
#pragma unroll
                for (int x = 0; x < 8; x++) t[x] = j + x < hi ? r.skeys[j + x] : ~0ull;
#pragma unroll
                for (int x = 0; x < 8; x++) below += (t[x] & ~0xffull) < key ? 1u : 0u;
            }
            return (u32)lo + below;
        }
    }
    while (lo < hi) {
        u64 mid = (lo + hi) >> 1;
        if ((r.skeys[mid] & ~0xffull) < key) lo = mid + 1; else hi = mid;
    }
    return (u32)lo;
}

// key2[k] = group << 32 | rank[idx + h]
// HIST: the digit histograms of the sort that follows are built on the way (saves the separate
// pass over the keys); vals_out (optional) receives a copy of idx to be sorted along.
template <bool HIST>
__global__ __launch_bounds__(256) void key2_kernel(const u32 *__restrict__ idx,
                                                   const u32 *__restrict__ grp, RankLookup r, u32 m,
                                                   u32 h, u64 *__restrict__ keys, u32 *__restrict__ vals_out,
                                                   RadixPlanDev plan, u32 *__restrict__ hist) {
    __shared__ u16 s_lut[256];
    __shared__ u32 s_h[HIST ? RDX_MAX_PASSES * RDX_BINS : 1];
    s_lut[threadIdx.x] = r.lut[threadIdx.x];
    if (HIST)
        for (int i = threadIdx.x; i < plan.npass * RDX_BINS; i += 256) s_h[i] = 0;
    __syncthreads();
    if (!HIST && !r.isa && !r.skeys && r.tkeys && vals_out == nullptr && m <= 65536u) {
        // a few thousand members whose partners' ranks are counts inside unsorted buckets: one wave per member
        const u32 wave = (blockIdx.x * 256 + threadIdx.x) >> 6, nwaves = gridDim.x * 4;
        for (u32 k = wave; k < m; k += nwaves) {
            const u32 i = idx[k];
            const u32 rk = rank_of<true>(r, s_lut, (u64)i + h);
            if (lane_id() == 0) keys[k] = ((u64)grp[k] << 32) | rk;
        }
        return;
    }
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < m; k += (u64)gridDim.x * 256) {
        const u32 i = idx[k];
        const u64 key = ((u64)grp[k] << 32) | rank_of(r, s_lut, (u64)i + h);  // idx + h <= n for a tied suffix
        keys[k] = key;
        if (vals_out) vals_out[k] = i;
        if (HIST)
            for (int p = 0; p < plan.npass; p++)
                atomicAdd(&s_h[p * RDX_BINS + (u32)((key >> plan.shift[p]) & plan.mask[p])], 1u);
    }
    if (HIST) {
        __syncthreads();
        for (int i = threadIdx.x; i < plan.npass * RDX_BINS; i += 256) {
            const u32 c = s_h[i];
            if (c) atomicAdd(&hist[i], c);
        }
    }
}

// ---- the key round (round 4): whole buckets of the MSD way's big finish leave as tied groups that share only the 9 symbols
// of the three levels.  Before any rank exists they are ordered by what the KEY still holds -- its bits 39..8, the next 12
// symbols -- with the machinery of a doubling round: key2 = group << 32 | those bits (read where the members lie:
// keys[slot]), the segmented sort, group_kernel<REFINE> without ranks.  The bucket's keys are put back in sorted order, so
// that the sorted-key lookups of the later rounds hold for the suffixes this round resolves.
__global__ __launch_bounds__(256) void key_round_kernel(const u32 *__restrict__ slot, const u32 *__restrict__ grp,
                                                        const u64 *__restrict__ keys, u32 m, u64 *__restrict__ k2,
                                                        u32 *__restrict__ kv) {
    const u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    k2[k] = ((u64)grp[k] << 32) | ((keys[slot[k]] >> 8) & 0xffffffffull);
    kv[k] = k;
}
// sorted position j goes to slot[j] (the slots of a group are its members' slots in order): key bits 39..8 from the sorted
// key2, the bits above from the key that lies there now (a member of the same bucket: the same 24 bits), low byte 0
__global__ __launch_bounds__(256) void key_round_store_kernel(const u64 *__restrict__ k2, const u32 *__restrict__ slot, u32 m,
                                                              u64 *__restrict__ keys) {
    const u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    const u32 p = slot[j];
    keys[p] = (keys[p] & 0xffffff0000000000ull) | ((k2[j] & 0xffffffffull) << 8);
}

// primary = rank of suffix 0 (its SA position once everything is resolved)
// TC_SA_TRACE: members per group-size class (class c: 2^c <= size < 2^(c+1)) of an active set in SA order
__global__ __launch_bounds__(256) void group_size_hist_kernel(const u32 *__restrict__ slot, const u32 *__restrict__ grp,
                                                               u32 m, u64 *__restrict__ hist) {
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    if (j >= m) return;
    if (j + 1 < m && grp[j + 1] == grp[j]) return;
    const u32 size = slot[j] - grp[j] + 1u;
    atomicAdd((unsigned long long *)&hist[31 - __builtin_clz(size)], (unsigned long long)size);
}

__global__ void primary_kernel(RankLookup r, u64 *scalars) {
    __shared__ u16 s_lut[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) s_lut[i] = r.lut[i];
    __syncthreads();
    if (!r.isa && !r.skeys && r.tkeys) {   // (a count inside an unsorted bucket: by the whole wave)
        if (threadIdx.x < 64) {
            const u32 rk = rank_of<true>(r, s_lut, 0);
            if (threadIdx.x == 0) scalars[0] = rk;
        }
        return;
    }
    if (threadIdx.x == 0) scalars[0] = rank_of(r, s_lut, 0);
}

// sparse table build: after sorting (idx, k) by idx: t_idx[q] = idx, t_rank[q] = grp[k], tpos[k] = q
__global__ __launch_bounds__(256) void table_build_kernel(const u64 *__restrict__ sorted_idx,
                                                          const u32 *__restrict__ sorted_k,
                                                          const u32 *__restrict__ grp, u32 m,
                                                          u32 *__restrict__ t_idx,
                                                          u32 *__restrict__ t_rank,
                                                          u32 *__restrict__ tpos, u64 *t_bits) {
    u32 q = blockIdx.x * 256 + threadIdx.x;
    const bool in = q < m;
    u32 p = 0;
    if (in) {
        u32 k = sorted_k[q];
        p = (u32)sorted_idx[q];
        t_idx[q] = p;
        t_rank[q] = grp[k];
        tpos[k] = q;
    }
    if (t_bits) {
        // the positions arrive sorted: the lanes of one bitmap word are neighbours -- their bits are combined in the wave and
        // the word's first lane issues ONE atomic (one per lane: on repeat-rich DNA up to 64 lanes queued on the same word)
        const u32 word = in ? p >> 6 : 0xffffffffu;
        u64 acc = in ? 1ull << (p & 63) : 0ull;
        const u32 lane = lane_id();
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u64 o = __shfl_down(acc, d, 64);
            const u32 ow = __shfl_down(word, d, 64);
            if (lane + d < 64 && ow == word) acc |= o;
        }
        const u32 pw = __shfl_up(word, 1, 64);
        if (in && (lane == 0 || pw != word)) atomicOr((unsigned long long *)&t_bits[word], acc);
    }
}

// t_dir[w] = number of set bits in t_bits[0 .. w): three small launches
#define BDIR_TILE 2048
__global__ __launch_bounds__(256) void bitdir_sum_kernel(const u64 *__restrict__ bits, u32 nwords, u32 *bsum) {
    __shared__ u32 sm[8];
    u32 c = 0;
    for (int k = 0; k < BDIR_TILE / 256; k++) {
        const u32 i = blockIdx.x * BDIR_TILE + k * 256 + threadIdx.x;
        if (i < nwords) c += (u32)__popcll(bits[i]);
    }
    c = wave_sum(c);
    if (lane_id() == 0) sm[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) bsum[blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}
__global__ __launch_bounds__(1024) void bitdir_spine_kernel(u32 *bsum, u32 nb) {
    __shared__ u32 sm[1024 / 64 + 1];
    const u32 per = (nb + 1023) / 1024;
    const u32 lo = threadIdx.x * per < nb ? threadIdx.x * per : nb;
    const u32 hi = lo + per < nb ? lo + per : nb;
    u32 mine = 0;
    for (u32 i = lo; i < hi; i++) mine += bsum[i];
    u32 total;
    u32 run = block_excl_sum<1024>(mine, sm, &total);
    for (u32 i = lo; i < hi; i++) {
        const u32 v = bsum[i];
        bsum[i] = run;
        run += v;
    }
}
__global__ __launch_bounds__(256) void bitdir_down_kernel(const u64 *__restrict__ bits, u32 nwords,
                                                          const u32 *__restrict__ bsum, u32 *__restrict__ dir) {
    __shared__ u32 sm[256 / 64 + 1];
    const u32 i0 = blockIdx.x * BDIR_TILE + threadIdx.x * (BDIR_TILE / 256);
    u32 c[BDIR_TILE / 256], mine = 0;
#pragma unroll
    for (int k = 0; k < BDIR_TILE / 256; k++) {
        c[k] = i0 + k < nwords ? (u32)__popcll(bits[i0 + k]) : 0u;
        mine += c[k];
    }
    u32 total;
    u32 run = bsum[blockIdx.x] + block_excl_sum<256>(mine, sm, &total);
#pragma unroll
    for (int k = 0; k < BDIR_TILE / 256; k++) {
        if (i0 + k < nwords) dir[i0 + k] = run;
        run += c[k];
    }
}
// The same directory in two streaming steps (round 4: 2^26 entries -- a range of ~16 keys at 1 GiB -- would be 2^26 binary
// searches the other way): kdir_mark_kernel writes the first index of every prefix value that occurs (the array starts as
// all ones, entry 2^bits = N), kdir_fill_* give an absent value the entry of the next one that occurs (a suffix minimum).
__global__ __launch_bounds__(256) void kdir_mark_kernel(const u64 *__restrict__ tkeys, u32 N, int bits, u32 *__restrict__ kdir) {
    const u64 j = (u64)blockIdx.x * 256 + threadIdx.x;
    if (j == 0) kdir[(size_t)1 << bits] = N;
    if (j >= N) return;
    const u64 v = tkeys[j] >> (64 - bits);
    if (j == 0 || (tkeys[j - 1] >> (64 - bits)) != v) kdir[v] = (u32)j;
}
#define KDF_CHUNK 4096
__global__ __launch_bounds__(256) void kdir_fill_min_kernel(const u32 *__restrict__ kdir, u64 entries, u32 *__restrict__ bmin) {
    __shared__ u32 sm[4];
    const u64 base = (u64)blockIdx.x * KDF_CHUNK;
    u32 m = 0xffffffffu;
    for (int k = 0; k < KDF_CHUNK / 256; k++) {
        const u64 i = base + (u64)k * 256 + threadIdx.x;
        if (i < entries) { const u32 x = kdir[i]; m = m < x ? m : x; }
    }
    for (int d = 32; d >= 1; d >>= 1) { const u32 o = (u32)__shfl_xor((int)m, d, 64); m = m < o ? m : o; }
    if (lane_id() == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 a = sm[0];
        for (int i = 1; i < 4; i++) a = a < sm[i] ? a : sm[i];
        bmin[blockIdx.x] = a;
    }
}
// bmin[b] := min over blocks > b (exclusive suffix minimum over at most ~16 K entries: one workgroup)
__global__ __launch_bounds__(1024) void kdir_fill_spine_kernel(u32 *bmin, u32 nb) {
    __shared__ u32 s_m[1024];
    const u32 per = (nb + 1023) / 1024;
    const u32 lo = threadIdx.x * per < nb ? threadIdx.x * per : nb;
    const u32 hi = lo + per < nb ? lo + per : nb;
    u32 m = 0xffffffffu;
    for (u32 b = lo; b < hi; b++) { const u32 x = bmin[b]; m = m < x ? m : x; }
    s_m[threadIdx.x] = m;
    __syncthreads();
    u32 run = 0xffffffffu;
    for (u32 t = 1023; t > threadIdx.x; t--) run = run < s_m[t] ? run : s_m[t];
    for (u32 b = hi; b-- > lo;) {
        const u32 x = bmin[b];
        bmin[b] = run;
        run = run < x ? run : x;
    }
}
__global__ __launch_bounds__(256) void kdir_fill_apply_kernel(u32 *__restrict__ kdir, u64 entries, const u32 *__restrict__ bmin) {
    __shared__ u32 s_t[256];
    constexpr int PER = KDF_CHUNK / 256;
    const u64 base = (u64)blockIdx.x * KDF_CHUNK + (u64)threadIdx.x * PER;
    u32 v[PER];
    u32 m = 0xffffffffu;
#pragma unroll
    for (int k = PER - 1; k >= 0; k--) {
        const u64 i = base + k;
        v[k] = i < entries ? kdir[i] : 0xffffffffu;
        m = m < v[k] ? m : v[k];
        v[k] = m;                       // suffix minimum inside the thread's stretch
    }
    s_t[threadIdx.x] = m;
    __syncthreads();
    u32 carry = bmin[blockIdx.x];       // everything behind this block
    for (int t = 255; t > (int)threadIdx.x; t--) carry = carry < s_t[t] ? carry : s_t[t];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const u64 i = base + k;
        if (i < entries) kdir[i] = v[k] < carry ? v[k] : carry;
    }
}

// kdir[v] = first index of tkeys whose top `bits` bits are >= v (v = 0 .. 2^bits)
__global__ __launch_bounds__(256) void kdir_build_kernel(const u64 *__restrict__ tkeys, u32 N, int bits,
                                                         u32 *__restrict__ kdir) {
    const u64 v = (u64)blockIdx.x * 256 + threadIdx.x;
    if (v > (1ull << bits)) return;
    u64 lo = 0, hi = N;
    while (lo < hi) {
        const u64 mid = (lo + hi) >> 1;
        if ((tkeys[mid] >> (64 - bits)) < v) lo = mid + 1; else hi = mid;
    }
    kdir[v] = (u32)lo;
}

__global__ __launch_bounds__(256) void widen_u32_kernel(const u32 *__restrict__ a,
                                                        u64 *__restrict__ b, u32 m) {
    u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k < m) b[k] = a[k];
}

#endif  // __HIPCC__
