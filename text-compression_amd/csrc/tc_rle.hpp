// tc_rle.hpp -- run-length encode / decode on the device.
//
// Replaces seqToRLE (reference RLE/Internal.hs:104-153) and seqFromRLE (:155-189).
// The reference's four-branch state machine (iRLE :134-153) is restated as
// per-position emission rules plus two monotone scans, so one pass with a
// decoupled look-back produces the pairs in order:
//   item after position i is x[i]; count_i = length so far of the run of equal
//   Just symbols ending at i, or -- at a Nothing -- the (stale, Q6) count of the
//   last Just position before it (1 if none).
//   own(i)  = ("1", Nothing)     iff x[i] is Nothing and i >= 1          (:137-138)
//   post(i) = (count_i, x[i])    iff i is last (:125-130), or x[i+1] is Nothing
//             (:135-136), or x[i], x[i+1] are both Just and differ (:149-150)
// count_i comes from H = 1 + last run-head position <= i and J = 1 + last Just
// position <= i (both max-scans): Just: i - H + 2; Nothing: J ? J - H + 1 : 1.
#pragma once
#include "tc_common.hpp"
#include "tc_mtf.hpp"

#define RLE_NT 512
#define RLE_ITEMS 8
#define RLE_TILE (RLE_NT * RLE_ITEMS)
#define RLE16_ITEMS 8
#define RLE16_TILE (RLE_NT * RLE16_ITEMS)

#ifdef __HIPCC__

struct RleArgs {
    u64 N;
    u32 *counts;
    void *syms;  // i16 or u16
    u64 cap;
    u64 *status_pair, *status_sum;
    u32 *ticket;
    u64 *scalars;  // [2] total pairs
    u32 *err;
    int diag;      // timing-only ablation bits (TC_RLE_DIAG): 1 no stores, 2 no look-back
};

// Flag ballots of one 64-position item: heads of Just runs, Just positions, own and
// post emissions (see the rules at the top of this file).
struct RleFlags {
    u64 hb, jb, ob, pb;
};
__device__ __forceinline__ RleFlags rle_flags(int xi, int xp, int xn, u64 j, u64 N) {
    const bool in = j < N;
    const bool just = xi >= 0;
    const bool head = in && just && (j == 0 || xp < 0 || xp != xi);
    const bool own = in && !just && j >= 1;
    bool post = false;
    if (in) {
        if (j == N - 1) post = true;
        else if (xn < 0) post = true;
        else if (!just) post = false;
        else post = xi != xn;
    }
    RleFlags f;
    f.hb = __ballot(head);
    f.jb = __ballot(in && just);
    f.ob = __ballot(own);
    f.pb = __ballot(post);
    return f;
}

template <class Acc, class SymT>
__global__ __launch_bounds__(RLE_NT, 4) void rle_encode_kernel(Acc acc, RleArgs a) {
    constexpr int NW = RLE_NT / 64;
    __shared__ u32 s_wh[NW], s_wj[NW], s_ws[NW];
    __shared__ u64 s_pref[2];
    __shared__ u32 s_tile;
    __shared__ __attribute__((aligned(16))) i16 s_x[RLE_TILE];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const u64 N = a.N;
    const u32 ntiles = (u32)((N + RLE_TILE - 1) / RLE_TILE);
    SymT *syms = reinterpret_cast<SymT *>(a.syms);
    // persistent blocks, but every tile is drawn from the ticket counter when a block is
    // ready for it: a tile only ever waits on tiles already claimed by running blocks, so
    // no co-residency of the whole grid is assumed (other kernels may share the device)
    for (;;) {
    __syncthreads();
    if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
    __syncthreads();
    const u32 tile = s_tile;
    if (tile >= ntiles) break;
    const u64 base = (u64)tile * RLE_TILE + (u64)w * 64 * RLE_ITEMS;

    stage_syms<RLE_TILE, RLE_NT>(acc, (u64)tile * RLE_TILE, N, s_x);
    int x[RLE_ITEMS];
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++) x[k] = staged_value<Acc>(s_x[w * 64 * RLE_ITEMS + k * 64 + l]);
    int xprev0 = -2, xnextT = -2;  // neighbours of the wave segment (wave-uniform loads)
    if (base > 0 && base <= N) xprev0 = acc(base - 1);
    if (base + (u64)64 * RLE_ITEMS < N) xnextT = acc(base + (u64)64 * RLE_ITEMS);

    auto item_flags = [&](int k) {
        int xi = x[k];
        int up = __shfl_up(xi, 1, 64);
        int dn = __shfl_down(xi, 1, 64);
        int p0 = (k == 0) ? xprev0 : __shfl(x[(k + RLE_ITEMS - 1) % RLE_ITEMS], 63, 64);
        int n0 = (k == RLE_ITEMS - 1) ? xnextT : __shfl(x[(k + 1) % RLE_ITEMS], 0, 64);
        int xp = (l == 0) ? p0 : up;
        int xn = (l == 63) ? n0 : dn;
        return rle_flags(xi, xp, xn, base + (u64)k * 64 + l, N);
    };

    // ---- phase 1: wave aggregates ----------------------------------------------------
    u32 wh = 0, wj = 0, ws = 0;  // 1 + last head pos, 1 + last Just pos, emissions
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++) {
        RleFlags f = item_flags(k);
        u64 jb0 = base + (u64)k * 64;
        if (f.hb) wh = (u32)(jb0 + 63u - (u32)__builtin_clzll(f.hb)) + 1u;
        if (f.jb) wj = (u32)(jb0 + 63u - (u32)__builtin_clzll(f.jb)) + 1u;
        ws += (u32)__popcll(f.ob) + (u32)__popcll(f.pb);
    }
    if (l == 0) {
        s_wh[w] = wh;
        s_wj[w] = wj;
        s_ws[w] = ws;
    }
    __syncthreads();
    u32 ph = 0, pj = 0, ps = 0, bh = 0, bj = 0, bs = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        if (i < w) {
            ph = ph > s_wh[i] ? ph : s_wh[i];
            pj = pj > s_wj[i] ? pj : s_wj[i];
            ps += s_ws[i];
        }
        bh = bh > s_wh[i] ? bh : s_wh[i];
        bj = bj > s_wj[i] ? bj : s_wj[i];
        bs += s_ws[i];
    }
    if (w == 0) {
        u64 e = lb_exclusive<OpMaxPair>(a.status_pair, tile, ((u64)bh << 31) | bj, a.err);
        if (l == 0) s_pref[0] = e;
    } else if (w == 1) {
        u64 e = lb_exclusive<OpSum>(a.status_sum, tile, bs, a.err);
        if (l == 0) {
            s_pref[1] = e;
            if ((u64)(tile + 1) * RLE_TILE >= N) a.scalars[2] = e + bs;
        }
    }
    __syncthreads();
    const u32 th = (u32)(s_pref[0] >> 31), tj = (u32)(s_pref[0] & 0x7fffffffu);
    u32 curH = ph > th ? ph : th, curJ = pj > tj ? pj : tj;
    u64 curE = s_pref[1] + ps;

    // ---- phase 2: emit ---------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++) {
        const u64 jb0 = base + (u64)k * 64;
        if (jb0 >= N) break;
        RleFlags f = item_flags(k);
        const u64 j = jb0 + l;
        const u64 le = (2ull << l) - 1ull;
        const u64 hm = f.hb & le, jm = f.jb & le;
        const u32 H = hm ? (u32)(jb0 + 63u - (u32)__builtin_clzll(hm)) + 1u : curH;
        const u32 J = jm ? (u32)(jb0 + 63u - (u32)__builtin_clzll(jm)) + 1u : curJ;
        u64 e = curE + (u64)__popcll(f.ob & lanemask_lt()) + (u64)__popcll(f.pb & lanemask_lt());
        if ((f.ob >> l) & 1ull) {
            if (e < a.cap) {
                a.counts[e] = 1u;
                syms[e] = (SymT)-1;
            }
            e++;
        }
        if ((f.pb >> l) & 1ull) {
            u32 cnt = x[k] >= 0 ? (u32)(j + 2 - H) : (J ? J - H + 1u : 1u);
            if (e < a.cap) {
                a.counts[e] = cnt;
                syms[e] = (SymT)x[k];
            }
        }
        if (f.hb) curH = (u32)(jb0 + 63u - (u32)__builtin_clzll(f.hb)) + 1u;
        if (f.jb) curJ = (u32)(jb0 + 63u - (u32)__builtin_clzll(f.jb)) + 1u;
        curE += (u64)__popcll(f.ob) + (u64)__popcll(f.pb);
    }
    __syncthreads();  // LDS prefix slots are reused by the next tile
    }
}

// Same pass for a stream that cannot contain Nothing (the MTF index stream): only run
// heads and run tails exist.  One tile = RLE16_SUB sub-tiles of 4096 values staged in LDS
// (neighbours are read from the LDS image, no shuffles); the two look-backs -- the waits on
// predecessor tiles measured at ~40 % of the 4096-value version -- are paid once per 32768
// values.
#ifndef RLE16_SUB
#define RLE16_SUB 8    // 1 GiB ACGTN: 2: 3.71 ms, 4: 2.94, 8: 2.56, 12: 2.72, 16: 2.80
#endif
#define RLE16_SUBTILE (RLE_NT * 8)
#undef RLE16_TILE
#define RLE16_TILE (RLE16_SUB * RLE16_SUBTILE)

// IT = u16 (any index stream) or u8 (fused encode of a small alphabet: half the bytes in)
template <class IT>
__global__ __launch_bounds__(RLE_NT, 4) void rle_encode_idx_kernel(const IT *__restrict__ src, RleArgs a) {
    constexpr int NW = RLE_NT / 64, NSEG = RLE16_SUB * NW;
    __shared__ u32 s_wh[NSEG], s_ws[NSEG];
    __shared__ u64 s_pref[2];
    __shared__ u32 s_tile, s_edge[2];
    __shared__ __attribute__((aligned(16))) IT s_x[RLE16_TILE];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const u64 N = a.N;
    const u32 ntiles = (u32)((N + RLE16_TILE - 1) / RLE16_TILE);
    u16 *vals = reinterpret_cast<u16 *>(a.syms);
    for (;;) {
        __syncthreads();
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= ntiles) break;
        const u64 tbase = (u64)tile * RLE16_TILE;
        if (tid == 0) {  // values just outside the tile; 0x10000 = impossible value (forces head / tail)
            s_edge[0] = tbase > 0 ? (u32)src[tbase - 1] : 0x10000u;
            s_edge[1] = tbase + RLE16_TILE < N ? (u32)src[tbase + RLE16_TILE] : 0x10000u;
        }
        {   // stage the tile, 16 bytes per lane per load
            constexpr u32 PER = 16 / sizeof(IT);
            const IT *tsrc = src + tbase;
            if ((((uintptr_t)tsrc) & 15) == 0) {
                for (u32 c = tid; c < RLE16_TILE / PER; c += RLE_NT) {
                    const u64 p0 = tbase + (u64)c * PER;
                    if (p0 + PER <= N) {
                        *reinterpret_cast<uint4 *>(s_x + c * PER) = *reinterpret_cast<const uint4 *>(tsrc + (u64)c * PER);
                    } else {
                        for (u32 q = 0; q < PER; q++) s_x[c * PER + q] = p0 + q < N ? tsrc[(u64)c * PER + q] : (IT)0;
                    }
                }
            } else {
                for (u32 c = tid; c < RLE16_TILE; c += RLE_NT) s_x[c] = tbase + c < N ? tsrc[c] : (IT)0;
            }
            __syncthreads();
        }
        // interior tiles (neither the first nor the last): every position is valid and has both
        // neighbours, the stream-edge tests drop out (block-uniform)
        const bool inner = tile > 0 && tbase + RLE16_TILE < N;
        auto item = [&](int sub, int k, u64 &hbk, u64 &pbk, u32 &xv) {
            const u32 p = (u32)sub * RLE16_SUBTILE + (u32)w * 512 + (u32)k * 64 + (u32)l;
            xv = (u32)s_x[p];
            const u32 xp = p > 0 ? (u32)s_x[p - 1] : s_edge[0];
            const u32 xn = p + 1 < RLE16_TILE ? (u32)s_x[p + 1] : s_edge[1];
            if (inner) {
                hbk = __ballot(xp != xv);
                pbk = __ballot(xn != xv);
            } else {
                const u64 j = tbase + p;
                const bool in = j < N;
                hbk = __ballot(in && (j == 0 || xp != xv));
                pbk = __ballot(in && (j == N - 1 || xn != xv));
            }
        };
        // ---- phase 1: aggregates per (sub-tile, wave) segment of 512 values
#pragma unroll
        for (int sub = 0; sub < RLE16_SUB; sub++) {
            u32 wh = 0, ws = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                u64 hbk, pbk;
                u32 xv;
                item(sub, k, hbk, pbk, xv);
                const u64 jb0 = tbase + (u64)sub * RLE16_SUBTILE + (u64)w * 512 + (u64)k * 64;
                if (hbk) wh = (u32)(jb0 + 63u - (u32)__builtin_clzll(hbk)) + 1u;
                ws += (u32)__popcll(pbk);
            }
            if (l == 0) {
                s_wh[sub * NW + w] = wh;
                s_ws[sub * NW + w] = ws;
            }
        }
        __syncthreads();
        u32 bh = 0, bs = 0;
#pragma unroll
        for (int i = 0; i < NSEG; i++) {
            bh = bh > s_wh[i] ? bh : s_wh[i];
            bs += s_ws[i];
        }
        if (a.diag & 2) {
            if (tid == 0) { s_pref[0] = 0; s_pref[1] = (u64)tile * 12000; a.scalars[2] = 1; }
        } else if (w == 0) {
            u64 e = lb_exclusive<OpMax>(a.status_pair, tile, bh, a.err);
            if (l == 0) s_pref[0] = e;
        } else if (w == 1) {
            u64 e = lb_exclusive<OpSum>(a.status_sum, tile, bs, a.err);
            if (l == 0) {
                s_pref[1] = e;
                if ((u64)(tile + 1) * RLE16_TILE >= N) a.scalars[2] = e + bs;
            }
        }
        __syncthreads();
        // ---- phase 2: emit, segment by segment in position order
#pragma unroll
        for (int sub = 0; sub < RLE16_SUB; sub++) {
            u32 curH = (u32)s_pref[0];
            u64 curE = s_pref[1];
            const int seg = sub * NW + w;
            for (int i = 0; i < seg; i++) {
                curH = curH > s_wh[i] ? curH : s_wh[i];
                curE += s_ws[i];
            }
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u64 jb0 = tbase + (u64)sub * RLE16_SUBTILE + (u64)w * 512 + (u64)k * 64;
                if (jb0 >= N) break;
                u64 hbk, pbk;
                u32 xv;
                item(sub, k, hbk, pbk, xv);
                if ((pbk >> l) & 1ull) {
                    const u64 hm = hbk & ((2ull << l) - 1ull);
                    const u32 H = hm ? (u32)(jb0 + 63u - (u32)__builtin_clzll(hm)) + 1u : curH;
                    const u64 e = curE + (u64)__popcll(pbk & lanemask_lt());
                    if (e < a.cap && !(a.diag & 1)) {
                        a.counts[e] = (u32)(jb0 + l + 2 - H);
                        vals[e] = (u16)xv;
                    }
                }
                if (hbk) curH = (u32)(jb0 + 63u - (u32)__builtin_clzll(hbk)) + 1u;
                curE += (u64)__popcll(pbk);
            }
        }
    }
}

// ---- decode: seqFromRLE (RLE/Internal.hs:155-189) -------------------------------
// (count, Nothing) => exactly one Nothing; else `count` copies.  One pass: a tile of 4096 runs
// computes its output lengths, scans them (block scan + decoupled look-back for the tile's offset)
// and fills its slice at once -- no length / offset arrays in HBM.
// generic device exclusive scan of u64 (three small kernels; run counts only)
#define SCAN_NT 256
#define SCAN_ITEMS 8
#define SCAN_TILE (SCAN_NT * SCAN_ITEMS)
__global__ __launch_bounds__(SCAN_NT) void scan64_reduce_kernel(const u64 *__restrict__ in, u64 n,
                                                                u64 *__restrict__ tsum) {
    __shared__ u64 s[SCAN_NT / 64];
    u64 base = (u64)blockIdx.x * SCAN_TILE;
    u64 v = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        u64 i = base + k * SCAN_NT + threadIdx.x;
        if (i < n) v += in[i];
    }
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) tsum[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}
// single block: exclusive scan of tsum in place; tsum[tiles] = total
__global__ __launch_bounds__(1024) void scan64_spine_kernel(u64 *tsum, u64 tiles) {
    __shared__ u64 s_part[1024];
    u64 per = (tiles + 1023) / 1024;
    u64 lo = threadIdx.x * per, hi = lo + per < tiles ? lo + per : tiles;
    u64 v = 0;
    for (u64 t = lo; t < hi; t++) v += tsum[t];
    s_part[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (int i = 0; i < 1024; i++) {
            u64 c = s_part[i];
            s_part[i] = run;
            run += c;
        }
        tsum[tiles] = run;
    }
    __syncthreads();
    u64 run = s_part[threadIdx.x];
    for (u64 t = lo; t < hi; t++) {
        u64 c = tsum[t];
        tsum[t] = run;
        run += c;
    }
}
__global__ __launch_bounds__(SCAN_NT) void scan64_down_kernel(const u64 *__restrict__ in, u64 n,
                                                              const u64 *__restrict__ tsum,
                                                              u64 *__restrict__ out) {
    // blocked arrangement: thread t owns items [t*ITEMS, (t+1)*ITEMS) of the tile
    __shared__ u64 s[SCAN_NT / 64];
    u64 base = (u64)blockIdx.x * SCAN_TILE + (u64)threadIdx.x * SCAN_ITEMS;
    u64 v[SCAN_ITEMS], tot = 0;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        u64 i = base + k;
        v[k] = i < n ? in[i] : 0;
        tot += v[k];
    }
    u64 inc = tot;
    for (int d = 1; d < 64; d <<= 1) {
        u64 t = __shfl_up(inc, d, 64);
        if ((int)(threadIdx.x & 63) >= d) inc += t;
    }
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = inc;
    __syncthreads();
    u64 pre = tsum[blockIdx.x];
    for (int i = 0; i < (int)(threadIdx.x >> 6); i++) pre += s[i];
    u64 run = pre + inc - tot;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        u64 i = base + k;
        if (i < n) out[i] = run;
        run += v[k];
    }
}

// one lane per run fills its slice; runs >= 32 are filled by the whole wave; runs >= RLE_HUGE are
// queued and filled by the whole grid afterwards (a block of "AAAA..." is a handful of runs)
#define RLE_HUGE 16384
struct HugeRun {
    u64 off, len;
    u32 sym, pad;
};
#define RLD_NT 256
#ifndef RLD_RPT
#define RLD_RPT 16   // runs per thread (a multiple of 8); 1 GiB decode: 8: 39.0 ms, 16: 36.0, 32: 36.3
#endif
#define RLD_TILE (RLD_NT * RLD_RPT)
#ifndef RLD_STAGE
#define RLD_STAGE 12288  // symbols a tile may expand to and still go through LDS
#endif
struct RleDecArgs {
    const u32 *counts;
    const void *syms;
    u64 nruns;
    int has_nothing;
    u64 cap;
    void *out;
    u64 *status;
    u32 *ticket;
    u64 *total;   // receives the expanded length
    u32 *err;
    HugeRun *huge;
    u32 *nhuge;
    u32 huge_cap;
    u32 ntiles;
};
__device__ __forceinline__ u64 wave_incl_sum64(u64 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u64 t = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += t;
    }
    return v;
}
// OutT: the element type written (SymT, or u8 for the index stream of a small alphabet)
template <class SymT, class OutT = SymT>
__global__ __launch_bounds__(RLD_NT) void rle_decode_fused_kernel(RleDecArgs a) {
    __shared__ u64 s_w[RLD_NT / 64];
    __shared__ u64 s_excl;
    __shared__ u32 s_tile;
    __shared__ OutT s_stage[RLD_STAGE];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    const SymT *syms = reinterpret_cast<const SymT *>(a.syms);
    OutT *out = reinterpret_cast<OutT *>(a.out);
    for (;;) {
        __syncthreads();
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= a.ntiles) break;
        const u64 r0 = (u64)tile * RLD_TILE + (u64)tid * RLD_RPT;
        u32 c[RLD_RPT];
        SymT sv[RLD_RPT];
        if (r0 + RLD_RPT <= a.nruns && ((((uintptr_t)a.counts) | ((uintptr_t)syms)) & 15) == 0) {
            const uint4 *pc = reinterpret_cast<const uint4 *>(a.counts + r0);
#pragma unroll
            for (int g = 0; g < RLD_RPT / 4; g++) {
                const uint4 t = pc[g];
                c[4 * g] = t.x; c[4 * g + 1] = t.y; c[4 * g + 2] = t.z; c[4 * g + 3] = t.w;
            }
            const uint4 *ps = reinterpret_cast<const uint4 *>(syms + r0);
#pragma unroll
            for (int g = 0; g < RLD_RPT / 8; g++) {
                const uint4 ts = ps[g];
                const u32 xs[4] = {ts.x, ts.y, ts.z, ts.w};
#pragma unroll
                for (int k = 0; k < 8; k++) sv[8 * g + k] = (SymT)((xs[k >> 1] >> (16 * (k & 1))) & 0xffffu);
            }
        } else {
#pragma unroll
            for (int k = 0; k < RLD_RPT; k++) {
                const bool ok = r0 + k < a.nruns;
                c[k] = ok ? a.counts[r0 + k] : 0u;
                sv[k] = ok ? syms[r0 + k] : (SymT)0;
            }
        }
        u64 len[RLD_RPT], mine = 0;
        if (sizeof(OutT) < sizeof(SymT)) {   // narrowed output: a value that does not fit is out of range anyway
            bool wide = false;
#pragma unroll
            for (int k = 0; k < RLD_RPT; k++) wide |= r0 + k < a.nruns && (u32)(u16)sv[k] > 255u;
            if (wide) atomicOr(a.err, 0x100u);
        }
#pragma unroll
        for (int k = 0; k < RLD_RPT; k++) {
            const bool ok = r0 + k < a.nruns;
            len[k] = !ok ? 0ull : ((a.has_nothing && sv[k] == (SymT)-1) ? 1ull : (u64)c[k]);
            mine += len[k];
        }
        const u64 inc = wave_incl_sum64(mine);
        if (l == 63) s_w[w] = inc;
        __syncthreads();
        u64 wbase = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < RLD_NT / 64; i++) {
            if (i < w) wbase += s_w[i];
            tot += s_w[i];
        }
        if (w == 0) {
            const u64 e = lb_exclusive<OpSum>(a.status, tile, tot, a.err);
            if (l == 0) {
                s_excl = e;
                if (tile + 1 == a.ntiles) *a.total = e + tot;
            }
        }
        __syncthreads();
        if (tot <= RLD_STAGE) {
            // the usual case (short runs): expand into LDS, then write the tile's slice coalesced
            u32 lo = (u32)(wbase + inc - mine);
#pragma unroll
            for (int k = 0; k < RLD_RPT; k++) {
                const u32 ln = (u32)len[k];
                for (u32 q = 0; q < ln; q++) s_stage[lo + q] = (OutT)sv[k];
                lo += ln;
            }
            __syncthreads();
            const u64 ob = s_excl;
            for (u32 i = tid; i < (u32)tot; i += RLD_NT)
                if (ob + i < a.cap) out[ob + i] = s_stage[i];
            continue;
        }
        u64 o = s_excl + wbase + inc - mine;
#pragma unroll
        for (int k = 0; k < RLD_RPT; k++) {
            const u64 ln = len[k];
            const SymT sk = sv[k];
            if (ln && ln < 32)
                for (u64 q = 0; q < ln; q++)
                    if (o + q < a.cap) out[o + q] = (OutT)sk;
            u64 longmask = __ballot(ln >= 32);
            while (longmask) {
                const int src = __builtin_ctzll(longmask);
                longmask &= longmask - 1;
                const u64 lo = __shfl(o, src, 64), ll = __shfl(ln, src, 64);
                const SymT ss = (SymT)__shfl((int)sk, src, 64);
                if (ll >= RLE_HUGE) {
                    if (l == 0) {
                        const u32 slot = atomicAdd(a.nhuge, 1u);
                        if (slot < a.huge_cap) a.huge[slot] = HugeRun{lo, ll, (u32)(u16)ss, 0u};
                    }
                    continue;
                }
                for (u64 q = l; q < ll; q += 64)
                    if (lo + q < a.cap) out[lo + q] = (OutT)ss;
            }
            o += ln;
        }
    }
}
template <class SymT>
__global__ __launch_bounds__(256) void rle_fill_huge_kernel(const HugeRun *__restrict__ huge,
                                                            const u32 *__restrict__ nhuge, u32 huge_cap,
                                                            u64 cap, SymT *__restrict__ out) {
    u32 n = *nhuge;
    if (n > huge_cap) n = huge_cap;
    for (u32 h = 0; h < n; h++) {
        const HugeRun r = huge[h];
        const SymT ss = (SymT)(u16)r.sym;
        for (u64 q = (u64)blockIdx.x * 256 + threadIdx.x; q < r.len; q += (u64)gridDim.x * 256)
            if (r.off + q < cap) out[r.off + q] = ss;
    }
}

#endif  // __HIPCC__
