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

#define RLE_NT 256
#define RLE_ITEMS 16
#define RLE_TILE (RLE_NT * RLE_ITEMS)

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
};

template <class Acc, class SymT>
__global__ __launch_bounds__(RLE_NT) void rle_encode_kernel(Acc acc, RleArgs a) {
    constexpr int NW = RLE_NT / 64;
    __shared__ u32 s_wh[NW], s_wj[NW], s_ws[NW];
    __shared__ u64 s_pref[2];
    __shared__ u32 s_tile;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
    __syncthreads();
    const u32 tile = s_tile;
    const u64 base = (u64)tile * RLE_TILE + (u64)w * 64 * RLE_ITEMS;
    const u64 N = a.N;

    int x[RLE_ITEMS];
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++) {
        u64 j = base + k * 64 + l;
        x[k] = j < N ? acc(j) : -2;
    }
    int xprev0 = -2, xnextT = -2;  // neighbours of the wave segment
    if (base > 0 && base <= N) xprev0 = acc(base - 1);
    {
        u64 jn = base + (u64)64 * RLE_ITEMS;
        if (jn < N) xnextT = acc(jn);
    }
    u32 hinc[RLE_ITEMS], jinc[RLE_ITEMS], eexc[RLE_ITEMS];
    u8 fl[RLE_ITEMS];  // bit0 own, bit1 post
    u32 ch = 0, cj = 0, cs = 0;
    int carry_prev = xprev0;
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++) {
        u64 j = base + k * 64 + l;
        bool in = j < N;
        int xi = x[k];
        int up = __shfl_up(xi, 1, 64);
        int xp = (l == 0) ? carry_prev : up;
        carry_prev = __shfl(xi, 63, 64);
        int dn = __shfl_down(xi, 1, 64);
        int nx0 = xnextT;
        if (k + 1 < RLE_ITEMS) nx0 = __shfl(x[(k + 1) % RLE_ITEMS], 0, 64);
        int xn = (l < 63) ? dn : nx0;
        bool just = xi >= 0;
        bool head = in && just && (j == 0 || xp < 0 || xp != xi);
        bool own = in && !just && j >= 1;
        bool post = false;
        if (in) {
            if (j == N - 1) post = true;
            else if (xn < 0) post = true;
            else if (!just) post = false;
            else post = xi != xn;
        }
        u32 hv = head ? (u32)(j + 1) : 0u;
        u32 jv = (in && just) ? (u32)(j + 1) : 0u;
        u32 hi = wave_incl_max(hv), ji = wave_incl_max(jv);
        hi = hi > ch ? hi : ch;
        ji = ji > cj ? ji : cj;
        hinc[k] = hi;
        jinc[k] = ji;
        ch = __shfl(hi, 63, 64);
        cj = __shfl(ji, 63, 64);
        u32 ev = (own ? 1u : 0u) + (post ? 1u : 0u);
        u32 ei = wave_incl_sum(ev);
        eexc[k] = cs + ei - ev;
        cs += __shfl(ei, 63, 64);
        fl[k] = (u8)((own ? 1 : 0) | (post ? 2 : 0));
    }
    if (l == 63) {
        s_wh[w] = ch;
        s_wj[w] = cj;
        s_ws[w] = cs;
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
    ph = ph > th ? ph : th;
    pj = pj > tj ? pj : tj;
    const u64 pe = s_pref[1] + ps;
    SymT *syms = reinterpret_cast<SymT *>(a.syms);
#pragma unroll
    for (int k = 0; k < RLE_ITEMS; k++) {
        if (!fl[k]) continue;
        u64 j = base + k * 64 + l;
        u32 H = hinc[k] > ph ? hinc[k] : ph;
        u32 J = jinc[k] > pj ? jinc[k] : pj;
        u64 e = pe + eexc[k];
        if (fl[k] & 1) {
            if (e < a.cap) {
                a.counts[e] = 1u;
                syms[e] = (SymT)-1;
            }
            e++;
        }
        if (fl[k] & 2) {
            u32 cnt = x[k] >= 0 ? (u32)(j + 2 - H) : (J ? J - H + 1 : 1u);
            if (e < a.cap) {
                a.counts[e] = cnt;
                syms[e] = (SymT)x[k];
            }
        }
    }
}

// ---- decode: seqFromRLE (RLE/Internal.hs:155-189) -------------------------------
// (count, Nothing) => exactly one Nothing; else `count` copies.  Exclusive scan of
// the output lengths, then each run fills its slice.
template <class SymT>
__global__ __launch_bounds__(256) void rle_len_kernel(const u32 *__restrict__ counts,
                                                      const SymT *__restrict__ syms, u64 nruns,
                                                      bool has_nothing, u64 *__restrict__ len) {
    u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k >= nruns) return;
    bool nothing = has_nothing && (syms[k] == (SymT)-1);
    len[k] = nothing ? 1ull : (u64)counts[k];
}

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

// one wave per run fills its slice (long runs are rare; short runs dominate on
// iid data where a wave handles 64 runs at once instead)
template <class SymT>
__global__ __launch_bounds__(256) void rle_fill_kernel(const u64 *__restrict__ offs,
                                                       const u32 *__restrict__ counts,
                                                       const SymT *__restrict__ syms, u64 nruns,
                                                       bool has_nothing, u64 cap,
                                                       SymT *__restrict__ out) {
    u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    bool in = k < nruns;
    u64 o = in ? offs[k] : 0;
    SymT s = in ? syms[k] : (SymT)0;
    u64 len = 0;
    if (in) len = (has_nothing && s == (SymT)-1) ? 1ull : (u64)counts[k];
    // short runs: each lane writes its own; long runs: the wave cooperates
    const u64 LONG = 32;
    if (in && len < LONG)
        for (u64 q = 0; q < len; q++)
            if (o + q < cap) out[o + q] = s;
    u64 longmask = __ballot(in && len >= LONG);
    while (longmask) {
        int src = __builtin_ctzll(longmask);
        longmask &= longmask - 1;
        u64 lo = __shfl(o, src, 64), ll = __shfl(len, src, 64);
        SymT ss = (SymT)__shfl((int)s, src, 64);
        for (u64 q = lane_id(); q < ll; q += 64)
            if (lo + q < cap) out[lo + q] = ss;
    }
}

#endif  // __HIPCC__
