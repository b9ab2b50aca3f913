// tc_pack.hpp -- wire format of a tc_block's runs (payload of the multi-GPU gather).
// The reference has no wire format; this one is ours and is an exact inverse pair.
//
// sigma <= 6 ("nibble stream", what an ACGTN record produces: 5 letters + sentinel):
//   code 0..11   start of a run: value = code % 6, count = 1 + code / 6
//   code 12, 13  follows a start with count 2: count becomes 3, 4
//   code 14      follows a start: count is the next word of the escape list (run order)
//   code 15      padding (only at the very end: the stream is dense and padded to a 16-byte boundary)
// so a run costs 4 bits (count 1, 2) or 8 bits; iid ACGTN: 1.04 nibbles per run.
// The stream is a function of the runs alone, so the two producers below give the same bytes:
// pack_nib_kernel (from a tc_block's run arrays) and rle_nib_kernel (straight from the MTF index
// stream: the fused encode -> container path, no run arrays in HBM).  Tiles of either kernel start at
// arbitrary nibble offsets; a 16-byte unit shared by neighbouring tiles is completed by atomicOr of
// 32-bit words, which is why the output must be ZERO before the kernel runs.
// 6 < sigma <= 16: one byte per run (value | min(count,15) << 4); sigma > 16: two bytes per run;
// both with an escape list of (run index, count) pairs.
#pragma once
#include "tc_common.hpp"
#include "tc_mtf.hpp"

#define PK_NT 256
#define PK_RPT 16                       // runs per thread per sub-tile
#define PK_SUBRUNS (PK_NT * PK_RPT)     // 4096
#define PK_SUB 4
#define PK_TILE (PK_SUBRUNS * PK_SUB)   // 16384 runs per tile
#define PK_WORDS (PK_TILE * 2 / 16)     // u64 words of nibbles, worst case (2 per run)
#define PK_NIB_SIGMA 6
static_assert(PK_TILE * 2 < 65536, "a tile's nibble count shares a 32-bit word with its escape count (16 bits each)");

static inline int pack_format(u32 sigma) { return sigma <= PK_NIB_SIGMA ? 0 : sigma <= 16 ? 1 : 2; }

struct PackNibArgs {
    const u32 *cnt;
    const u16 *val;
    u64 nruns;
    u8 *out;        // 16-byte aligned
    u64 cap_units;  // 16-byte units available in `out`
    u32 *esc;
    u64 esc_cap;
    u64 *status;
    u32 *ticket;
    u32 *err;
    u32 ntiles;
};

// append the 1 or 2 nibbles of one run to the 128-bit string (L, H) of `len` nibbles
__device__ __forceinline__ void pk_append(u64 &L, u64 &H, u32 &len, u32 &nesc, u32 c, u32 v) {
    const u32 c1 = c - 1u;  // 0..3 for counts 1..4
    u32 pair, nl;
    if (c1 < 2u) {
        pair = v + 6u * c1;
        nl = 1;
    } else if (c1 < 4u) {
        pair = (v + 6u) | ((10u + c1) << 4);
        nl = 2;
    } else {
        pair = v | (14u << 4);
        nl = 2;
        nesc++;
    }
    const u32 sh = (len & 15u) * 4u;
    if (len < 16u) {
        L |= (u64)pair << sh;
        if (len == 15u && nl == 2u) H |= (u64)(pair >> 4);
    } else {
        H |= (u64)pair << sh;
    }
    len += nl;
}

// the tile's nibble image (u64 words, aligned to the GLOBAL 16-byte unit grid: nibble q0 of the stream sits at
// image nibble q0 & 31) goes out: units that lie wholly inside [q0, q0 + tn) by one 16-byte store, the (up to
// two) units shared with neighbouring tiles by atomicOr of their non-zero 32-bit words.  `pad`: the stream
// ends with this tile -- the rest of its last unit is filled with code 15.  Every thread of the block calls.
template <int NT>
__device__ __forceinline__ void nib_image_out(const u64 *img, u64 q0, u32 tn, bool pad, u8 *out, u64 cap_units) {
    if (tn == 0) return;
    const u64 u_first = q0 >> 5, u_last = (q0 + tn - 1) >> 5;
    const u32 nunits = (u32)(u_last - u_first) + 1u;
    for (u32 u = threadIdx.x; u < nunits; u += NT) {
        u64 w0 = img[2 * u], w1 = img[2 * u + 1];
        const u64 g = u_first + u;
        if (g >= cap_units) continue;
        const u64 n0 = g << 5;                       // first nibble of this unit in the stream
        const bool whole = n0 >= q0 && n0 + 32 <= q0 + tn;
        if (pad && g == u_last) {
            const u32 used = (u32)(q0 + tn - n0);    // 1..32 nibbles of this unit are real
            if (used < 16u) { w0 |= ~0ull << (used * 4u); w1 = ~0ull; }
            else if (used < 32u) w1 |= ~0ull << ((used - 16u) * 4u);
        }
        if (whole || (pad && g == u_last && n0 >= q0)) {
            reinterpret_cast<ulonglong2 *>(out)[g] = make_ulonglong2(w0, w1);
        } else {
            u32 *o32 = reinterpret_cast<u32 *>(out) + 4 * g;
            if ((u32)w0) atomicOr(o32, (u32)w0);
            if ((u32)(w0 >> 32)) atomicOr(o32 + 1, (u32)(w0 >> 32));
            if ((u32)w1) atomicOr(o32 + 2, (u32)w1);
            if ((u32)(w1 >> 32)) atomicOr(o32 + 3, (u32)(w1 >> 32));
        }
    }
}

// look-back value of both producers: nibbles so far (< 2^32) << 30 | escapes so far (< 2^30: an encoded record
// has a count >= 5 behind each, so at most N / 5 of them; run arrays handed to tc_block_pack_dev with more
// escapes than that are refused: device flag 0x2, an error instead of wrong bytes)
#define NIB_LB_SHIFT 30
#define NIB_LB(nibs, esc) (((u64)(nibs) << NIB_LB_SHIFT) | (u64)(esc))
#define NIB_LB_ESC(v) ((v) & ((1ull << NIB_LB_SHIFT) - 1))

__global__ __launch_bounds__(PK_NT) void pack_nib_kernel(PackNibArgs a) {
    __shared__ u64 nib[PK_WORDS + 4];
    __shared__ u32 wsum[PK_SUB][PK_NT / 64];
    __shared__ u32 s_tile;
    __shared__ u64 s_excl;
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    while (true) {
        __syncthreads();   // the image and s_tile of the previous tile have been read by everybody
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
        for (int i = tid; i < PK_WORDS + 4; i += PK_NT) nib[i] = 0;
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= a.ntiles) break;
        const u64 base = (u64)tile * PK_TILE;

        u64 L[PK_SUB], H[PK_SUB];
        u32 x[PK_SUB];  // nibbles | escapes << 16
#pragma unroll
        for (int s = 0; s < PK_SUB; s++) {
            const u64 r0 = base + (u64)s * PK_SUBRUNS + (u64)tid * PK_RPT;
            u32 c[PK_RPT], v[PK_RPT];
            if (r0 + PK_RPT <= a.nruns) {
                const uint4 *pc = reinterpret_cast<const uint4 *>(a.cnt + r0);
                const uint4 *pv = reinterpret_cast<const uint4 *>(a.val + r0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint4 t = pc[q];
                    c[4 * q] = t.x; c[4 * q + 1] = t.y; c[4 * q + 2] = t.z; c[4 * q + 3] = t.w;
                }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    uint4 t = pv[q];
                    v[8 * q] = t.x & 0xffffu; v[8 * q + 1] = t.x >> 16;
                    v[8 * q + 2] = t.y & 0xffffu; v[8 * q + 3] = t.y >> 16;
                    v[8 * q + 4] = t.z & 0xffffu; v[8 * q + 5] = t.z >> 16;
                    v[8 * q + 6] = t.w & 0xffffu; v[8 * q + 7] = t.w >> 16;
                }
            } else {
#pragma unroll
                for (int j = 0; j < PK_RPT; j++) {
                    const bool ok = r0 + j < a.nruns;
                    c[j] = ok ? a.cnt[r0 + j] : 0u;
                    v[j] = ok ? a.val[r0 + j] : 0xffffu;  // 0xffff: no run
                }
            }
            u64 l = 0, h = 0;
            u32 len = 0, ne = 0;
#pragma unroll
            for (int j = 0; j < PK_RPT; j++)
                if (v[j] != 0xffffu) pk_append(l, h, len, ne, c[j], v[j] < 5u ? v[j] : 5u);
            L[s] = l; H[s] = h;
            x[s] = len | (ne << 16);
        }
        // offsets: sub-tile major, then thread order
        u32 inc[PK_SUB];
#pragma unroll
        for (int s = 0; s < PK_SUB; s++) inc[s] = wave_incl_sum(x[s]);
        if (lane == 63) {
#pragma unroll
            for (int s = 0; s < PK_SUB; s++) wsum[s][w] = inc[s];
        }
        __syncthreads();
        u32 run = 0, excl[PK_SUB];
#pragma unroll
        for (int s = 0; s < PK_SUB; s++) {
            u32 b = run;
#pragma unroll
            for (int i = 0; i < PK_NT / 64; i++) {
                const u32 t = wsum[s][i];
                if (i < w) b += t;
                run += t;
            }
            excl[s] = b + inc[s] - x[s];
        }
        const u32 tile_len = run & 0xffffu, tile_esc = run >> 16;
        if (w == 0) {
            const u64 e = lb_exclusive<OpSum>(a.status, tile, NIB_LB(tile_len, tile_esc), a.err);
            if (lane == 0) s_excl = e;
        }
        __syncthreads();
        const u64 q0 = s_excl >> NIB_LB_SHIFT, excl_esc = NIB_LB_ESC(s_excl);
        if (tid == 0 && excl_esc + tile_esc >= (1ull << NIB_LB_SHIFT)) atomicOr(a.err, 2u);
        const u32 a0 = (u32)(q0 & 31u);    // where the tile's first nibble sits in the image
#pragma unroll
        for (int s = 0; s < PK_SUB; s++) {
            const u32 off = a0 + (excl[s] & 0xffffu);
            const u32 wd = off >> 4, sh = (off & 15u) * 4u;
            const u64 l = L[s], h = H[s];
            const u64 w0 = l << sh;
            const u64 w1 = (sh ? l >> (64 - sh) : 0ull) | (h << sh);
            const u64 w2 = sh ? h >> (64 - sh) : 0ull;
            if (w0) atomicOr((unsigned long long *)&nib[wd], (unsigned long long)w0);
            if (w1) atomicOr((unsigned long long *)&nib[wd + 1], (unsigned long long)w1);
            if (w2) atomicOr((unsigned long long *)&nib[wd + 2], (unsigned long long)w2);
        }
        __syncthreads();
        nib_image_out<PK_NT>(nib, q0, tile_len, tile + 1 == a.ntiles, a.out, a.cap_units);
        // escapes are rare: their owners re-read the counts and append them in run order
#pragma unroll
        for (int s = 0; s < PK_SUB; s++) {
            if (x[s] >> 16) {
                const u64 r0 = base + (u64)s * PK_SUBRUNS + (u64)tid * PK_RPT;
                u64 e = excl_esc + (excl[s] >> 16);
                for (int j = 0; j < PK_RPT; j++) {
                    if (r0 + j >= a.nruns) break;
                    const u32 c = a.cnt[r0 + j];
                    if (c - 1u >= 4u) {
                        if (e < a.esc_cap) a.esc[e] = c;
                        e++;
                    }
                }
            }
        }
    }
}

// ---- the fused producer: MTF index stream (one byte per symbol, values < 6) -> nibble stream -----------------
// Replaces, for the container of a small-alphabet record, rle_encode_idx_kernel + pack_nib_kernel: the 6-byte
// run records (5.2 GB for a 1 GiB ACGTN record) are never written and re-read.  seqToRLE of the index stream
// (reference RLE/Internal.hs:104-153; no Nothing in this stream, so only run ends exist) in a blocked
// arrangement: a thread owns 16 consecutive positions per sub-tile (one 16-byte load).
//   E    bit i: position p0 + i ends a run (its successor differs, or it is the last position): SWAR byte compare
//   the length of a run is the distance from the previous end, which is the previous set bit of E or -- for
//   the first run of a chunk -- the last end before the chunk: a max-scan over the tile in position order
//   (inside a wave: ballot + one bpermute; waves and sub-tiles through LDS; tiles by look-back A).
//   With X = E << 4 | (the last end before the chunk, if within 4 positions) the length classes are masks:
//   length >= 2: X & ~(X << 1), >= 3: & ~(X << 2), >= 5: & ~(X << 3) & ~(X << 4) -- so nibbles and escapes per
//   chunk are popcounts, their prefix over the tile one packed scan, their prefix over the tiles look-back B
//   (sum), and the nibble strings are built -- position by position, four positions to a 32-bit group -- when
//   their place in the stream is already known.
#ifndef RN_NT
#define RN_NT 512
#endif
#define RN_SUB 4
#define RN_SUBSZ (RN_NT * 16)             // 8192 positions
#define RN_TILE (RN_SUB * RN_SUBSZ)       // 32768 positions
#define RN_NSEG (RN_SUB * (RN_NT / 64))   // (sub-tile, wave) segments of 1024 positions, in position order
#define RN_IMG_WORDS (RN_TILE / 16 + 4)   // a tile emits at most one nibble per position
static_assert(RN_NSEG <= 64, "segment scans are done by one wave");

struct RleNibArgs {
    const u8 *src;   // index stream; 16-byte aligned; readable up to the next 16-byte boundary behind N
    u64 N;
    u8 *out;         // 16-byte aligned, ZERO for cap_units * 16 bytes
    u64 cap_units;
    u32 *esc;
    u64 esc_cap;
    u64 *status_a, *status_b;
    u32 *ticket;
    u64 *totals;     // [0] runs (atomicAdd), [1] nibbles, [2] escapes (written by the last tile)
    u32 *err;
    u32 ntiles;
    int diag;        // timing-only ablation bits (TC_RLE_DIAG; results are wrong): 1 no look-back B, 2 no strings, 4 no output
};

// bits 0..3: which of the four bytes of d are non-zero
__device__ __forceinline__ u32 nz_bytes4(u32 d) {
    const u32 t = (((d & 0x7f7f7f7fu) + 0x7f7f7f7fu) | d) & 0x80808080u;
    return ((t >> 7) * 0x01020408u) >> 24;
}
// exclusive prefix over the RN_NSEG per-segment values of a tile (a wave computes it for its RN_SUB segments):
// lane g holds segment g's value
template <class Op>
__device__ __forceinline__ u32 seg_incl_scan(u32 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v = (u32)Op::apply(v, t);
    }
    return v;
}

// The last run end before a tile (1 + its position; 0: none), for wave 0 of the tile's workgroup.  It almost always
// lies within the 64 positions in front of the tile: those are read directly -- no dependence on the predecessor
// tile at all -- and only a run of 64 or more across the tile's edge falls back to the look-back chain.  Either way
// the tile's own inclusive value (its last end, else the incoming one) is published for tiles that do chain.
__device__ __forceinline__ u32 rn_last_end_before(const u8 *src, u64 tbase, u32 tile, u32 agg, u64 *status_a, u32 *err) {
    if (tile == 0) {
        if (lane_id() == 0) lb_store(&status_a[0], LB_FLAG_INC | (u64)agg);
        return 0u;
    }
    const u64 p = tbase - 64 + lane_id();          // (tbase >= RN_TILE >= 64; p + 1 <= tbase < N)
    const u64 m = __ballot(src[p] != src[p + 1]);
    if (m) {
        const u32 tin = (u32)(tbase - 64) + (u32)(63 - __builtin_clzll(m)) + 1u;
        if (lane_id() == 0) lb_store(&status_a[tile], LB_FLAG_INC | (u64)(agg ? agg : tin));
        return tin;
    }
    return (u32)lb_exclusive<OpMax>(status_a, tile, agg, err);
}

__global__ __launch_bounds__(RN_NT, 2) void rle_nib_kernel(RleNibArgs a) {
    constexpr int NW = RN_NT / 64;
    __shared__ u64 img[RN_IMG_WORDS];
    __shared__ u32 s_last[RN_NSEG], s_carry[RN_NSEG], s_sum[RN_NSEG], s_runs[NW];
    __shared__ u32 s_tile;
    __shared__ u64 s_pref;
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const u64 N = a.N;
    for (;;) {
        __syncthreads();   // the image, s_tile and s_pref of the previous tile have been read by everybody
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
        for (int i = tid; i < RN_IMG_WORDS; i += RN_NT) img[i] = 0;
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= a.ntiles) break;
        const u64 tbase = (u64)tile * RN_TILE;
        const bool edge = tbase + RN_TILE >= N;   // the stream ends in this tile (block-uniform)

        // ---- 1: load, run ends, last end per chunk ------------------------------------------------------
        uint4 x[RN_SUB];
        u32 E[RN_SUB], pv[RN_SUB];
        u32 hasmask = 0;
        u32 nxl[RN_SUB];   // (lane 63) the byte behind the wave's last group
#pragma unroll
        for (int s = 0; s < RN_SUB; s++) {   // all of the tile's loads first: one memory latency, not RN_SUB in a row
            const u64 p0 = tbase + (u64)s * RN_SUBSZ + (u64)tid * 16;
            x[s] = make_uint4(0, 0, 0, 0);
            nxl[s] = 0;
            if (!edge || p0 < N) x[s] = *reinterpret_cast<const uint4 *>(a.src + p0);
            if (lane == 63 && (!edge || p0 + 16 < N)) nxl[s] = (u32)a.src[p0 + 16];
        }
#pragma unroll
        for (int s = 0; s < RN_SUB; s++) {
            const u64 p0 = tbase + (u64)s * RN_SUBSZ + (u64)tid * 16;
            const uint4 q = x[s];
            u32 nx = __shfl_down(q.x, 1, 64);
            if (lane == 63) nx = nxl[s];
            const u32 y0 = __builtin_amdgcn_alignbyte(q.y, q.x, 1), y1 = __builtin_amdgcn_alignbyte(q.z, q.y, 1);
            const u32 y2 = __builtin_amdgcn_alignbyte(q.w, q.z, 1), y3 = __builtin_amdgcn_alignbyte(nx, q.w, 1);
            u32 e = nz_bytes4(q.x ^ y0) | (nz_bytes4(q.y ^ y1) << 4) | (nz_bytes4(q.z ^ y2) << 8) | (nz_bytes4(q.w ^ y3) << 12);
            if (edge) {
                const u32 valid = p0 >= N ? 0u : (N - p0 >= 16 ? 0xffffu : ((1u << (u32)(N - p0)) - 1u));
                e &= valid;
                if (p0 < N && N - p0 <= 16) e |= 1u << (u32)(N - 1 - p0);   // the last position ends its run
            }
            x[s] = q;
            E[s] = e;
            const u32 last1 = e ? (u32)p0 + 32u - (u32)__builtin_clz(e) : 0u;   // 1 + position of the chunk's last end
            const u64 m = __ballot(e != 0);
            const u64 pm = m & lanemask_lt();
            const int src = pm ? 63 - __builtin_clzll(pm) : 0;
            pv[s] = __shfl(last1, src, 64);
            if (pm) hasmask |= 1u << s;
            const u32 wlast = m ? __shfl(last1, 63 - __builtin_clzll(m), 64) : 0u;
            if (lane == 0) s_last[s * NW + w] = wlast;
        }
        __syncthreads();
        // ---- 2: last end before every segment (wave 0: scan over the segments + look-back A over the tiles)
        if (w == 0) {
            const u32 v = lane < RN_NSEG ? s_last[lane] : 0u;
            const u32 inc = seg_incl_scan<OpMax>(v);
            const u32 agg = __shfl(inc, RN_NSEG - 1, 64);
            u32 ex = __shfl_up(inc, 1, 64);
            if (lane == 0) ex = 0;
            const u32 tin = rn_last_end_before(a.src, tbase, tile, agg, a.status_a, a.err);
            if (lane < RN_NSEG) s_carry[lane] = ex > tin ? ex : tin;
        }
        __syncthreads();
        // ---- 3: length classes, nibbles and escapes per chunk, prefix inside the tile ---------------------
        u32 prev1[RN_SUB], cnt[RN_SUB], inc[RN_SUB];   // cnt: nibbles | escapes << 16
        u32 myruns = 0;
#pragma unroll
        for (int s = 0; s < RN_SUB; s++) {
            const u32 p0 = (u32)(tbase + (u64)s * RN_SUBSZ + (u64)tid * 16);
            const u32 pe = ((hasmask >> s) & 1u) ? pv[s] : s_carry[s * NW + w];
            prev1[s] = pe;
            // the last end before the chunk is position pe - 1 (pe == 0: the virtual end before position 0)
            const u32 back = p0 + 1u - pe;                       // 1: directly before the chunk, 2, 3, 4 ...
            const u32 vb = (back - 1u < 4u) ? 1u << (4u - back) : 0u;
            const u32 X = (E[s] << 4) | vb;
            const u32 g2 = X & ~(X << 1);
            const u32 g3 = g2 & ~(X << 2);
            const u32 g5 = g3 & ~(X << 3) & ~(X << 4);
            const u32 runs = (u32)__popc(E[s]);
            myruns += runs;
            cnt[s] = (runs + (u32)__popc((g3 >> 4) & 0xffffu)) | ((u32)__popc((g5 >> 4) & 0xffffu) << 16);
            inc[s] = wave_incl_sum(cnt[s]);
            if (lane == 63) s_sum[s * NW + w] = inc[s];
        }
        myruns = wave_sum(myruns);
        if (lane == 0) s_runs[w] = myruns;
        __syncthreads();
        u32 excl[RN_SUB];
        u32 tile_cnt;
        {
            const u32 v = lane < RN_NSEG ? s_sum[lane] : 0u;
            const u32 sc = seg_incl_scan<OpSum>(v);
            tile_cnt = __shfl(sc, RN_NSEG - 1, 64);
#pragma unroll
            for (int s = 0; s < RN_SUB; s++) {
                const int g = s * NW + w;
                const u32 before = g ? __shfl(sc, g - 1, 64) : 0u;
                excl[s] = before + inc[s] - cnt[s];
            }
        }
        const u32 tn = tile_cnt & 0xffffu, te = tile_cnt >> 16;
        if (w == 1) {
            const u64 e = (a.diag & 1) ? NIB_LB((u64)tile * 27000ull, (u64)tile * 50ull)
                                       : lb_exclusive<OpSum>(a.status_b, tile, NIB_LB(tn, te), a.err);
            if (lane == 0) {
                s_pref = e;
                if (tile + 1 == a.ntiles) {
                    a.totals[1] = (e >> NIB_LB_SHIFT) + tn;
                    a.totals[2] = NIB_LB_ESC(e) + te;
                }
            }
        } else if (tid == 0) {
            u32 r = 0;
#pragma unroll
            for (int i = 0; i < NW; i++) r += s_runs[i];
            if (r) atomicAdd((unsigned long long *)&a.totals[0], (unsigned long long)r);
        }
        __syncthreads();
        const u64 q0 = s_pref >> NIB_LB_SHIFT, e0 = NIB_LB_ESC(s_pref);
        const u32 a0 = (u32)(q0 & 31u);
        // ---- 4: the nibble strings, into the tile's image --------------------------------------------------
        if (!(a.diag & 2))
#pragma unroll
        for (int s = 0; s < RN_SUB; s++) {
            const u32 p0 = (u32)(tbase + (u64)s * RN_SUBSZ + (u64)tid * 16);
            const u32 xs[4] = {x[s].x, x[s].y, x[s].z, x[s].w};
            u32 prev = prev1[s];
            u64 L = 0;
            u32 H = 0, len = 0;     // the chunk's string: at most 17 nibbles
#pragma unroll
            for (int d = 0; d < 4; d++) {
                u32 g = 0, sh = 0;  // the nibbles of four positions: at most five
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int i = 4 * d + k;
                    const bool end = (E[s] >> i) & 1u;
                    const u32 pos1 = p0 + (u32)i + 1u;
                    const u32 c1 = pos1 - prev - 1u;                      // run length - 1
                    const u32 v = (xs[d] >> (8 * k)) & 0xffu;
                    const u32 first = v + ((c1 - 1u < 3u) ? 6u : 0u);     // lengths 2, 3, 4: value + 6
                    const u32 second = (c1 < 4u ? c1 : 4u) + 10u;         // 3 -> 12, 4 -> 13, >= 5 -> 14
                    const bool two = c1 >= 2u;
                    const u32 pair = first | (two ? second << 4 : 0u);
                    if (end) {
                        g |= pair << sh;
                        sh += two ? 8u : 4u;
                        prev = pos1;
                    }
                }
                const u32 bsh = len * 4u;
                if (len < 16u) {
                    L |= (u64)g << bsh;
                    if (bsh > 32u) H |= g >> (64u - bsh);
                } else {
                    H |= g << (bsh - 64u);
                }
                len += sh >> 2;
            }
            const u32 off = a0 + (excl[s] & 0xffffu);
            const u32 wd = off >> 4, bs = (off & 15u) * 4u;
            const u64 w0 = L << bs;
            const u64 w1 = (bs ? L >> (64u - bs) : 0ull) | ((u64)H << bs);
            if (w0) atomicOr((unsigned long long *)&img[wd], (unsigned long long)w0);
            if (w1) atomicOr((unsigned long long *)&img[wd + 1], (unsigned long long)w1);
            if (cnt[s] >> 16) {   // rare: lengths >= 5 go to the escape list, in run order
                u64 e = e0 + (excl[s] >> 16);
                u32 pr = prev1[s], em = E[s];
                while (em) {
                    const u32 i = (u32)__builtin_ctz(em);
                    em &= em - 1u;
                    const u32 c = p0 + i + 1u - pr;
                    pr = p0 + i + 1u;
                    if (c >= 5u) {
                        if (e < a.esc_cap) a.esc[e] = c;
                        e++;
                    }
                }
            }
        }
        __syncthreads();
        if (!(a.diag & 4)) nib_image_out<RN_NT>(img, q0, tn, tile + 1 == a.ntiles, a.out, a.cap_units);
    }
}

// ---- the same blocked formulation writing the run ARRAYS (tc_block: run_count u32[], run_value u16[]) -------------
// seqToRLE of a byte-wide index stream (the fused encode of an alphabet of <= 256 symbols), replacing the striped
// rle_encode_idx_kernel<u8> on that path: run ends, previous ends and the two look-backs exactly as in rle_nib_kernel;
// a tile's runs are staged in LDS as one byte each for the length (255 = "255 or more": the owner stores those
// lengths itself) and for the value, and leave as coalesced 4-byte / 2-byte stores.
struct RleBlkArgs {
    const u8 *src;   // 16-byte aligned; readable up to the next 16-byte boundary behind N
    u64 N;
    u32 *counts;
    u16 *vals;
    u64 cap;
    u64 *status_a, *status_b;
    u32 *ticket;
    u64 *scalars;    // [2] total runs
    u32 *err;
    u32 ntiles;
    u32 wide;        // counts and vals are 16-byte aligned: groups of 8 runs may leave as 16-byte stores
};

// PACK4 (values < 16: the index stream of an alphabet of at most 16 symbols): one staged byte per run, value in the low
// and min(length, 15) in the high nibble (15 = "15 or more") -- 32 KB of staging instead of 64, twice the workgroups per CU
template <bool PACK4>
__global__ __launch_bounds__(RN_NT, 4) void rle_blk_kernel(RleBlkArgs a) {
    constexpr int NW = RN_NT / 64;
    constexpr u32 LONGC = PACK4 ? 15u : 255u;
    // (run j of the tile is staged at index j + (e0 & 7): groups of 8 staged runs that are 8-byte aligned in LDS are
    // then 32-byte aligned in run_count[] and 16-byte aligned in run_value[] -- they leave as 16-byte stores)
    __shared__ __attribute__((aligned(16))) u8 s_c8[RN_TILE + 16], s_v8[PACK4 ? 16 : RN_TILE + 16];
    __shared__ u32 s_last[RN_NSEG], s_carry[RN_NSEG], s_sum[RN_NSEG];
    __shared__ u32 s_tile;
    __shared__ u64 s_pref;
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const u64 N = a.N;
    for (;;) {
        __syncthreads();   // the staged runs, s_tile and s_pref of the previous tile have been read by everybody
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= a.ntiles) break;
        const u64 tbase = (u64)tile * RN_TILE;
        const bool edge = tbase + RN_TILE >= N;
        uint4 x[RN_SUB];
        u32 E[RN_SUB], pv[RN_SUB];
        u32 hasmask = 0;
        u32 nxl[RN_SUB];   // (lane 63) the byte behind the wave's last group
#pragma unroll
        for (int s = 0; s < RN_SUB; s++) {   // all of the tile's loads first: one memory latency, not RN_SUB in a row
            const u64 p0 = tbase + (u64)s * RN_SUBSZ + (u64)tid * 16;
            x[s] = make_uint4(0, 0, 0, 0);
            nxl[s] = 0;
            if (!edge || p0 < N) x[s] = *reinterpret_cast<const uint4 *>(a.src + p0);
            if (lane == 63 && (!edge || p0 + 16 < N)) nxl[s] = (u32)a.src[p0 + 16];
        }
#pragma unroll
        for (int s = 0; s < RN_SUB; s++) {
            const u64 p0 = tbase + (u64)s * RN_SUBSZ + (u64)tid * 16;
            const uint4 q = x[s];
            u32 nx = __shfl_down(q.x, 1, 64);
            if (lane == 63) nx = nxl[s];
            const u32 y0 = __builtin_amdgcn_alignbyte(q.y, q.x, 1), y1 = __builtin_amdgcn_alignbyte(q.z, q.y, 1);
            const u32 y2 = __builtin_amdgcn_alignbyte(q.w, q.z, 1), y3 = __builtin_amdgcn_alignbyte(nx, q.w, 1);
            u32 e = nz_bytes4(q.x ^ y0) | (nz_bytes4(q.y ^ y1) << 4) | (nz_bytes4(q.z ^ y2) << 8) | (nz_bytes4(q.w ^ y3) << 12);
            if (edge) {
                const u32 valid = p0 >= N ? 0u : (N - p0 >= 16 ? 0xffffu : ((1u << (u32)(N - p0)) - 1u));
                e &= valid;
                if (p0 < N && N - p0 <= 16) e |= 1u << (u32)(N - 1 - p0);
            }
            x[s] = q;
            E[s] = e;
            const u32 last1 = e ? (u32)p0 + 32u - (u32)__builtin_clz(e) : 0u;
            const u64 m = __ballot(e != 0);
            const u64 pm = m & lanemask_lt();
            const int src = pm ? 63 - __builtin_clzll(pm) : 0;
            pv[s] = __shfl(last1, src, 64);
            if (pm) hasmask |= 1u << s;
            const u32 wlast = m ? __shfl(last1, 63 - __builtin_clzll(m), 64) : 0u;
            if (lane == 0) s_last[s * NW + w] = wlast;
        }
        __syncthreads();
        if (w == 0) {
            const u32 v = lane < RN_NSEG ? s_last[lane] : 0u;
            const u32 inc = seg_incl_scan<OpMax>(v);
            const u32 agg = __shfl(inc, RN_NSEG - 1, 64);
            u32 ex = __shfl_up(inc, 1, 64);
            if (lane == 0) ex = 0;
            const u32 tin = rn_last_end_before(a.src, tbase, tile, agg, a.status_a, a.err);
            if (lane < RN_NSEG) s_carry[lane] = ex > tin ? ex : tin;
        }
        __syncthreads();
        u32 cnt[RN_SUB], inc[RN_SUB];
#pragma unroll
        for (int s = 0; s < RN_SUB; s++) {
            cnt[s] = (u32)__popc(E[s]);
            inc[s] = wave_incl_sum(cnt[s]);
            if (lane == 63) s_sum[s * NW + w] = inc[s];
        }
        __syncthreads();
        u32 excl[RN_SUB], truns;
        {
            const u32 v = lane < RN_NSEG ? s_sum[lane] : 0u;
            const u32 sc = seg_incl_scan<OpSum>(v);
            truns = __shfl(sc, RN_NSEG - 1, 64);
#pragma unroll
            for (int s = 0; s < RN_SUB; s++) {
                const int g = s * NW + w;
                const u32 before = g ? __shfl(sc, g - 1, 64) : 0u;
                excl[s] = before + inc[s] - cnt[s];
            }
        }
        if (w == 1) {
            const u64 e = lb_exclusive<OpSum>(a.status_b, tile, (u64)truns, a.err);
            if (lane == 0) {
                s_pref = e;
                if (tile + 1 == a.ntiles) a.scalars[2] = e + truns;
            }
        }
        __syncthreads();
        const u64 e0 = s_pref;
        const u32 sh = (u32)(e0 & 7u);
#pragma unroll
        for (int s = 0; s < RN_SUB; s++) {
            const u32 p0 = (u32)(tbase + (u64)s * RN_SUBSZ + (u64)tid * 16);
            const u32 xs[4] = {x[s].x, x[s].y, x[s].z, x[s].w};
            u32 prev = ((hasmask >> s) & 1u) ? pv[s] : s_carry[s * NW + w];
            u32 j = excl[s] + sh;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if ((E[s] >> i) & 1u) {
                    const u32 pos1 = p0 + (u32)i + 1u;
                    const u32 c = pos1 - prev;
                    prev = pos1;
                    const u32 v = (xs[i >> 2] >> (8 * (i & 3))) & 0xffu;
                    const u32 cc = c < LONGC ? c : LONGC;
                    if (PACK4) {
                        s_c8[j] = (u8)((v & 15u) | (cc << 4));
                    } else {
                        s_c8[j] = (u8)cc;
                        s_v8[j] = (u8)v;
                    }
                    if (c >= LONGC && e0 + (j - sh) < a.cap) a.counts[e0 + (j - sh)] = c;
                    j++;
                }
            }
        }
        __syncthreads();
        // groups of 8 staged runs: whole groups inside [sh, sh + truns) and below the capacity leave as three 16-byte
        // stores (unless one of them is a long run, whose length its owner has stored), the rest run by run
        const u64 gbase = e0 - sh;                       // run index of staged slot 0 (a multiple of 8)
        const u32 ngroups = (sh + truns + 7u) >> 3;
        for (u32 q = tid; q < ngroups; q += RN_NT) {
            const u32 lo = 8u * q;
            const u64 cw = *reinterpret_cast<const u64 *>(s_c8 + lo);
            const u64 vw = PACK4 ? 0ull : *reinterpret_cast<const u64 *>(s_v8 + lo);
            u32 c[8], v[8];
            bool plain = a.wide && lo >= sh && lo + 8u <= sh + truns && gbase + lo + 8u <= a.cap;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u32 b = (u32)(cw >> (8 * k)) & 255u;
                c[k] = PACK4 ? b >> 4 : b;
                v[k] = PACK4 ? b & 15u : (u32)(vw >> (8 * k)) & 255u;
                plain = plain && c[k] != LONGC;
            }
            if (plain) {
                uint4 *pc = reinterpret_cast<uint4 *>(a.counts + gbase + lo);
                pc[0] = make_uint4(c[0], c[1], c[2], c[3]);
                pc[1] = make_uint4(c[4], c[5], c[6], c[7]);
                *reinterpret_cast<uint4 *>(a.vals + gbase + lo) =
                    make_uint4(v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16));
            } else {
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const u32 i = lo + (u32)k;
                    const u64 g = gbase + i;
                    if (i >= sh && i < sh + truns && g < a.cap) {
                        if (c[k] != LONGC) a.counts[g] = c[k];
                        a.vals[g] = (u16)v[k];
                    }
                }
            }
        }
    }
}

// ---- MTF and RLE of a DNA-like record in ONE kernel (sigma <= 8, the fused encode) ------------------------------
// seqToMTF (MTF/Internal.hs:128-175) as mtf_nib_apply_kernel<.., SMALL> does it -- a tile of 32768 symbols, a lane per
// 128-symbol chunk, the list in one 32-bit register, the tile's incoming list by a backward scan of the last column --
// and then, with the tile's ranks still in LDS, seqToRLE of them (RLE/Internal.hs:104-153) as rle_blk_kernel does it:
// the index stream (1 GiB written, 1 GiB read again per 1 GiB record) never exists.  The successor of a tile's last
// rank is the rank of the next tile's first symbol in this tile's OUTGOING list (incoming list, then the tile's
// summary).  Tiles come by ticket; the last run end before a tile and the runs before it by look-back (the ranks in
// front of the tile are in no memory, so the direct probe of rle_blk_kernel is not available here).
struct MtfRleArgs {
    BwtAcc acc;
    u64 N;
    Lut8 lut;
    u32 sigma;
    u32 *flag;       // raised when a tile's incoming list could not be recovered by the backward scan
    u32 *counts;
    u16 *vals;
    u64 cap;
    u64 *status_a, *status_b;
    u32 *ticket;
    u64 *scalars;    // [2] total runs
    u32 *err;
    u32 ntiles;
    u32 wide;
    // NIB = true (the container of a record over <= 6 symbols): the runs leave as the nibble stream of rle_nib_kernel
    u8 *out;         // 16-byte aligned, ZERO for cap_units * 16 bytes
    u64 cap_units;
    u32 *esc;
    u64 esc_cap;
    u64 *totals;     // [0] runs (atomicAdd), [1] nibbles, [2] escapes (written by the last tile)
};

// run-end bits of eight positions, one per nibble (bit 4 i + 3), as eight adjacent bits
__device__ __forceinline__ u32 nib_bits8(u32 e) {
    u32 x = (e >> 3) & 0x11111111u;
    x = (x | (x >> 3)) & 0x03030303u;
    x = (x | (x >> 6)) & 0x000F000Fu;
    return (x | (x >> 12)) & 0xFFu;
}

template <bool NIB>
__global__ __launch_bounds__(MTF_NT) void mtf_rle_kernel(MtfRleArgs a) {
    constexpr int NW = MTF_NT / 64, SUBS = MTF_TILE / (MTF_NT * 16), NSEG = SUBS * NW;
    static_assert(NSEG <= 64 && MTF_NT * MTF_STRIDE >= MTF_TILE + 16, "segment scan by one wave; the code image doubles as run staging");
    __shared__ __attribute__((aligned(16))) u8 s_code[MTF_NT * MTF_STRIDE];
    __shared__ u8 s_lut[260], s_lut11[260];
    __shared__ NibSumm s_w[NW];
    __shared__ u64 s_in;
    __shared__ u32 s_next, s_tile;
    __shared__ u32 s_last[NSEG], s_carry[NSEG], s_sum[NSEG], s_wruns[NW];
    __shared__ u64 s_pref;
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const u64 N = a.N;
    for (int i = tid; i < 257; i += MTF_NT) {
        s_lut[i] = a.lut.v[i];
        s_lut11[i] = (u8)((a.lut.v[i] & 7u) * 0x11u);
    }
    if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
    __syncthreads();
    const u32 tile = s_tile;
    if (tile >= a.ntiles) return;
#ifdef MTFRLE_PROFILE
    u32 tq[10];
#define MR_T(i) tq[i] = (u32)__builtin_readcyclecounter()
#else
#define MR_T(i)
#endif
    MR_T(0);
    const u64 base = (u64)tile * MTF_TILE;
    const bool edge = base + MTF_TILE >= N;
    // (wave 0 recovers the tile's incoming list, and thread 0 fetches the symbol behind the tile, while the tile's own
    // loads are in flight: their latencies overlap instead of following one another)
    u32 nsym = 0;
    nib_stage_hook(a.acc, N, base, s_lut11, s_code, [&]() {
        if (tid == 0 && base + MTF_TILE < N) nsym = (u32)(a.acc(base + MTF_TILE) + 1);
        if (tid < 64) {
            u64 l0 = NIB_IDENT;
            const bool ok = nib_list_before(a.acc, base, a.sigma, s_lut, &l0, a.flag);
            if (tid == 0) {
                s_in = l0;
                if (!ok) atomicOr(a.flag, 1u);
            }
        }
    });
    __syncthreads();
    MR_T(1);
    // ---- MTF: one pass per chunk from the identity list; first occurrences replayed from the true incoming list
    u32 *cw = reinterpret_cast<u32 *>(s_code + tid * MTF_STRIDE);
    NibSumm mine{NIB_IDENT, 0u};
    u64 evp = 0;
    u32 evc = 0, k4 = 0;
    {
        u32 lst = (u32)NIB_IDENT, seen = 0;
        if (edge) nib8_chunk_ranks<true>(cw, lst, seen, evc, evp, k4);
        else nib8_chunk_ranks<false>(cw, lst, seen, evc, evp, k4);
        mine.perm = (NIB_IDENT & 0xFFFFFFFF00000000ull) | (u64)lst;
        mine.mask = seen;
    }
    MR_T(2);
    NibSumm agg{NIB_IDENT, 0u};
    const NibSumm exc = nib_block_excl<true>(mine, s_w, &agg);
    {
        u32 list = (u32)nib8_combine(NibSumm{s_in, 0u}, exc).perm;   // (sigma <= 8: every list lives in 32 bits)
        for (u32 e4 = 0; e4 < k4; e4 += 4) {
            const u32 c = (evc >> e4) & 15u;
            const u32 pp = (u32)(evp >> (2 * e4)) & 255u;
            const u32 pos = nib8_find(list, c);
            list = nib8_front(list, pos, c);
            const u32 shf = 4u * (pp & 7u);
            cw[pp >> 3] = (cw[pp >> 3] & ~(15u << shf)) | (pos << shf);
        }
    }
    if (tid == 0) {   // the rank the next tile's first symbol will get: the successor of this tile's last rank
        u32 nx = 0x100u;
        if (base + MTF_TILE < N) {
            const u32 outl = (u32)nib8_combine(NibSumm{s_in, 0u}, agg).perm;
            nx = nib8_find(outl, (u32)s_lut[nsym]);
        }
        s_next = nx;
    }
    __syncthreads();
    MR_T(3);
    // ---- RLE of the tile's ranks (rle_blk_kernel's phases on nibbles; a thread owns 16 consecutive ranks per sub-tile)
    auto nibw = [&](u32 p) -> const u32 * {
        return reinterpret_cast<const u32 *>(s_code + (p / MTF_CH) * MTF_STRIDE + (p % MTF_CH) / 2);
    };
    u32 n0[SUBS], n1[SUBS], E0[SUBS], E1[SUBS], pv[SUBS];
    u32 hasmask = 0, myruns = 0;
#pragma unroll
    for (int s = 0; s < SUBS; s++) {
        const u32 pl = (u32)s * (MTF_NT * 16) + (u32)tid * 16;       // tile-local position of the group
        const u64 p0 = base + pl;
        const u32 *gp = nibw(pl);
        const u32 a0 = gp[0], a1 = gp[1];
        const u32 nx = pl + 16 < MTF_TILE ? nibw(pl + 16)[0] : s_next;
        const u32 d0 = a0 ^ __builtin_amdgcn_alignbit(a1, a0, 4), d1 = a1 ^ __builtin_amdgcn_alignbit(nx, a1, 4);
        u32 e0 = (((d0 & 0x77777777u) + 0x77777777u) | d0) & 0x88888888u;   // bit 4 i + 3: rank i differs from its successor
        u32 e1 = (((d1 & 0x77777777u) + 0x77777777u) | d1) & 0x88888888u;
        if (pl + 16 == MTF_TILE && (s_next & 0x100u)) e1 |= 0x80000000u;      // no successor: the stream ends with this tile
        if (edge) {
            const u32 nv = p0 >= N ? 0u : (N - p0 >= 16 ? 16u : (u32)(N - p0));
            const u32 v0 = nv >= 8 ? 8u : nv, v1 = nv - v0;
            e0 &= v0 >= 8 ? 0xffffffffu : ((1u << (4 * v0)) - 1u);
            e1 &= v1 >= 8 ? 0xffffffffu : ((1u << (4 * v1)) - 1u);
            if (nv && N - p0 <= 16) {                                          // the last position ends its run
                if (nv <= 8) e0 |= 8u << (4 * (nv - 1));
                else e1 |= 8u << (4 * (nv - 9));
            }
        }
        n0[s] = a0; n1[s] = a1;
        E0[s] = e0; E1[s] = e1;
        myruns += (u32)__popc(e0) + (u32)__popc(e1);
        const u32 last1 = e1 ? (u32)p0 + 16u - ((u32)__builtin_clz(e1) >> 2)
                             : (e0 ? (u32)p0 + 8u - ((u32)__builtin_clz(e0) >> 2) : 0u);   // 1 + position of the last end
        const u64 m = __ballot((e0 | e1) != 0);
        const u64 pm = m & lanemask_lt();
        const int src = pm ? 63 - __builtin_clzll(pm) : 0;
        pv[s] = __shfl(last1, src, 64);
        if (pm) hasmask |= 1u << s;
        const u32 wlast = m ? __shfl(last1, 63 - __builtin_clzll(m), 64) : 0u;
        if (lane == 0) s_last[s * NW + w] = wlast;
    }
    {   // the tile's run count, for the look-back over the run counts to start together with the one over the run ends
        const u32 wr = wave_incl_sum(myruns);
        if (lane == 63) s_wruns[w] = wr;
    }
    __syncthreads();
    MR_T(4);
    if constexpr (NIB) {
        // ---- the nibble stream (rle_nib_kernel's phases 2-4 on this tile's ranks; the image takes the place of the ranks)
        u64 *img = reinterpret_cast<u64 *>(s_code);
        static_assert(MTF_NT * MTF_STRIDE >= (MTF_TILE / 16 + 4) * 8, "a tile emits at most one nibble per position");
        for (int i = tid; i < MTF_TILE / 16 + 4; i += MTF_NT) img[i] = 0;   // (every thread has its ranks in registers)
        if (w == 0) {
            const u32 v = lane < NSEG ? s_last[lane] : 0u;
            const u32 inc = seg_incl_scan<OpMax>(v);
            const u32 aggl = __shfl(inc, NSEG - 1, 64);
            u32 ex = __shfl_up(inc, 1, 64);
            if (lane == 0) ex = 0;
            const u32 tin = (u32)lb_exclusive_last(a.status_a, tile, aggl, a.err);
            if (lane < NSEG) s_carry[lane] = ex > tin ? ex : tin;
        }
        __syncthreads();
        u32 E16[SUBS], prev1[SUBS], cnt[SUBS], inc[SUBS];
        u32 runs_mine = 0;
#pragma unroll
        for (int s = 0; s < SUBS; s++) {
            const u32 p0 = (u32)(base + (u64)s * (MTF_NT * 16) + (u64)tid * 16);
            const u32 pe = ((hasmask >> s) & 1u) ? pv[s] : s_carry[s * NW + w];
            prev1[s] = pe;
            E16[s] = nib_bits8(E0[s]) | (nib_bits8(E1[s]) << 8);
            // the last end before the group is position pe - 1 (pe == 0: the virtual end before position 0)
            const u32 back = p0 + 1u - pe;                       // 1: directly before the group, 2, 3, 4 ...
            const u32 vb = (back - 1u < 4u) ? 1u << (4u - back) : 0u;
            const u32 X = (E16[s] << 4) | vb;
            const u32 g2 = X & ~(X << 1);
            const u32 g3 = g2 & ~(X << 2);
            const u32 g5 = g3 & ~(X << 3) & ~(X << 4);
            const u32 runs = (u32)__popc(E16[s]);
            runs_mine += runs;
            cnt[s] = (runs + (u32)__popc((g3 >> 4) & 0xffffu)) | ((u32)__popc((g5 >> 4) & 0xffffu) << 16);
            inc[s] = wave_incl_sum(cnt[s]);
            if (lane == 63) s_sum[s * NW + w] = inc[s];
        }
        runs_mine = wave_sum(runs_mine);
        if (lane == 0) s_wruns[w] = runs_mine;
        __syncthreads();
        u32 excl[SUBS], tile_cnt;
        {
            const u32 v = lane < NSEG ? s_sum[lane] : 0u;
            const u32 sc = seg_incl_scan<OpSum>(v);
            tile_cnt = __shfl(sc, NSEG - 1, 64);
#pragma unroll
            for (int s = 0; s < SUBS; s++) {
                const int g = s * NW + w;
                const u32 before = g ? __shfl(sc, g - 1, 64) : 0u;
                excl[s] = before + inc[s] - cnt[s];
            }
        }
        const u32 tn = tile_cnt & 0xffffu, te = tile_cnt >> 16;
        if (w == 1) {
            const u64 e = lb_exclusive<OpSum>(a.status_b, tile, NIB_LB(tn, te), a.err);
            if (lane == 0) {
                s_pref = e;
                if (tile + 1 == a.ntiles) {
                    a.totals[1] = (e >> NIB_LB_SHIFT) + tn;
                    a.totals[2] = NIB_LB_ESC(e) + te;
                }
            }
        } else if (tid == 0) {
            u32 r = 0;
#pragma unroll
            for (int i = 0; i < NW; i++) r += s_wruns[i];
            if (r) atomicAdd((unsigned long long *)&a.totals[0], (unsigned long long)r);
        }
        __syncthreads();
        const u64 q0 = s_pref >> NIB_LB_SHIFT, e0 = NIB_LB_ESC(s_pref);
        const u32 a0 = (u32)(q0 & 31u);
#pragma unroll
        for (int s = 0; s < SUBS; s++) {
            const u32 p0 = (u32)(base + (u64)s * (MTF_NT * 16) + (u64)tid * 16);
            u32 prev = prev1[s];
            u64 L = 0;
            u32 H = 0, len = 0;     // the group's string: at most 17 nibbles
#pragma unroll
            for (int d = 0; d < 4; d++) {
                u32 g = 0, sh = 0;  // the nibbles of four positions: at most five
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const int i = 4 * d + k;
                    const bool end = (E16[s] >> i) & 1u;
                    const u32 pos1 = p0 + (u32)i + 1u;
                    const u32 c1 = pos1 - prev - 1u;                      // run length - 1
                    const u32 v = ((i < 8 ? n0[s] : n1[s]) >> (4 * (i & 7))) & 15u;
                    const u32 first = v + ((c1 - 1u < 3u) ? 6u : 0u);     // lengths 2, 3, 4: value + 6
                    const u32 second = (c1 < 4u ? c1 : 4u) + 10u;         // 3 -> 12, 4 -> 13, >= 5 -> 14
                    const bool two = c1 >= 2u;
                    const u32 pair = first | (two ? second << 4 : 0u);
                    if (end) {
                        g |= pair << sh;
                        sh += two ? 8u : 4u;
                        prev = pos1;
                    }
                }
                const u32 bsh = len * 4u;
                if (len < 16u) {
                    L |= (u64)g << bsh;
                    if (bsh > 32u) H |= g >> (64u - bsh);
                } else {
                    H |= g << (bsh - 64u);
                }
                len += sh >> 2;
            }
            const u32 off = a0 + (excl[s] & 0xffffu);
            const u32 wd = off >> 4, bs = (off & 15u) * 4u;
            const u64 w0 = L << bs;
            const u64 w1 = (bs ? L >> (64u - bs) : 0ull) | ((u64)H << bs);
            if (w0) atomicOr((unsigned long long *)&img[wd], (unsigned long long)w0);
            if (w1) atomicOr((unsigned long long *)&img[wd + 1], (unsigned long long)w1);
            if (cnt[s] >> 16) {   // rare: lengths >= 5 go to the escape list, in run order
                u64 e = e0 + (excl[s] >> 16);
                u32 pr = prev1[s], em = E16[s];
                while (em) {
                    const u32 i = (u32)__builtin_ctz(em);
                    em &= em - 1u;
                    const u32 c = p0 + i + 1u - pr;
                    pr = p0 + i + 1u;
                    if (c >= 5u) {
                        if (e < a.esc_cap) a.esc[e] = c;
                        e++;
                    }
                }
            }
        }
        __syncthreads();
        nib_image_out<MTF_NT>(img, q0, tn, tile + 1 == a.ntiles, a.out, a.cap_units);
        return;
    }
    if (w == 0) {
        const u32 v = lane < NSEG ? s_last[lane] : 0u;
        const u32 inc = seg_incl_scan<OpMax>(v);
        const u32 aggl = __shfl(inc, NSEG - 1, 64);
        u32 ex = __shfl_up(inc, 1, 64);
        if (lane == 0) ex = 0;
        const u32 tin = (u32)lb_exclusive_last(a.status_a, tile, aggl, a.err);
        if (lane < NSEG) s_carry[lane] = ex > tin ? ex : tin;
    }
    if (w == 1) {
        u32 tr = 0;
#pragma unroll
        for (int i = 0; i < NW; i++) tr += s_wruns[i];
        const u64 e = lb_exclusive<OpSum>(a.status_b, tile, (u64)tr, a.err);
        if (lane == 0) {
            s_pref = e;
            if (tile + 1 == a.ntiles) a.scalars[2] = e + tr;
        }
    }
    u32 cnt[SUBS], inc[SUBS];
#pragma unroll
    for (int s = 0; s < SUBS; s++) {
        cnt[s] = (u32)__popc(E0[s]) + (u32)__popc(E1[s]);
        inc[s] = wave_incl_sum(cnt[s]);
        if (lane == 63) s_sum[s * NW + w] = inc[s];
    }
    __syncthreads();
    MR_T(5);
    u32 excl[SUBS], truns;
    {
        const u32 v = lane < NSEG ? s_sum[lane] : 0u;
        const u32 sc = seg_incl_scan<OpSum>(v);
        truns = __shfl(sc, NSEG - 1, 64);
#pragma unroll
        for (int s = 0; s < SUBS; s++) {
            const int g = s * NW + w;
            const u32 before = g ? __shfl(sc, g - 1, 64) : 0u;
            excl[s] = before + inc[s] - cnt[s];
        }
    }
    // (no barrier here: the runs before the tile were published before the last one, and every thread has had its
    // ranks in registers since the first barrier of this part -- the image becomes the run staging area)
    MR_T(6);
    const u64 e0g = s_pref;
    const u32 sh = (u32)(e0g & 7u);
    const u64 gbase = e0g - sh;
    const u32 lim = gbase >= a.cap ? 0u : (a.cap - gbase > 0xffffffffull ? 0xffffffffu : (u32)(a.cap - gbase));
    u8 *s_c8 = s_code;
#pragma unroll
    for (int s = 0; s < SUBS; s++) {
        if (!cnt[s]) continue;
        const u32 p0 = (u32)(base + (u64)s * (MTF_NT * 16) + (u64)tid * 16);
        const u32 prev = ((hasmask >> s) & 1u) ? pv[s] : s_carry[s * NW + w];
        const int prel0 = (int)(prev - p0);                // <= 0: the previous end, relative to the group
        int prel = prel0;
        const u32 j0 = excl[s] + sh;
        u32 j = j0;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const u32 ek = i < 8 ? E0[s] : E1[s], nk = i < 8 ? n0[s] : n1[s];
            if ((ek >> (4 * (i & 7) + 3)) & 1u) {
                const u32 c = (u32)(i + 1 - prel);         // (>= 15 only for the group's first run, or bits 0 and 15 alone)
                prel = i + 1;
                s_c8[j] = (u8)(((nk >> (4 * (i & 7))) & 15u) | (c << 4));
                j++;
            }
        }
        const u32 fi = E0[s] ? (u32)__builtin_ctz(E0[s]) >> 2 : 8u + ((u32)__builtin_ctz(E1[s]) >> 2);
        const u32 firstc = fi + 1u - (u32)prel0;
        if (firstc >= 15u) {                               // a long run: count nibble 15, the count itself by its owner
            s_c8[j0] |= 0xF0u;
            if (j0 < lim) a.counts[gbase + j0] = firstc;
        }
        if (E0[s] == 0x8u && E1[s] == 0x80000000u) {
            s_c8[j0 + 1] |= 0xF0u;
            if (j0 + 1 < lim) a.counts[gbase + j0 + 1] = 15u;
        }
    }
    __syncthreads();
    MR_T(7);
    const u32 ngroups = (sh + truns + 7u) >> 3;
    for (u32 q = tid; q < ngroups; q += MTF_NT) {
        const u32 lo = 8u * q;
        const uint2 cwd = *reinterpret_cast<const uint2 *>(s_c8 + lo);
        const u32 tx = cwd.x & 0xF0F0F0F0u, ty = cwd.y & 0xF0F0F0F0u;
        // a count nibble of 15  <=>  a zero byte in t ^ 0xF0F0F0F0
        const u32 zx = tx ^ 0xF0F0F0F0u, zy = ty ^ 0xF0F0F0F0u;
        const bool any15 = (((zx - 0x01010101u) & ~zx & 0x80808080u) | ((zy - 0x01010101u) & ~zy & 0x80808080u)) != 0;
        const bool plain = a.wide && !any15 && lo >= sh && lo + 8u <= sh + truns && lo + 8u <= lim;
        if (plain) {
            uint4 *pc = reinterpret_cast<uint4 *>(a.counts + gbase + lo);
            pc[0] = make_uint4((cwd.x >> 4) & 15u, (cwd.x >> 12) & 15u, (cwd.x >> 20) & 15u, cwd.x >> 28);
            pc[1] = make_uint4((cwd.y >> 4) & 15u, (cwd.y >> 12) & 15u, (cwd.y >> 20) & 15u, cwd.y >> 28);
            *reinterpret_cast<uint4 *>(a.vals + gbase + lo) =
                make_uint4(__builtin_amdgcn_perm(0u, cwd.x, 0x0c010c00u) & 0x000F000Fu, __builtin_amdgcn_perm(0u, cwd.x, 0x0c030c02u) & 0x000F000Fu,
                           __builtin_amdgcn_perm(0u, cwd.y, 0x0c010c00u) & 0x000F000Fu, __builtin_amdgcn_perm(0u, cwd.y, 0x0c030c02u) & 0x000F000Fu);
        } else {
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const u32 bb = ((k < 4 ? cwd.x : cwd.y) >> (8 * (k & 3))) & 255u;
                const u32 i = lo + (u32)k;
                if (i >= sh && i < sh + truns && i < lim) {
                    if ((bb >> 4) != 15u) a.counts[gbase + i] = bb >> 4;
                    a.vals[gbase + i] = (u16)(bb & 15u);
                }
            }
        }
    }
#ifdef MTFRLE_PROFILE
    MR_T(8);
    if (tid == 0 && (tile & 63u) == 7u) {
        u64 *dbg = a.scalars + 112;
        for (int i = 0; i < 8; i++) atomicAdd((unsigned long long *)&dbg[i], (unsigned long long)(tq[i + 1] - tq[i]));
        atomicAdd((unsigned long long *)&dbg[8], 1ull);
    }
#endif
}

#define UP_NT 256
#define UP_NPT 32                   // nibbles per thread = one 16-byte unit
#define UP_TILE_UNITS UP_NT         // 4 KB of body per tile

struct UnpackNibArgs {
    const u8 *body;
    u64 units;
    const u32 *esc;
    u64 nesc;
    u64 nruns;
    u32 *cnt;
    u16 *val;
    u64 *status;
    u32 *ticket;
    u32 *err;
    u32 ntiles;
};

__global__ __launch_bounds__(UP_NT) void unpack_nib_kernel(UnpackNibArgs a) {
    __shared__ u32 s_first[UP_NT + 1];
    __shared__ u32 sm[UP_NT / 64 + 1];
    __shared__ u32 s_tile;
    __shared__ u64 s_excl;
    __shared__ u32 s_cnt[UP_NT * UP_NPT];   // a tile holds at most one run per nibble
    __shared__ u8 s_val[UP_NT * UP_NPT];
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    while (true) {
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= a.ntiles) break;
        const u64 unit = (u64)tile * UP_TILE_UNITS + tid;
        u64 w0 = ~0ull, w1 = ~0ull;
        if (unit < a.units) {
            const ulonglong2 t = reinterpret_cast<const ulonglong2 *>(a.body)[unit];
            w0 = t.x; w1 = t.y;
        }
        s_first[tid] = (u32)(w0 & 15u);
        if (tid == UP_NT - 1) s_first[UP_NT] = unit + 1 < a.units ? (u32)(a.body[(unit + 1) * 16] & 15u) : 15u;
        const u64 m = 0x1111111111111111ull;
        const u64 ns0 = (w0 >> 3) & (w0 >> 2) & m, ns1 = (w1 >> 3) & (w1 >> 2) & m;   // codes >= 12
        const u32 starts = (u32)__popcll(~ns0 & m) + (u32)__popcll(~ns1 & m);
        const u32 escs = (u32)__popcll(ns0 & (w0 >> 1) & ~w0) + (u32)__popcll(ns1 & (w1 >> 1) & ~w1);
        u32 total;
        const u32 x = starts | (escs << 16);
        const u32 ex = block_excl_sum<UP_NT>(x, sm, &total);
        if (w == 0) {
            const u64 e = lb_exclusive<OpSum>(a.status, tile, ((u64)(total & 0xffffu) << 31) | (total >> 16), a.err);
            if (lane == 0) s_excl = e;
        }
        __syncthreads();
        const u64 kbase = s_excl >> 31;          // first run of this tile
        u32 kl = ex & 0xffffu;                   // tile-local run index
        u64 e = (s_excl & 0x7fffffffull) + (ex >> 16);
        if ((w0 & 15u) == 14u) e++;  // belongs to the run that starts in the previous unit
        const u32 nextfirst = s_first[tid + 1];
        // runs are staged tile-locally (count | value << 28 would not hold escaped counts: two arrays)
#pragma unroll
        for (int p = 0; p < UP_NPT; p++) {
            const u32 code = (u32)((p < 16 ? w0 >> (4 * p) : w1 >> (4 * (p - 16))) & 15u);
            if (code < 12u) {
                const u32 nx = p == 31 ? nextfirst : (u32)((p + 1 < 16 ? w0 >> (4 * (p + 1)) : w1 >> (4 * (p - 15))) & 15u);
                u32 c = code >= 6u ? 2u : 1u;
                const u32 v = code >= 6u ? code - 6u : code;
                if (nx == 12u) c = 3u;
                else if (nx == 13u) c = 4u;
                else if (nx == 14u) {
                    c = e < a.nesc ? a.esc[e] : 0u;
                    e++;
                }
                s_cnt[kl] = c;
                s_val[kl] = (u8)v;
                kl++;
            }
        }
        __syncthreads();
        // coalesced write-out of the tile's runs
        const u32 truns = total & 0xffffu;
        for (u32 i = tid; i < truns; i += UP_NT) {
            const u64 k = kbase + i;
            if (k < a.nruns) {
                a.cnt[k] = s_cnt[i];
                a.val[k] = (u16)s_val[i];
            }
        }
        __syncthreads();  // s_first / s_excl reuse
    }
}

// ---- byte formats (6 < sigma) ------------------------------------------------------
// bytes per run: 1 (sigma <= 16) or 2 (value byte, count byte with escape; the ninth value bit
// of sigma = 257 rides in the count byte's top bit -> counts escape at 127)
// byte formats.  A block owns PR_TILE consecutive runs (a thread 8 of them); escapes are listed in
// run order -- pass 1 counts them per block, a scan gives every block its slot, pass 2 writes -- so
// that the packed bytes are a function of the runs alone.
#define PR_TILE 2048
__device__ __forceinline__ bool pr_escapes(int bpr, u32 c) { return bpr == 1 ? c >= 15 : c >= 127; }
__global__ __launch_bounds__(256) void pack_runs_count_kernel(const u32 *__restrict__ cnt, u64 nruns, int bpr,
                                                              u64 *__restrict__ tcnt) {
    __shared__ u32 s[4];
    const u64 base = (u64)blockIdx.x * PR_TILE;
    u32 e = 0;
    for (int k = 0; k < 8; k++) {
        const u64 i = base + (u64)k * 256 + threadIdx.x;
        if (i < nruns && pr_escapes(bpr, cnt[i])) e++;
    }
    for (int d = 32; d >= 1; d >>= 1) e += __shfl_xor(e, d, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) tcnt[blockIdx.x] = (u64)s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void pack_runs_kernel(const u32 *__restrict__ cnt,
                                                        const u16 *__restrict__ val, u64 nruns,
                                                        int bpr, u8 *__restrict__ out,
                                                        u32 *__restrict__ esc, const u64 *__restrict__ tcnt,
                                                        u64 esc_cap) {
    __shared__ u32 s[4];
    const u64 base = (u64)blockIdx.x * PR_TILE + (u64)threadIdx.x * 8;
    u32 c[8], ne = 0;
    for (int k = 0; k < 8; k++) {
        const u64 i = base + k;
        c[k] = 0;
        if (i >= nruns) continue;
        c[k] = cnt[i];
        const u32 v = val[i];
        const bool e = pr_escapes(bpr, c[k]);
        ne += e;
        if (bpr == 1) {
            out[i] = (u8)((v & 15u) | ((e ? 15u : c[k]) << 4));
        } else {
            out[2 * i] = (u8)v;
            out[2 * i + 1] = (u8)((e ? 127u : c[k]) | ((v >> 8) << 7));
        }
    }
    u32 inc = ne;
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(inc, d, 64);
        if ((int)(threadIdx.x & 63) >= d) inc += t;
    }
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = inc;
    __syncthreads();
    u64 slot = tcnt[blockIdx.x] + (inc - ne);
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) slot += s[w];
    if (ne) {
        for (int k = 0; k < 8; k++) {
            if (base + k < nruns && pr_escapes(bpr, c[k])) {
                if (slot < esc_cap) {
                    esc[2 * slot] = (u32)(base + k);
                    esc[2 * slot + 1] = c[k];
                }
                slot++;
            }
        }
    }
}
__global__ __launch_bounds__(256) void unpack_runs_kernel(const u8 *__restrict__ in, u64 nruns, int bpr,
                                                          u32 *__restrict__ cnt, u16 *__restrict__ val) {
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < nruns; k += (u64)gridDim.x * 256) {
        if (bpr == 1) {
            u8 b = in[k];
            val[k] = (u16)(b & 15);
            cnt[k] = (u32)(b >> 4);
        } else {
            u8 lo = in[2 * k], hi = in[2 * k + 1];
            val[k] = (u16)(lo | ((hi >> 7) << 8));
            cnt[k] = (u32)(hi & 127);
        }
    }
}
__global__ __launch_bounds__(256) void unpack_esc_kernel(const u32 *__restrict__ esc, u64 nesc,
                                                         u64 nruns, u32 *__restrict__ cnt) {
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < nesc && esc[2 * i] < nruns) cnt[esc[2 * i]] = esc[2 * i + 1];
}
