// tc_seg.hpp -- the sort of a prefix-doubling round as a SEGMENTED sort (round 4).
//
// A doubling round of createSuffixArray's replacement (reference BWT/Internal.hs:110-134; tc_sa.hpp) orders the
// members of every tied group by rank[i + h].  The active set arrives in SA order, so the groups are contiguous
// runs of equal `grp` (the top 32 bits of key2 = grp << 32 | rank): what the round needs is a sort INSIDE each
// run by one 32-bit value -- not the eight stable LSD passes over (group, rank) the rounds took until now
// (each a full read and write of 12 bytes per member at 0.34-0.39 of the HBM peak).  Here:
//   * a run of at most SEG_CAP members is sorted in LDS by the workgroup whose window holds its head: one read
//     and one write of the member (seg_small_kernel: a bitonic network over 4096 composite keys
//     (run ordinal, rank, source slot) -- the ordinal keeps every run in its own slots, so the network needs no
//     knowledge of the run boundaries);
//   * a longer run is PARTITIONED by 8 bits of the rank, most significant first (seg_count / seg_scan /
//     seg_scatter): the children are ordered among themselves, each child is a new run; children that are
//     still longer go through the next level (at most 4 levels: then all 32 bits are used and the members of a
//     child are equal -- a tied group of the next round, which needs no sort at all).  A partition level is
//     unstable (one LDS atomic per member gives its place), takes its positions from a per-run histogram, and
//     moves a run between the two buffers; `ybits` records, per slot, which buffer holds it.
// Results land in the caller's primary buffers (keys / vals), every slot written exactly once by the small kernel.
#pragma once
#include "tc_common.hpp"

#define SEG_W 4096                    // slots of a window (LDS image)
#define SEG_CAP 1024                  // runs up to this length are sorted in LDS
#define SEG_SPAN (SEG_W - SEG_CAP)    // slots a window is responsible for (a multiple of 64)
#define SEG_NT 256
#define SEG_ITEMS (SEG_W / SEG_NT)    // 16
#define SEG_PT 4096                   // members per tile of a partition level
#define SEG_EXTW (SEG_CAP / 64 + SEG_W / 64 + 1)   // bit words a window looks at: [t0 - CAP, t0 + W]

struct SegBuffers {
    u64 *segbits;      // 1 bit per slot: a run starts here
    u64 *ybits;        // 1 bit per slot: the slot's valid copy is in the alternate buffers
    u32 *lstart[2];    // long runs of the current / next level
    u32 *lsize[2];     // bit 31: the run sits in the alternate buffers
    u32 *ltbase[2];    // first tile of the run
    u32 *tile_seg;     // tile -> run
    u32 *hist;         // [runs][256] digit counts, then cursors
    u32 *counters;     // [0..1] runs / tiles of level A, [2..3] of level B
    size_t cap_runs, cap_tiles;
};

static inline size_t seg_bit_words(u64 m) { return (size_t)(m / 64 + SEG_EXTW + 8); }

#ifdef __HIPCC__

// ---- run heads, long runs ---------------------------------------------------------------
// segbits, and the list of runs longer than SEG_CAP (the keys are sorted by their top 32 bits, so a run headed at k
// is long iff slot k + CAP still belongs to it; its end by a binary search)
__global__ __launch_bounds__(256) void seg_init_kernel(const u64 *__restrict__ keys, u32 m, u64 *__restrict__ segbits,
                                                       u32 nwords, u32 *__restrict__ lstart, u32 *__restrict__ lsize,
                                                       u32 *__restrict__ ltbase, u32 *__restrict__ counters, u32 cap_runs) {
    const u32 nw_used = (m + 63) / 64;
    for (u64 k0 = ((u64)blockIdx.x * 256 + threadIdx.x) & ~63ull; k0 < (u64)nwords * 64; k0 += (u64)gridDim.x * 256) {
        const u64 k = k0 + (threadIdx.x & 63);
        bool head = false;
        u32 g = 0;
        if (k < m) {
            g = (u32)(keys[k] >> 32);
            head = k == 0 || (u32)(keys[k - 1] >> 32) != g;
        }
        const u64 hb = __ballot(head);
        if ((threadIdx.x & 63) == 0 && (k0 >> 6) < nwords) segbits[k0 >> 6] = (k0 >> 6) < nw_used ? hb : 0ull;
        if (head && k + SEG_CAP < m && (u32)(keys[k + SEG_CAP] >> 32) == g) {
            u64 lo = k + SEG_CAP + 1, hi = m;   // first slot of another group
            while (lo < hi) {
                const u64 mid = (lo + hi) >> 1;
                if ((u32)(keys[mid] >> 32) == g) lo = mid + 1; else hi = mid;
            }
            const u32 size = (u32)(lo - k);
            const u32 s = atomicAdd(&counters[0], 1u);
            const u32 tb = atomicAdd(&counters[1], (size + SEG_PT - 1) / SEG_PT);
            if (s < cap_runs) {
                lstart[s] = (u32)k;
                lsize[s] = size;
                ltbase[s] = tb;
            }
        }
    }
}

// tile -> run (one wave per run)
__global__ __launch_bounds__(256) void seg_tilemap_kernel(const u32 *__restrict__ lsize, const u32 *__restrict__ ltbase,
                                                          u32 nruns, u32 *__restrict__ tile_seg, u32 cap_tiles) {
    const u32 nwaves = gridDim.x * 4;
    for (u32 s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nruns; s += nwaves) {
        const u32 nt = ((lsize[s] & 0x7fffffffu) + SEG_PT - 1) / SEG_PT, tb = ltbase[s];
        for (u32 j = threadIdx.x & 63; j < nt; j += 64)
            if (tb + j < cap_tiles) tile_seg[tb + j] = s;
    }
}

// digit counts of every long run: hist[run][digit]
__global__ __launch_bounds__(256) void seg_count_kernel(const u64 *__restrict__ kx, const u64 *__restrict__ ky,
                                                        const u32 *__restrict__ lstart, const u32 *__restrict__ lsize,
                                                        const u32 *__restrict__ ltbase, const u32 *__restrict__ tile_seg,
                                                        const u32 *__restrict__ counters, int shift, u32 *__restrict__ hist) {
    __shared__ u32 s_h[256];
    const u32 tile = blockIdx.x;
    if (tile >= counters[1]) return;
    const u32 s = tile_seg[tile];
    const u32 sz = lsize[s];
    const u64 *src = (sz >> 31) ? ky : kx;
    const u32 size = sz & 0x7fffffffu;
    const u32 off = (tile - ltbase[s]) * SEG_PT;
    const u32 cnt = size - off < SEG_PT ? size - off : SEG_PT;
    const u64 base = (u64)lstart[s] + off;
    s_h[threadIdx.x] = 0;
    __syncthreads();
    u64 v[SEG_PT / 256];
#pragma unroll
    for (int q = 0; q < SEG_PT / 256; q++) {
        const u32 p = q * 256 + threadIdx.x;
        v[q] = p < cnt ? src[base + p] : 0ull;
    }
#pragma unroll
    for (int q = 0; q < SEG_PT / 256; q++) {
        const u32 p = q * 256 + threadIdx.x;
        if (p < cnt) atomicAdd(&s_h[((u32)v[q] >> shift) & 255u], 1u);
    }
    __syncthreads();
    const u32 c = s_h[threadIdx.x];
    if (c) atomicAdd(&hist[(size_t)s * 256 + threadIdx.x], c);
}

// per run (one wave): counts -> cursors; the children's heads into segbits; children still longer than the cap onto the
// next level's list (not after the last level: its children hold equal ranks)
__global__ __launch_bounds__(256) void seg_scan_kernel(const u32 *__restrict__ lstart, const u32 *__restrict__ lsize,
                                                       const u32 *__restrict__ counters, u32 *__restrict__ hist,
                                                       u64 *__restrict__ segbits, int last_level,
                                                       u32 *__restrict__ nstart, u32 *__restrict__ nsize,
                                                       u32 *__restrict__ ntbase, u32 *__restrict__ ncounters, u32 cap_runs) {
    const u32 nruns = counters[0] < cap_runs ? counters[0] : cap_runs;
    const u32 nwaves = gridDim.x * 4, l = threadIdx.x & 63;
    for (u32 s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nruns; s += nwaves) {
        u32 *h = hist + (size_t)s * 256;
        const uint4 c = *reinterpret_cast<const uint4 *>(h + 4 * l);
        const u32 mine = c.x + c.y + c.z + c.w;
        const u32 excl = wave_incl_sum(mine) - mine;
        const u32 start = lstart[s];
        const u32 other = (~lsize[s]) & 0x80000000u;   // the children sit in the buffers this level writes
        const u32 cc[4] = {c.x, c.y, c.z, c.w};
        u32 run = start + excl;
        u32 cur[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            cur[q] = run;
            if (cc[q]) {
                atomicOr((unsigned long long *)&segbits[run >> 6], 1ull << (run & 63));
                if (cc[q] > SEG_CAP && !last_level) {
                    const u32 t = atomicAdd(&ncounters[0], 1u);
                    const u32 tb = atomicAdd(&ncounters[1], (cc[q] + SEG_PT - 1) / SEG_PT);
                    if (t < cap_runs) {
                        nstart[t] = run;
                        nsize[t] = cc[q] | other;
                        ntbase[t] = tb;
                    }
                }
            }
            run += cc[q];
        }
        *reinterpret_cast<uint4 *>(h + 4 * l) = make_uint4(cur[0], cur[1], cur[2], cur[3]);
    }
}

// one tile of a long run -> its children (unstable: the order inside a child is settled later); the tile's slots change
// buffers
__global__ __launch_bounds__(256) void seg_scatter_kernel(u64 *__restrict__ kx, u32 *__restrict__ vx,
                                                          u64 *__restrict__ ky, u32 *__restrict__ vy,
                                                          const u32 *__restrict__ lstart, const u32 *__restrict__ lsize,
                                                          const u32 *__restrict__ ltbase, const u32 *__restrict__ tile_seg,
                                                          const u32 *__restrict__ counters, int shift, u32 *__restrict__ cursor,
                                                          u64 *__restrict__ ybits) {
    __shared__ u64 s_k[SEG_PT];
    __shared__ u32 s_v[SEG_PT];
    __shared__ u32 s_cnt[256], s_lb[256], s_gb[256];
    __shared__ u32 s_scan[256 / 64 + 1];
    const u32 tile = blockIdx.x, tid = threadIdx.x;
    if (tile >= counters[1]) return;
    const u32 s = tile_seg[tile];
    const u32 sz = lsize[s];
    const bool from_y = (sz >> 31) != 0;
    const u64 *ksrc = from_y ? ky : kx;
    const u32 *vsrc = from_y ? vy : vx;
    u64 *kdst = from_y ? kx : ky;
    u32 *vdst = from_y ? vx : vy;
    const u32 size = sz & 0x7fffffffu;
    const u32 off = (tile - ltbase[s]) * SEG_PT;
    const u32 cnt = size - off < SEG_PT ? size - off : SEG_PT;
    const u64 base = (u64)lstart[s] + off;
    s_cnt[tid] = 0;
    __syncthreads();
    u64 k[SEG_PT / 256];
    u32 v[SEG_PT / 256], r[SEG_PT / 256];
#pragma unroll
    for (int q = 0; q < SEG_PT / 256; q++) {
        const u32 p = q * 256 + tid;
        k[q] = p < cnt ? ksrc[base + p] : 0ull;
        v[q] = p < cnt ? vsrc[base + p] : 0u;
    }
#pragma unroll
    for (int q = 0; q < SEG_PT / 256; q++) {
        const u32 p = q * 256 + tid;
        r[q] = p < cnt ? atomicAdd(&s_cnt[((u32)k[q] >> shift) & 255u], 1u) : 0u;
    }
    __syncthreads();
    const u32 c = s_cnt[tid];
    u32 tot;
    const u32 lb = block_excl_sum<256>(c, s_scan, &tot);
    s_lb[tid] = lb;
    s_gb[tid] = c ? atomicAdd(&cursor[(size_t)s * 256 + tid], c) : 0u;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SEG_PT / 256; q++) {
        const u32 p = q * 256 + tid;
        if (p < cnt) {
            const u32 o = s_lb[((u32)k[q] >> shift) & 255u] + r[q];
            s_k[o] = k[q];
            s_v[o] = v[q];
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SEG_PT / 256; q++) {
        const u32 p = q * 256 + tid;
        if (p < cnt) {
            const u64 x = s_k[p];
            const u32 d = ((u32)x >> shift) & 255u;
            const u32 g = s_gb[d] + (p - s_lb[d]);
            kdst[g] = x;
            vdst[g] = s_v[p];
        }
    }
    // the slots [base, base + cnt) now live in the other buffers
    const u64 e = base + cnt;
    for (u64 w = (base >> 6) + tid; w <= ((e - 1) >> 6); w += 256) {
        const u64 lo = w << 6;
        u64 mask = ~0ull;
        if (lo < base) mask &= ~0ull << (base - lo);
        if (lo + 64 > e) mask &= ~0ull >> (lo + 64 - e);
        atomicXor((unsigned long long *)&ybits[w], mask);
    }
}

// ---- runs up to SEG_CAP members: sorted in LDS ------------------------------------------------------------
#define SEG_PADX(i) ((i) + ((i) >> 4))      // a thread's 16 consecutive keys start 17 * 8 bytes apart
// compare-exchange steps of a bitonic network on the 16 keys a thread holds in layout LO: key q of thread t is element
// ((t >> LO) << (LO + 4)) | (q << LO) | (t & ((1 << LO) - 1)); bits [LO, LO + 4) of the element index are the thread's own.
// Steps for element-index bits bhi .. blo (both inside the layout's four), merge size 2^lk.
template <int LO>
__device__ __forceinline__ void seg_steps(u64 (&e)[16], const u32 tid, const int lk, const int bhi, const int blo) {
    const u32 ibase = ((tid >> LO) << (LO + 4)) | (tid & ((1u << LO) - 1u));
#pragma unroll
    for (int b = 3; b >= 0; b--) {
        if (b + LO > bhi || b + LO < blo) continue;
        const int jq = 1 << b;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if (q & jq) continue;
            const u32 i = ibase | ((u32)q << LO);
            const bool asc = lk >= 12 || ((i >> lk) & 1u) == 0u;
            const u64 a = e[q], c = e[q | jq];
            const bool sw = (a > c) == asc;
            e[q] = sw ? c : a;
            e[q | jq] = sw ? a : c;
        }
    }
}
template <int LO>
__device__ __forceinline__ void seg_lds_load(const u64 *s_key, u64 (&e)[16], const u32 tid) {
    const u32 ibase = ((tid >> LO) << (LO + 4)) | (tid & ((1u << LO) - 1u));
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const u32 i = ibase | ((u32)q << LO);
        e[q] = s_key[SEG_PADX(i)];
    }
}
template <int LO>
__device__ __forceinline__ void seg_lds_store(u64 *s_key, const u64 (&e)[16], const u32 tid) {
    const u32 ibase = ((tid >> LO) << (LO + 4)) | (tid & ((1u << LO) - 1u));
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const u32 i = ibase | ((u32)q << LO);
        s_key[SEG_PADX(i)] = e[q];
    }
}

// Window w is responsible for the slots [w * SPAN, (w + 1) * SPAN): it sorts every run of at most CAP members whose head
// lies there (the last of them may reach CAP - 1 slots further: the window's image holds W = SPAN + CAP slots), and it
// brings the slots of longer runs (equal ranks: nothing to sort) home from the alternate buffers.  Output in place into the
// primary buffers: no other window reads what this one writes (a neighbour may LOAD such a slot as part of its image, but
// never uses it).
__global__ __launch_bounds__(SEG_NT) void seg_small_kernel(u64 *__restrict__ kx, u32 *__restrict__ vx,
                                                           const u64 *__restrict__ ky, const u32 *__restrict__ vy, u32 m,
                                                           const u64 *__restrict__ segbits, const u64 *__restrict__ ybits) {
    __shared__ u64 s_key[SEG_W + SEG_W / 16];
    __shared__ u32 s_val[SEG_W];
    __shared__ u64 s_bits[SEG_EXTW], s_yb[SEG_W / 64];
    __shared__ i32 s_last[SEG_EXTW + 1], s_first[SEG_EXTW + 1];   // last head before word j / first head in words >= j (ext. slots)
    __shared__ u32 s_pop[SEG_EXTW + 1];                           // heads of the image's words before word j
    __shared__ u16 s_wm[SEG_NT];
    __shared__ u32 s_any;
    const u32 tid = threadIdx.x;
    const u64 t0 = (u64)blockIdx.x * SEG_SPAN;
    constexpr int LBW = SEG_CAP / 64;          // look-back words
    constexpr i32 BIG = 1 << 30;
    // bit words of the extended window [t0 - CAP, t0 + W]; slot m counts as a head
    if (tid < SEG_EXTW) {
        const i64 gw = (i64)(t0 >> 6) - LBW + tid;
        u64 b = gw >= 0 ? segbits[gw] : 0ull;
        if (gw >= 0 && (u64)gw == ((u64)m >> 6)) b |= 1ull << (m & 63);
        s_bits[tid] = b;
    }
    if (tid >= 128 && tid < 128 + SEG_W / 64) s_yb[tid - 128] = ybits[(t0 >> 6) + (tid - 128)];
    if (tid == 0) s_any = 0;
    __syncthreads();
    if (tid <= SEG_EXTW) {
        i32 last = -BIG, first = BIG;
        u32 pop = 0;
        for (int j = 0; j < SEG_EXTW; j++) {
            const u64 b = s_bits[j];
            if (j < (int)tid) {
                if (b) last = j * 64 + 63 - __builtin_clzll(b);
                if (j >= LBW) pop += (u32)__popcll(b);
            } else if (b && first == BIG) {
                first = j * 64 + __builtin_ctzll(b);
            }
        }
        s_last[tid] = last;
        s_first[tid] = first;
        s_pop[tid] = pop;
    }
    __syncthreads();
    // attributes of this thread's 16 consecutive slots
    const u32 p0 = tid * 16;
    const int wi = LBW + (int)(tid >> 2), sub = (int)(tid & 3) * 16;
    const u64 word = s_bits[wi];
    const u32 yw = (u32)(s_yb[tid >> 2] >> sub) & 0xffffu;
    u32 mine16 = 0, pass16 = 0, long16 = 0;
    u32 ordv[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const int b = sub + q;
        const u64 le = word & ((2ull << b) - 1ull);
        const u64 gt = b == 63 ? 0ull : word & ~((2ull << b) - 1ull);
        const i32 hs = le ? wi * 64 + 63 - __builtin_clzll(le) : s_last[wi];
        const i32 he = gt ? wi * 64 + __builtin_ctzll(gt) : s_first[wi + 1];
        const bool valid = t0 + p0 + q < m;
        const bool small = hs >= 0 && he < BIG && he - hs <= SEG_CAP;
        const bool mine = valid && small && hs >= SEG_CAP && hs < SEG_CAP + SEG_SPAN;
        const bool pass = valid && !small && p0 + q < SEG_SPAN && ((yw >> q) & 1u);
        mine16 |= (mine ? 1u : 0u) << q;
        pass16 |= (pass ? 1u : 0u) << q;
        long16 |= (valid && !small ? 1u : 0u) << q;
        ordv[q] = s_pop[wi] + (u32)__popcll(le);
    }
    if (mine16) s_any = 1;
    __syncthreads();
    if (!s_any) {
        // nothing to sort: only slots of long runs to bring home
        s_wm[tid] = (u16)pass16;
        __syncthreads();
        for (u32 p = tid; p < SEG_SPAN; p += SEG_NT) {
            if ((s_wm[p >> 4] >> (p & 15)) & 1u) {
                kx[t0 + p] = ky[t0 + p];
                vx[t0 + p] = vy[t0 + p];
            }
        }
        return;
    }
    // image: keys and values, coalesced, each slot from the buffer that holds it
#pragma unroll
    for (int q = 0; q < SEG_ITEMS; q++) {
        const u32 p = q * SEG_NT + tid;
        const u64 k = t0 + p;
        u64 key = 0;
        u32 val = 0;
        if (k < m) {
            const bool iny = (s_yb[p >> 6] >> (p & 63)) & 1ull;
            key = iny ? ky[k] : kx[k];
            val = iny ? vy[k] : vx[k];
        }
        s_key[SEG_PADX(p)] = key;
        s_val[p] = val;
    }
    __syncthreads();
    u64 e[16];
    u32 hi[16];
    seg_lds_load<0>(s_key, e, tid);
#pragma unroll
    for (int q = 0; q < 16; q++) {
        hi[q] = (u32)(e[q] >> 32);
        const u32 p = p0 + q;
        const bool valid = t0 + p < m;
        // (a long run holds equal ranks by now, so its slots keep their places and their keys; a slot of the neighbour
        // window's run is neither read nor written here: it is pinned by its slot number)
        const u32 rk = (((mine16 | long16) >> q) & 1u) ? (u32)e[q] : p;
        e[q] = valid ? ((u64)ordv[q] << 44) | ((u64)rk << 12) | p : ~0ull;
    }
    // the network: merge sizes 2 .. 16 inside the thread, then with the three layouts
    for (int lk = 1; lk <= 4; lk++) seg_steps<0>(e, tid, lk, lk - 1, 0);
    for (int lk = 5; lk <= 12; lk++) {
        __syncthreads();
        seg_lds_store<0>(s_key, e, tid);
        __syncthreads();
        if (lk > 8) {
            seg_lds_load<8>(s_key, e, tid);
            seg_steps<8>(e, tid, lk, lk - 1, 8);
            __syncthreads();
            seg_lds_store<8>(s_key, e, tid);
            __syncthreads();
        }
        seg_lds_load<4>(s_key, e, tid);
        seg_steps<4>(e, tid, lk, lk - 1 < 7 ? lk - 1 : 7, 4);
        __syncthreads();
        seg_lds_store<4>(s_key, e, tid);
        __syncthreads();
        seg_lds_load<0>(s_key, e, tid);
        seg_steps<0>(e, tid, lk, 3, 0);
    }
    // slot p0 + q now holds the member e[q] & 4095 came from
    u32 ov[16];
#pragma unroll
    for (int q = 0; q < 16; q++) ov[q] = s_val[(u32)e[q] & 4095u];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 16; q++) {
        const u32 p = p0 + q;
        s_key[SEG_PADX(p)] = ((u64)hi[q] << 32) | ((e[q] >> 12) & 0xffffffffull);
        s_val[p] = ov[q];
    }
    s_wm[tid] = (u16)(mine16 | pass16);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < SEG_ITEMS; q++) {
        const u32 p = q * SEG_NT + tid;
        if ((s_wm[p >> 4] >> (p & 15)) & 1u) {
            kx[t0 + p] = s_key[SEG_PADX(p)];
            vx[t0 + p] = s_val[p];
        }
    }
}

__global__ __launch_bounds__(256) void seg_iota_kernel(u32 *__restrict__ v, u32 m) {
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k < m) v[k] = (u32)k;
}

#endif  // __HIPCC__
