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
    u32 *lshift[2];    // the run's digit = (rank >> shift) & 255 at this level
    u32 *tile_seg;     // tile -> run
    u32 *hist;         // [runs][256] digit counts, then cursors
    u32 *mm;           // [runs][2] smallest / largest rank of a run (equal: nothing to sort, the run is left alone)
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
                                                       u32 *__restrict__ ltbase, u32 *__restrict__ lshift, u32 shift0,
                                                       u32 *__restrict__ counters, u32 cap_runs) {
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
                lshift[s] = shift0;
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
                                                        const u32 *__restrict__ ltbase, const u32 *__restrict__ lshift,
                                                        const u32 *__restrict__ tile_seg, const u32 *__restrict__ counters,
                                                        u32 *__restrict__ hist, u32 *__restrict__ mm) {
    __shared__ u32 s_h[256];
    __shared__ u32 s_mm[2];
    const u32 tile = blockIdx.x;
    if (tile >= counters[1]) return;
    const u32 s = tile_seg[tile];
    const u32 sz = lsize[s];
    const u32 shift = lshift[s];
    const u64 *src = (sz >> 31) ? ky : kx;
    const u32 size = sz & 0x7fffffffu;
    const u32 off = (tile - ltbase[s]) * SEG_PT;
    const u32 cnt = size - off < SEG_PT ? size - off : SEG_PT;
    const u64 base = (u64)lstart[s] + off;
    s_h[threadIdx.x] = 0;
    if (threadIdx.x == 0) { s_mm[0] = 0xffffffffu; s_mm[1] = 0u; }
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
    {
        u32 lo = 0xffffffffu, hi = 0u;
#pragma unroll
        for (int q = 0; q < SEG_PT / 256; q++) {
            const u32 p = q * 256 + threadIdx.x;
            if (p < cnt) {
                lo = lo < (u32)v[q] ? lo : (u32)v[q];
                hi = hi > (u32)v[q] ? hi : (u32)v[q];
            }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const u32 a = (u32)__shfl_xor((int)lo, d, 64), b = (u32)__shfl_xor((int)hi, d, 64);
            lo = lo < a ? lo : a;
            hi = hi > b ? hi : b;
        }
        if ((threadIdx.x & 63) == 0) {
            atomicMin(&s_mm[0], lo);
            atomicMax(&s_mm[1], hi);
        }
    }
    __syncthreads();
    const u32 c = s_h[threadIdx.x];
    if (c) atomicAdd(&hist[(size_t)s * 256 + threadIdx.x], c);
    if (threadIdx.x == 0) {
        atomicMin(&mm[2 * (size_t)s], s_mm[0]);
        atomicMax(&mm[2 * (size_t)s + 1], s_mm[1]);
    }
}

// per run (one wave): counts -> cursors; the children's heads into segbits; children still longer than the cap onto the
// next level's list (not after the last level: its children hold equal ranks)
__global__ __launch_bounds__(256) void seg_scan_kernel(const u32 *__restrict__ lstart, const u32 *__restrict__ lsize,
                                                       const u32 *__restrict__ lshift, const u32 *__restrict__ counters,
                                                       u32 *__restrict__ hist, u32 *__restrict__ mm, u64 *__restrict__ segbits,
                                                       int last_level, u32 *__restrict__ nstart, u32 *__restrict__ nsize,
                                                       u32 *__restrict__ ntbase, u32 *__restrict__ nshift,
                                                       u32 *__restrict__ ncounters, u32 cap_runs) {
    const u32 nruns = counters[0] < cap_runs ? counters[0] : cap_runs;
    const u32 nwaves = gridDim.x * 4, l = threadIdx.x & 63;
    for (u32 s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nruns; s += nwaves) {
        const u32 rmin = mm[2 * (size_t)s], rmax = mm[2 * (size_t)s + 1];
        if (rmin == rmax) continue;   // equal ranks: the run stays as it is, where it is
        u32 *h = hist + (size_t)s * 256;
        const uint4 c = *reinterpret_cast<const uint4 *>(h + 4 * l);
        const u32 mine = c.x + c.y + c.z + c.w;
        const u32 excl = wave_incl_sum(mine) - mine;
        const u32 start = lstart[s];
        const u32 shift = lshift[s];
        // ALL members in one digit (the ranks of a repeat's members lie next to each other: the top digits separate
        // nothing): no copy -- the run goes onto the next level's list as it is, with the shift that puts its highest
        // differing rank bit on top of the digit (that level then separates its smallest from its largest rank)
        {
            const u64 nz = __ballot(mine != 0);
            const bool one = __popcll(nz) == 1 && (c.x == mine || c.y == mine || c.z == mine || c.w == mine);
            if (__ballot(mine != 0 && one) == nz && __popcll(nz) == 1) {   // (wave-uniform)
                if (l == 0 && !last_level) {
                    const u32 hb = 31u - (u32)__builtin_clz(rmin ^ rmax);
                    const u32 sz = lsize[s];
                    const u32 t = atomicAdd(&ncounters[0], 1u);
                    const u32 tb = atomicAdd(&ncounters[1], ((sz & 0x7fffffffu) + SEG_PT - 1) / SEG_PT);
                    if (t < cap_runs) {
                        nstart[t] = start;
                        nsize[t] = sz;
                        ntbase[t] = tb;
                        nshift[t] = hb >= 7u ? hb - 7u : 0u;
                    }
                    mm[2 * (size_t)s] = 0u;          // (seg_scatter_kernel leaves a run with equal marks alone)
                    mm[2 * (size_t)s + 1] = 0u;
                }
                continue;
            }
        }
        const u32 other = (~lsize[s]) & 0x80000000u;   // the children sit in the buffers this level writes
        const u32 cc[4] = {c.x, c.y, c.z, c.w};
        u32 run = start + excl;
        u32 cur[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            cur[q] = run;
            if (cc[q]) {
                atomicOr((unsigned long long *)&segbits[run >> 6], 1ull << (run & 63));
                if (cc[q] > SEG_CAP && !last_level && shift > 0u) {   // (shift 0: the children hold equal ranks)
                    const u32 t = atomicAdd(&ncounters[0], 1u);
                    const u32 tb = atomicAdd(&ncounters[1], (cc[q] + SEG_PT - 1) / SEG_PT);
                    if (t < cap_runs) {
                        nstart[t] = run;
                        nsize[t] = cc[q] | other;
                        ntbase[t] = tb;
                        nshift[t] = shift >= 8u ? shift - 8u : 0u;
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
                                                          const u32 *__restrict__ ltbase, const u32 *__restrict__ lshift,
                                                          const u32 *__restrict__ tile_seg, const u32 *__restrict__ counters,
                                                          u32 *__restrict__ cursor, const u32 *__restrict__ mm, u64 *__restrict__ ybits) {
    __shared__ u64 s_k[SEG_PT];
    __shared__ u32 s_v[SEG_PT];
    __shared__ u32 s_cnt[256], s_lb[256], s_gb[256];
    __shared__ u32 s_scan[256 / 64 + 1];
    const u32 tile = blockIdx.x, tid = threadIdx.x;
    if (tile >= counters[1]) return;
    const u32 s = tile_seg[tile];
    if (mm[2 * (size_t)s] == mm[2 * (size_t)s + 1]) return;   // (equal ranks, or all in one digit: see seg_scan_kernel)
    const u32 sz = lsize[s];
    const u32 shift = lshift[s];
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
// Three kinds of runs in a window: TINY (2 .. SEG_TINY members: every member counts the members of its run that
// precede it -- a few LDS reads), MID (up to SEG_CAP: the mid members of the window are compacted into one array
// and sorted by a bitonic network over (run head, rank, source slot) -- the head keeps every run in its own slots, so
// the network needs no knowledge of the run boundaries; its size is the power of two that holds the mid members, so a
// window pays for what it has), and the rest (singletons, long runs: equal ranks by now, slots of a neighbour's run),
// which keep their places.
#define SEG_TINY 15
#define SEG_PADX(i) ((i) + ((i) >> 4))      // a thread's 16 consecutive keys start 17 * 8 bytes apart
// compare-exchange steps of the network on the 16 keys a thread holds in layout LO: key q of thread t is element
// ((t >> LO) << (LO + 4)) | (q << LO) | (t & ((1 << LO) - 1)).  Steps for the element-index bits BHI .. BLO (inside the
// layout's four), merge size 2^LK (compile time: the direction of a pair is a constant or one flag per thread).
template <int LO, int LK, int BHI, int BLO>
__device__ __forceinline__ void seg_steps(u64 (&e)[16], const u32 tid) {
    const u32 ibase = ((tid >> LO) << (LO + 4)) | (tid & ((1u << LO) - 1u));
    const bool tdesc = (LK >= LO + 4 || LK < LO) ? ((ibase >> LK) & 1u) != 0u : false;   // (bit LK of the index is a thread bit)
#pragma unroll
    for (int b = 3; b >= 0; b--) {
        if (b + LO > BHI || b + LO < BLO) continue;
        const int jq = 1 << b;
#pragma unroll
        for (int q = 0; q < 16; q++) {
            if (q & jq) continue;
            const bool qdesc = (LK >= LO && LK < LO + 4) ? ((q >> (LK - LO)) & 1) != 0 : false;   // (.. or one of the key's own)
            const u64 a = e[q], c = e[q | jq];
            const bool sw = (a > c) != (tdesc || qdesc);
            e[q] = sw ? c : a;
            e[q | jq] = sw ? a : c;
        }
    }
}
// (Q: the keys of a thread that exist in this layout when the network covers fewer than 4096 elements)
template <int LO>
__device__ __forceinline__ void seg_lds_load(const u64 *s_key, u64 (&e)[16], const u32 tid, const u32 Q = 16) {
    const u32 ibase = ((tid >> LO) << (LO + 4)) | (tid & ((1u << LO) - 1u));
#pragma unroll
    for (int q = 0; q < 16; q++)
        if ((u32)q < Q) e[q] = s_key[SEG_PADX(ibase + ((u32)q << LO))];
}
template <int LO>
__device__ __forceinline__ void seg_lds_store(u64 *s_key, const u64 (&e)[16], const u32 tid, const u32 Q = 16) {
    const u32 ibase = ((tid >> LO) << (LO + 4)) | (tid & ((1u << LO) - 1u));
#pragma unroll
    for (int q = 0; q < 16; q++)
        if ((u32)q < Q) s_key[SEG_PADX(ibase + ((u32)q << LO))] = e[q];
}
// one merge size of the network over the first 2^lw elements of s_key (lw >= 4, lw >= LK).  Which threads hold keys, and
// how many, depends on the layout: layout 0: threads below 2^(lw - 4), 16 keys each; layout 4: threads whose upper four
// bits are below 2^(lw - 8), 16 keys (lw < 8: the first 16 threads, 2^(lw - 4) keys); layout 8: every thread, 2^(lw - 8) keys.
template <int LK>
__device__ __forceinline__ void seg_merge(u64 *s_key, u64 (&e)[16], const u32 tid, const int lw) {
    const bool on0 = tid < (1u << (lw - 4));
    if constexpr (LK <= 4) {
        if (on0) seg_steps<0, LK, LK - 1, 0>(e, tid);
    } else {
        if (on0) seg_lds_store<0>(s_key, e, tid);
        __syncthreads();
        if constexpr (LK > 8) {
            const u32 Q8 = 1u << (lw - 8);
            seg_lds_load<8>(s_key, e, tid, Q8);
            seg_steps<8, LK, LK - 1, 8>(e, tid);
            __syncthreads();
            seg_lds_store<8>(s_key, e, tid, Q8);
            __syncthreads();
        }
        const bool on4 = lw >= 8 ? (tid >> 4) < (1u << (lw - 8)) : tid < 16u;
        const u32 Q4 = lw >= 8 ? 16u : 1u << (lw - 4);
        if (on4) {
            seg_lds_load<4>(s_key, e, tid, Q4);
            seg_steps<4, LK, (LK - 1 < 7 ? LK - 1 : 7), 4>(e, tid);
        }
        __syncthreads();
        if (on4) seg_lds_store<4>(s_key, e, tid, Q4);
        __syncthreads();
        if (on0) {
            seg_lds_load<0>(s_key, e, tid);
            seg_steps<0, LK, 3, 0>(e, tid);
        }
    }
}

#ifdef SEG_PROFILE
__device__ u64 seg_prof[16];   // cycles per phase of seg_small_kernel (thread 0 of every window), windows, mid members
#endif
// Window w is responsible for the slots [w * SPAN, (w + 1) * SPAN): it sorts every run of at most CAP members whose head
// lies there (the last of them may reach CAP - 1 slots further: the window's image holds W = SPAN + CAP slots), and it
// brings the slots of longer runs (equal ranks: nothing to sort) home from the alternate buffers.  Output in place into the
// primary buffers: no other window reads what this one writes (a neighbour may LOAD such a slot as part of its image, but
// never uses it).
__global__ __launch_bounds__(SEG_NT, 2) void seg_small_kernel(u64 *__restrict__ kx, u32 *__restrict__ vx,
                                                           const u64 *__restrict__ ky, const u32 *__restrict__ vy, u32 m,
                                                           const u64 *__restrict__ segbits, const u64 *__restrict__ ybits) {
    __shared__ u64 s_cmp[SEG_W + SEG_W / 16];    // the network's keys; at the end the staged output keys
    __shared__ u32 s_rank[SEG_W], s_hi[SEG_W / 64], s_val[SEG_W];
    __shared__ __attribute__((aligned(16))) u16 s_att[SEG_W];                 // [11:0] head of the slot's run (window slot), [15:12] 0 nothing to do, 1 .. 14: tiny run of 2 .. 15, 15: mid
    __shared__ u64 s_bits[SEG_EXTW], s_yb[SEG_W / 64], s_midw[SEG_W / 64], s_tinyw[SEG_W / 64];
    __shared__ u32 s_merge;
    __shared__ i32 s_last[SEG_EXTW + 1], s_first[SEG_EXTW + 1];   // last head before word j / first head in words >= j (ext. slots)
    __shared__ u32 s_wpre[SEG_W / 64 + 1];
    __shared__ u32 s_lastp[128], s_firstp[128], s_scan2w[4];
    __shared__ u16 s_wm[SEG_NT];
    __shared__ u32 s_any, s_tmax;
    (void)s_hi;
    const u32 tid = threadIdx.x;
    const u64 t0 = (u64)blockIdx.x * SEG_SPAN;
    constexpr int LBW = SEG_CAP / 64;          // look-back words
    constexpr i32 BIG = 1 << 30;
#ifdef SEG_PROFILE
    u64 sp_t[10];
    if (tid == 0) sp_t[0] = __builtin_readcyclecounter();
#endif
    // bit words of the extended window [t0 - CAP, t0 + W]; slot m counts as a head
    if (tid < SEG_EXTW) {
        const i64 gw = (i64)(t0 >> 6) - LBW + tid;
        u64 b = gw >= 0 ? segbits[gw] : 0ull;
        if (gw >= 0 && (u64)gw == ((u64)m >> 6)) b |= 1ull << (m & 63);
        s_bits[tid] = b;
    }
    if (tid >= 128 && tid < 128 + SEG_W / 64) s_yb[tid - 128] = ybits[(t0 >> 6) + (tid - 128)];
    if (tid == 0) { s_any = 0; s_tmax = 0; }
    __syncthreads();
    // s_last[j]: last head in the words before j; s_first[j]: first head in the words from j on (two waves, shuffles:
    // a loop over the 81 words per thread was 20 000 cycles of a window's 138 000 -- SEG_PROFILE)
    if (tid < 128) {
        const u32 l = tid & 63, w = tid >> 6;
        const u64 bf = tid < SEG_EXTW ? s_bits[tid] : 0ull;                       // forward: word tid
        const u32 rj = 127u - tid;                                                // backward: word 127 - tid
        const u64 bb = rj < SEG_EXTW ? s_bits[rj] : 0ull;
        u32 lp = bf ? (u32)(tid * 64 + 63 - __builtin_clzll(bf)) + 1u : 0u;       // (1 + position; 0: none)
        u32 fp = bb ? (u32)(rj * 64 + __builtin_ctzll(bb)) : 0x7fffffffu;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const u32 a = (u32)__shfl_up((int)lp, d, 64), c = (u32)__shfl_up((int)fp, d, 64);
            if ((int)l >= d) { lp = lp > a ? lp : a; fp = fp < c ? fp : c; }
        }
        if (l == 63) { s_scan2w[w] = lp; s_scan2w[2 + w] = fp; }
        s_lastp[tid] = lp;      // inclusive over the wave's words up to tid
        s_firstp[tid] = fp;     // inclusive over the wave's (reversed) words
    }
    __syncthreads();
    if (tid <= SEG_EXTW) {
        // exclusive prefix maximum over words < tid
        u32 lp = 0;
        if (tid > 0) {
            const u32 j = tid - 1;
            lp = s_lastp[j];
            if (j >= 64) lp = lp > s_scan2w[0] ? lp : s_scan2w[0];
        }
        s_last[tid] = lp ? (i32)(lp - 1u) : -BIG;
        // inclusive suffix minimum over words >= tid: reversed index 127 - tid
        u32 fp = 0x7fffffffu;
        if (tid < SEG_EXTW) {
            const u32 t = 127u - tid;
            fp = s_firstp[t];
            if (t >= 64) fp = fp < s_scan2w[2] ? fp : s_scan2w[2];
        }
        s_first[tid] = fp == 0x7fffffffu ? BIG : (i32)fp;
    }
    __syncthreads();
#ifdef SEG_PROFILE
    if (tid == 0) sp_t[1] = __builtin_readcyclecounter();
#endif
    // attributes of this thread's 16 consecutive slots
    const u32 p0 = tid * 16;
    u32 mine16 = 0, pass16 = 0, mid16 = 0, tiny16 = 0, tmax = 0;
    {
        const int wi = LBW + (int)(tid >> 2), sub = (int)(tid & 3) * 16;
        const u64 word = s_bits[wi];
        const u32 yw = (u32)(s_yb[tid >> 2] >> sub) & 0xffffu;
        // the run of each of the 16 slots: its head = the last head at or before the slot, its end = the first head behind
        // it -- two 64-bit looks outside the thread's 16 bits, then two walks over them in 32-bit arithmetic (sixteen
        // clz / ctz of 64-bit masks per thread were 15 000 cycles of a window's 138 000)
        const u32 bits16 = (u32)(word >> sub) & 0xffffu;
        const u64 below = sub ? word & ((1ull << sub) - 1ull) : 0ull;
        const u64 above = sub + 16 < 64 ? word >> (sub + 16) : 0ull;
        const i32 base = wi * 64 + sub;
        i32 hsv[16], hev[16];
        {
            i32 cur = below ? wi * 64 + 63 - __builtin_clzll(below) : s_last[wi];
#pragma unroll
            for (int q = 0; q < 16; q++) {
                if ((bits16 >> q) & 1u) cur = base + q;
                hsv[q] = cur;
            }
            cur = above ? base + 16 + __builtin_ctzll(above) : s_first[wi + 1];
#pragma unroll
            for (int q = 15; q >= 0; q--) {
                hev[q] = cur;
                if ((bits16 >> q) & 1u) cur = base + q;
            }
        }
        u32 attv[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const i32 hs = hsv[q], he = hev[q];
            const bool valid = t0 + p0 + q < m;
            const bool small = hs >= 0 && he < BIG && he - hs <= SEG_CAP;
            const bool mine = valid && small && hs >= SEG_CAP && hs < SEG_CAP + SEG_SPAN;
            const bool pass = valid && !small && p0 + q < SEG_SPAN && ((yw >> q) & 1u);
            const u32 size = mine ? (u32)(he - hs) : 0u;
            const u32 code = size < 2 ? 0u : (size <= SEG_TINY ? size - 1u : 15u);
            attv[q >> 1] |= (((mine ? (u32)(hs - SEG_CAP) : 0u) | (code << 12)) & 0xffffu) << (16 * (q & 1));
            mine16 |= (mine ? 1u : 0u) << q;
            pass16 |= (pass ? 1u : 0u) << q;
            mid16 |= (code == 15u ? 1u : 0u) << q;
            tiny16 |= (code >= 1u && code < 15u ? 1u : 0u) << q;
            if (code >= 1u && code < 15u && size > tmax) tmax = size;
        }
        // (the 16 attributes as two 16-byte stores: sixteen 2-byte stores 32 bytes apart between lanes collide on the banks)
        uint4 *ap = reinterpret_cast<uint4 *>(s_att + p0);
        ap[0] = make_uint4(attv[0], attv[1], attv[2], attv[3]);
        ap[1] = make_uint4(attv[4], attv[5], attv[6], attv[7]);
    }
    s_wm[tid] = (u16)(mine16 | pass16);
    if (mine16) s_any = 1;
    if (tmax) atomicMax(&s_tmax, tmax);
    // mid members per 64-slot word (four threads a word) and their prefix: the compact index of a mid slot
    {
        u64 w4 = (u64)mid16 << ((tid & 3) * 16);
        w4 |= __shfl_xor(w4, 1, 64);
        w4 |= __shfl_xor(w4, 2, 64);
        if ((tid & 3) == 0) s_midw[tid >> 2] = w4;
        u64 t4 = (u64)tiny16 << ((tid & 3) * 16);
        t4 |= __shfl_xor(t4, 1, 64);
        t4 |= __shfl_xor(t4, 2, 64);
        if ((tid & 3) == 0) s_tinyw[tid >> 2] = t4;
    }
    __syncthreads();
    if (!s_any) {
        // nothing to sort: only slots of long runs to bring home
        for (u32 p = tid; p < SEG_SPAN; p += SEG_NT) {
            if ((s_wm[p >> 4] >> (p & 15)) & 1u) {
                kx[t0 + p] = ky[t0 + p];
                vx[t0 + p] = vy[t0 + p];
            }
        }
        return;
    }
    if (tid < 64) {
        // The network's size is the power of two that holds the mid members.  When the window's tiny members fit into the
        // same power of two they go through the network as well -- it costs the same with them -- instead of each
        // counting its run (a walk as long as the wave's longest tiny run, for all 16 slots of a thread).
        u64 mw = s_midw[tid];
        const u64 tw = s_tinyw[tid];
        const u32 nm = wave_sum((u32)__popcll(mw)), nt = wave_sum((u32)__popcll(tw));
        int la = 4, lb = 4;
        while ((1u << la) < nm) la++;
        while ((1u << lb) < nm + nt) lb++;
        const bool merge = nm > 0 && nt > 0 && la == lb;
        if (merge) {
            mw |= tw;
            s_midw[tid] = mw;
        }
        if (tid == 0) s_merge = merge ? 1u : 0u;
        const u32 c = (u32)__popcll(mw);
        const u32 inc = wave_incl_sum(c);
        s_wpre[tid] = inc - c;
        if (tid == 63) s_wpre[64] = inc;
    }
#ifdef SEG_PROFILE
    if (tid == 0) sp_t[2] = __builtin_readcyclecounter();
#endif
    // image: keys and values, coalesced, each slot from the buffer that holds it (the top key halves stay in registers)
    u32 hi[SEG_ITEMS];
    // (a thread's loads four slots at a time -- eight loads in flight -- then their stores into the image: as one loop it
    // was a load, a wait, a store, sixteen times: 20 000 cycles of the window by SEG_PROFILE; all sixteen slots at once
    // cost 64 address registers and spilled)
#pragma unroll
    for (int q0 = 0; q0 < SEG_ITEMS; q0 += 4) {
        u64 ikey[4];
        u32 ival[4];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const u32 p = (q0 + x) * SEG_NT + tid;
            const u64 k = t0 + p;
            ikey[x] = 0;
            ival[x] = 0;
            if (k < m) {
                const bool iny = (s_yb[p >> 6] >> (p & 63)) & 1ull;
                ikey[x] = iny ? ky[k] : kx[k];
                ival[x] = iny ? vy[k] : vx[k];
            }
        }
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const u32 p = (q0 + x) * SEG_NT + tid;
            hi[q0 + x] = (u32)(ikey[x] >> 32);
            s_rank[p] = (u32)ikey[x];
            s_val[p] = ival[x];
        }
    }
    __syncthreads();
#ifdef SEG_PROFILE
    if (tid == 0) sp_t[3] = __builtin_readcyclecounter();
#endif
    const u32 nmid = s_wpre[64];
    int lw = 4;                                   // the network covers 2^lw >= nmid elements
    while ((1u << lw) < nmid) lw++;
    // tiny runs: a member's place = head + the members of its run that sort before it; mid members: into the compact array
    u32 tdest[SEG_ITEMS], trank[SEG_ITEMS], tval[SEG_ITEMS];
    const bool merge_tiny = s_merge != 0u;
    u32 wtiny = 0;   // the longest tiny run among the slots of this wave (0: none, or they go through the network)
    if (!merge_tiny && s_tmax) {
#pragma unroll
        for (int q = 0; q < SEG_ITEMS; q++) {
            const u32 c = (u32)s_att[q * SEG_NT + tid] >> 12;
            const u32 sz = (c >= 1u && c < 15u) ? c + 1u : 0u;
            wtiny = wtiny > sz ? wtiny : sz;
        }
#pragma unroll
        for (int dd = 32; dd >= 1; dd >>= 1) {
            const u32 o = (u32)__shfl_xor((int)wtiny, dd, 64);
            wtiny = wtiny > o ? wtiny : o;
        }
    }
#pragma unroll
    for (int q = 0; q < SEG_ITEMS; q++) {
        const u32 p = q * SEG_NT + tid;
        const u32 att = s_att[p];
        const u32 hsw = att & 4095u;
        const u32 code = (merge_tiny && (att >> 12) != 0u) ? 15u : att >> 12;
        const u32 rk = s_rank[p];
        tdest[q] = 0xffffffffu;
        trank[q] = rk;
        tval[q] = s_val[p];
        if (code == 15u) {
            const u32 c = s_wpre[p >> 6] + (u32)__popcll(s_midw[p >> 6] & ((1ull << (p & 63)) - 1ull));
            s_cmp[SEG_PADX(c)] = ((u64)hsw << 44) | ((u64)rk << 12) | p;
        }
        if (wtiny) {   // (wave-uniform: the longest tiny run among ALL slots this wave handles, one reduction per window)
            const u32 size = (code >= 1u && code < 15u) ? code + 1u : 0u;
            u32 cnt = 0;
#pragma unroll 5
            for (u32 d = 0; d < wtiny; d++) {   // (unrolled: five image reads in flight instead of one per turn)
                const u32 j = hsw + d;
                const u32 r = s_rank[j & (SEG_W - 1)];
                cnt += (d < size && (r < rk || (r == rk && j < p))) ? 1u : 0u;
            }
            if (size) tdest[q] = hsw + cnt;
        }
    }
    if (nmid) {
        for (u32 c = nmid + tid; c < (1u << lw); c += SEG_NT) s_cmp[SEG_PADX(c)] = ~0ull;
    }
    __syncthreads();
#ifdef SEG_PROFILE
    if (tid == 0) sp_t[4] = __builtin_readcyclecounter();
#endif
    // tiny members to their places (all reads of the image are done)
#pragma unroll
    for (int q = 0; q < SEG_ITEMS; q++) {
        if (tdest[q] != 0xffffffffu) {
            s_rank[tdest[q]] = trank[q];
            s_val[tdest[q]] = tval[q];
        }
    }
#ifdef SEG_PROFILE
    if (tid == 0) sp_t[5] = __builtin_readcyclecounter();
#endif
    if (nmid) {   // (block-uniform)
        const bool on = p0 < (1u << lw);
        u64 e[16];
        if (on) seg_lds_load<0>(s_cmp, e, tid);
        seg_merge<1>(s_cmp, e, tid, lw);
        seg_merge<2>(s_cmp, e, tid, lw);
        seg_merge<3>(s_cmp, e, tid, lw);
        seg_merge<4>(s_cmp, e, tid, lw);
        if (lw >= 5) seg_merge<5>(s_cmp, e, tid, lw);
        if (lw >= 6) seg_merge<6>(s_cmp, e, tid, lw);
        if (lw >= 7) seg_merge<7>(s_cmp, e, tid, lw);
        if (lw >= 8) seg_merge<8>(s_cmp, e, tid, lw);
        if (lw >= 9) seg_merge<9>(s_cmp, e, tid, lw);
        if (lw >= 10) seg_merge<10>(s_cmp, e, tid, lw);
        if (lw >= 11) seg_merge<11>(s_cmp, e, tid, lw);
        if (lw >= 12) seg_merge<12>(s_cmp, e, tid, lw);
        // compact slot c = 16 tid + q holds the member that came from window slot e[q] & 4095; its place: the head of its
        // run + its distance from the head's compact slot
        u32 mv[16];
#pragma unroll
        for (int q = 0; q < 16; q++) mv[q] = (on && p0 + q < nmid) ? s_val[(u32)e[q] & 4095u] : 0u;
        __syncthreads();
        if (on) {
#pragma unroll
            for (int q = 0; q < 16; q++) {
                const u32 c = p0 + q;
                if (c < nmid) {
                    const u32 hsw = (u32)(e[q] >> 44) & 4095u;
                    const u32 ch = s_wpre[hsw >> 6] + (u32)__popcll(s_midw[hsw >> 6] & ((1ull << (hsw & 63)) - 1ull));
                    const u32 dest = hsw + (c - ch);
                    s_rank[dest] = (u32)(e[q] >> 12);
                    s_val[dest] = mv[q];
                }
            }
        }
    }
    __syncthreads();
#ifdef SEG_PROFILE
    if (tid == 0) sp_t[6] = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int q = 0; q < SEG_ITEMS; q++) {
        const u32 p = q * SEG_NT + tid;
        if ((s_wm[p >> 4] >> (p & 15)) & 1u) {
            kx[t0 + p] = ((u64)hi[q] << 32) | s_rank[p];
            vx[t0 + p] = s_val[p];
        }
    }
#ifdef SEG_PROFILE
    if (tid == 0) {
        sp_t[7] = __builtin_readcyclecounter();
        for (int x = 0; x < 7; x++) atomicAdd((unsigned long long *)&seg_prof[x], (unsigned long long)(sp_t[x + 1] - sp_t[x]));
        atomicAdd((unsigned long long *)&seg_prof[7], 1ull);
        atomicAdd((unsigned long long *)&seg_prof[8], (unsigned long long)nmid);
    }
#endif
}

// ---- a tied set of at most 4096 members (an iid text: 2 404 of 2^30 suffixes) ---------------------------------
// One workgroup and the network above instead of two radix sorts of a few thousand pairs (each: a histogram, a scan, a
// memset, four passes, copies -- ~20 launches of ~8 us for 30 KB of data): MODE 0 brings the set (slot, idx, grp) into SA
// order in place (void entries, slot = ~0, go last); MODE 1 builds the sparse rank table from it -- t_idx sorted by text
// position, t_rank = the member's group, tpos[k] = its row -- (table_build_kernel in tc_sa.hpp).
template <int MODE>
__global__ __launch_bounds__(SEG_NT) void tied_small_kernel(u32 *__restrict__ slot, u32 *__restrict__ idx, u32 *__restrict__ grp,
                                                            u32 m, u32 *__restrict__ t_idx, u32 *__restrict__ t_rank,
                                                            u32 *__restrict__ tpos) {
    __shared__ u64 s_cmp[SEG_W + SEG_W / 16];
    __shared__ u32 s_a[SEG_W], s_b[SEG_W];
    const u32 tid = threadIdx.x, p0 = tid * 16;
    int lw = 4;
    while ((1u << lw) < m) lw++;
    for (u32 p = tid; p < (1u << lw); p += SEG_NT) {
        u64 c = ~0ull;
        if (p < m) {
            const u32 k = MODE == 0 ? slot[p] : idx[p];
            c = ((u64)k << 12) | p;
            s_a[p] = MODE == 0 ? idx[p] : grp[p];
            if (MODE == 0) s_b[p] = grp[p];
        }
        s_cmp[SEG_PADX(p)] = c;
    }
    __syncthreads();
    u64 e[16];
    if (p0 < (1u << lw)) seg_lds_load<0>(s_cmp, e, tid);
    seg_merge<1>(s_cmp, e, tid, lw);
    seg_merge<2>(s_cmp, e, tid, lw);
    seg_merge<3>(s_cmp, e, tid, lw);
    seg_merge<4>(s_cmp, e, tid, lw);
    if (lw >= 5) seg_merge<5>(s_cmp, e, tid, lw);
    if (lw >= 6) seg_merge<6>(s_cmp, e, tid, lw);
    if (lw >= 7) seg_merge<7>(s_cmp, e, tid, lw);
    if (lw >= 8) seg_merge<8>(s_cmp, e, tid, lw);
    if (lw >= 9) seg_merge<9>(s_cmp, e, tid, lw);
    if (lw >= 10) seg_merge<10>(s_cmp, e, tid, lw);
    if (lw >= 11) seg_merge<11>(s_cmp, e, tid, lw);
    if (lw >= 12) seg_merge<12>(s_cmp, e, tid, lw);
    if (p0 < (1u << lw)) {
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const u32 c = p0 + q;
            if (c < m) {
                const u32 src = (u32)e[q] & 4095u, k = (u32)(e[q] >> 12);
                if (MODE == 0) {
                    slot[c] = k;
                    idx[c] = s_a[src];
                    grp[c] = s_b[src];
                } else {
                    t_idx[c] = k;
                    t_rank[c] = s_a[src];
                    tpos[src] = c;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void seg_mm_init_kernel(u32 *__restrict__ mm, u32 nruns) {
    const u32 s = blockIdx.x * 256 + threadIdx.x;
    if (s < nruns) { mm[2 * (size_t)s] = 0xffffffffu; mm[2 * (size_t)s + 1] = 0u; }
}

__global__ __launch_bounds__(256) void seg_iota_kernel(u32 *__restrict__ v, u32 m) {
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k < m) v[k] = (u32)k;
}

#endif  // __HIPCC__
