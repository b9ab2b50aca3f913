// tc_chain.hpp -- the CHAIN ROUND of the prefix doubling (round 4): periodic text in O(1) rounds instead of log2 n.
//
// A doubling round of createSuffixArray's replacement (reference BWT/Internal.hs:110-134; tc_sa.hpp) orders the members
// of a tied group g by rank[i + h].  On a text with a long period p every residue class is one group, every member but
// the few nearest the end sees the SAME rank at i + h, the round sheds those few -- and log2(n / h) rounds each sort all
// N members for nothing (1 GiB: 27 rounds, 2.5 s).  But what decides the order of two members a, b of g is known without
// more rounds: follow both by steps of h while they keep seeing the same thing.  Give every tied group g a reference
// rank ref[g] (the rank one of its members sees at + h: ANY choice is correct, the majority's is the useful one).
// Position q (group g) is ON PATH iff rank[q + h] == ref[g]; otherwise it is a terminal with sign s = (rank[q + h] > ref[g]).
// For a member a let k(a) = the number of on-path steps from a (a, a + h, .. a + (k - 1) h on path, a + k h terminal)
// and s(a) the terminal's sign.  Two members of one group walk through the same groups while both are on path, so
//   k(a) < k(b):  at step k(a) b sees ref, a sees something smaller (s = 0: a < b) or larger (s = 1: a > b);
//   k(a) = k(b), s differ: the s = 0 one is smaller;   k, s equal: decided by the rank the terminals see (or tied on).
// So the order inside g is by the code  s = 0: k   |   s = 1: 0xffffffff - k   (pass 1), then by rank[a + (k + 1) h] (pass 2),
// and members equal on both share at least (k + 2) h >= 2 h symbols: the doubling goes on with 2 h as after a plain round.
// A plain round is the special case k = 0 everywhere; on periodic text k(a) ~ (n - a) / h is different for every member of
// a residue class and ONE such round resolves the text.  A position that is not tied (ref = none) is a terminal: at most
// one member of a group can reach it on path (its rank is its own), so its code needs no sign.
//
// (k, s) per position is a suffix scan along stride h: k(q) = on_path(q) ? k(q + h) + 1 : 0.  Here: flags as two bitmaps
// (chain_flags_kernel: streaming over the dense rank array + one gather of ref), then a three-phase scan over the
// (row block, column) grid of the N / h x h matrix of positions: block summaries (A), their scan down every column (B),
// the codes (C).  Both sorts are the segmented sort of the ordinary rounds (tc_seg.hpp), the groups group_kernel<REFINE>.
// Dense ranks (a text that ties nearly everything: flags from the rank array) and sparse ranks (a periodic stretch, a long run
// of one symbol inside a larger text: flags from the members through rank_of; row blocks without a position on path are
// skipped); chosen by the host when a plain round shed next to nothing.
#pragma once
#include "tc_common.hpp"
#include "tc_sa.hpp"   // RankLookup, rank_of

#define CHAIN_NONE 0xffffffffu
#define CHAIN_THREADS_LOG2 18     // (row block, column) cells the scan is cut into, about

struct ChainDims {
    u32 N, h;
    u32 rows;   // ceil(N / h)
    u32 bk;     // rows per block
    u32 nb;     // blocks
};

static inline ChainDims chain_dims(u64 N, u64 h) {
    ChainDims d;
    d.N = (u32)N; d.h = (u32)h;
    d.rows = (u32)((N + h - 1) / h);
    u64 nb = h >= (1ull << CHAIN_THREADS_LOG2) ? 1 : ((1ull << CHAIN_THREADS_LOG2) + h - 1) / h;
    if (nb > d.rows) nb = d.rows;
    if (nb < 1) nb = 1;
    d.bk = (u32)((d.rows + nb - 1) / nb);
    d.nb = (u32)((d.rows + d.bk - 1) / d.bk);
    return d;
}
// words of the block summaries (nb * h <= 2^18 + h < 2^19 whenever nb > 1), then one word per row block (nb <= 2^18):
// does the block hold a position that is on path at all
static inline size_t chain_any_offset() { return ((size_t)2 << CHAIN_THREADS_LOG2) + 64; }
static inline size_t chain_summ_words() { return chain_any_offset() + ((size_t)1 << CHAIN_THREADS_LOG2) + 64; }

#ifdef __HIPCC__

// ref[g] for every tied group: the rank its MIDDLE member (in the active set's order) sees at + h.  (Not the last one:
// round 0 leaves a group's members in the order of their positions, the plain rounds leave equal keys where they are --
// so the last member is the one nearest the end of the text, the very one that deviates on periodic text.)  The members
// of a group are contiguous in the active set and fill the group's slots: its size is the last member's slot - head + 1.
__global__ __launch_bounds__(256) void chain_ref_kernel(const u32 *__restrict__ slot, const u32 *__restrict__ idx,
                                                        const u32 *__restrict__ grp, RankLookup r, u32 m, u32 h,
                                                        u32 *__restrict__ ref) {
    __shared__ u16 s_lut[256];
    s_lut[threadIdx.x] = r.lut[threadIdx.x];
    __syncthreads();
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    const u32 g = grp[k];
    if (k + 1 < m && grp[k + 1] == g) return;
    u32 size = slot[k] - g + 1u;
    if ((u64)size > k + 1) size = (u32)(k + 1);   // (cannot happen while the invariant above holds)
    ref[g] = rank_of(r, s_lut, (u64)idx[k - size / 2] + h);
}

// on-path / sign bits of every position (one 64-bit word per wave and turn)
__global__ __launch_bounds__(256) void chain_flags_kernel(const u32 *__restrict__ isa, const u32 *__restrict__ ref, u32 N, u32 h,
                                                          u64 *__restrict__ pathbits, u64 *__restrict__ signbits, u32 nwords) {
    const u32 lane = threadIdx.x & 63;
    for (u64 w = ((u64)blockIdx.x * 256 + threadIdx.x) >> 6; w < nwords; w += ((u64)gridDim.x * 256) >> 6) {
        const u64 q = w * 64 + lane;
        bool on = false, sg = false;
        if (q < N) {
            const u32 r = ref[isa[q]];
            if (r != CHAIN_NONE) {
                const u32 nxt = q + h < N ? isa[q + h] : 0u;
                on = nxt == r;
                sg = nxt > r;
            }
        }
        const u64 bo = __ballot(on), bs = __ballot(sg);
        if (lane == 0) {
            pathbits[w] = bo;
            signbits[w] = bs;
        }
    }
}

// the same bits from the MEMBERS (sparse ranks: the tied set is a small part of the text, and the rank of a position comes
// through the rank table / the sorted keys -- rank_of): every member's position gets its bits by atomicOr into zeroed bitmaps
__global__ __launch_bounds__(256) void chain_flags_members_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ grp,
                                                                  RankLookup r, u32 m, u32 h, const u32 *__restrict__ ref,
                                                                  u64 *__restrict__ pathbits, u64 *__restrict__ signbits) {
    __shared__ u16 s_lut[256];
    s_lut[threadIdx.x] = r.lut[threadIdx.x];
    __syncthreads();
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < m; k += (u64)gridDim.x * 256) {
        const u32 i = idx[k];
        const u32 rf = ref[grp[k]];
        if (rf == CHAIN_NONE) continue;   // (every group of the set has one)
        const u32 nxt = rank_of(r, s_lut, (u64)i + h);
        if (nxt == rf) atomicOr((unsigned long long *)&pathbits[i >> 6], 1ull << (i & 63));
        else if (nxt > rf) atomicOr((unsigned long long *)&signbits[i >> 6], 1ull << (i & 63));
    }
}

// any[b]: does row block b (positions [b * bk * h, (b + 1) * bk * h)) hold a position on path?  A block without one is
// all terminals: its summaries are known and none of its positions needs a code (k = 0: the sign bit alone)
// (gridDim.y workgroups share a block's words: late rounds have few, huge row blocks; `any` is zeroed by the host)
__global__ __launch_bounds__(256) void chain_blockany_kernel(const u64 *__restrict__ pathbits, ChainDims d, u32 *__restrict__ any) {
    const u32 b = blockIdx.x;
    const u64 p0 = (u64)b * d.bk * d.h;
    u64 p1 = p0 + (u64)d.bk * d.h;
    if (p1 > d.N) p1 = d.N;
    const u64 w0 = p0 >> 6, w1 = (p1 - 1) >> 6;   // first / last word of the block
    u64 acc = 0;
    for (u64 w = w0 + (u64)blockIdx.y * 256 + threadIdx.x; w <= w1; w += (u64)gridDim.y * 256) {
        u64 x = pathbits[w];
        if (w == w0) x &= ~0ull << (p0 & 63);
        if (w == w1 && ((p1 & 63) != 0)) x &= (1ull << (p1 & 63)) - 1ull;
        acc |= x;
    }
    if (__ballot(acc != 0ull) != 0ull && lane_id() == 0) atomicOr(&any[b], 1u);
}

// state of a walk up a column: k on-path steps seen, s the terminal's sign; `thru`: no terminal met yet in this block
// (packed: bit 31 thru, bit 30 s, bits 29..0 k -- k < N / h < 2^30: the host takes h >= 4)
__device__ __forceinline__ u32 chain_pack(bool thru, u32 s, u32 k) { return (thru ? 0x80000000u : 0u) | (s << 30) | k; }

__device__ __forceinline__ bool chain_bit(const u64 *__restrict__ bits, u64 q) { return (bits[q >> 6] >> (q & 63)) & 1ull; }

// bits of rows row - 1, row - 2, .. (nr <= 4 of them) of column c: f = on path | sign << 1; a cell beyond the text (last row
// only; never stepped onto) reads as a terminal
__device__ __forceinline__ void chain_fetch4(const u64 *__restrict__ pathbits, const u64 *__restrict__ signbits, const ChainDims &d,
                                             u32 row, u32 nr, u32 c, u32 (&f)[4]) {
    u64 pw[4], sw[4];
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
        const u64 q = j < nr ? (u64)(row - 1u - j) * d.h + c : 0ull;
        const bool in = j < nr && q < d.N;
        pw[j] = in ? pathbits[q >> 6] : 0ull;
        sw[j] = in ? signbits[q >> 6] : 0ull;
    }
#pragma unroll
    for (u32 j = 0; j < 4; j++) {
        const u64 q = j < nr ? (u64)(row - 1u - j) * d.h + c : 0ull;
        f[j] = (u32)((pw[j] >> (q & 63)) & 1ull) | ((u32)((sw[j] >> (q & 63)) & 1ull) << 1);
    }
}

// phase A: summary of block b of column c -- walked from its last row up to its first
__global__ __launch_bounds__(256) void chain_scan_a_kernel(const u64 *__restrict__ pathbits, const u64 *__restrict__ signbits,
                                                           ChainDims d, const u32 *__restrict__ any, u32 *__restrict__ summ) {
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= (u64)d.nb * d.h) return;
    const u32 b = (u32)(t / d.h), c = (u32)(t - (u64)b * d.h);
    if (!any[b]) {   // (all terminals: what the block hands upwards is its first row's cell -- k = 0 and that cell's sign)
        const u64 q0 = (u64)b * d.bk * d.h + c;
        summ[t] = chain_pack(false, q0 < d.N && chain_bit(signbits, q0) ? 1u : 0u, 0);
        return;
    }
    const u32 row0 = b * d.bk;
    const u32 row1 = row0 + d.bk < d.rows ? row0 + d.bk : d.rows;
    bool thru = true;
    u32 k = 0, s = 0;
    for (u32 row = row1; row > row0;) {   // (four rows' bits fetched together: the addresses do not depend on the state)
        const u32 nr = row - row0 < 4u ? row - row0 : 4u;
        u32 f[4];
        chain_fetch4(pathbits, signbits, d, row, nr, c, f);
#pragma unroll
        for (u32 j = 0; j < 4; j++) {
            if (j >= nr) break;
            if (f[j] & 1u) k++;
            else { thru = false; k = 0; s = f[j] >> 1; }
        }
        row -= nr;
    }
    summ[t] = chain_pack(thru, s, k);
}

// phase B: per column, the state that enters every block from below (in place of its summary)
__global__ __launch_bounds__(256) void chain_scan_b_kernel(u32 *__restrict__ summ, ChainDims d) {
    const u32 c = blockIdx.x * 256 + threadIdx.x;
    if (c >= d.h) return;
    u32 k = 0, s = 0;
    // (eight summaries fetched together: the addresses do not depend on the carry -- one load per turn was nb memory latencies
    // in a row, 6 ms of a 1 GiB record's chain round at h = 42)
    for (u32 b = d.nb; b > 0;) {
        const u32 nr = b < 8u ? b : 8u;
        u32 x[8];
#pragma unroll
        for (u32 j = 0; j < 8; j++) x[j] = j < nr ? summ[(size_t)(b - 1u - j) * d.h + c] : 0u;
#pragma unroll
        for (u32 j = 0; j < 8; j++) {
            if (j >= nr) break;
            summ[(size_t)(b - 1u - j) * d.h + c] = chain_pack(false, s, k);
            if (x[j] & 0x80000000u) k += x[j] & 0x3fffffffu;
            else { k = x[j] & 0x3fffffffu; s = (x[j] >> 30) & 1u; }
        }
        b -= nr;
    }
}

// phase C: the codes -- s = 0: k, s = 1: 0xffffffff - k
__global__ __launch_bounds__(256) void chain_scan_c_kernel(const u64 *__restrict__ pathbits, const u64 *__restrict__ signbits,
                                                           ChainDims d, const u32 *__restrict__ any, const u32 *__restrict__ summ,
                                                           u32 *__restrict__ code) {
    const u64 t = (u64)blockIdx.x * 256 + threadIdx.x;
    if (t >= (u64)d.nb * d.h) return;
    const u32 b = (u32)(t / d.h), c = (u32)(t - (u64)b * d.h);
    if (!any[b]) return;   // (no position of the block is on path: their codes are their sign bits -- chain_code_of)
    const u32 row0 = b * d.bk;
    const u32 row1 = row0 + d.bk < d.rows ? row0 + d.bk : d.rows;
    u32 k = 0, s = 0;
    if (d.nb > 1) {
        const u32 x = summ[t];
        k = x & 0x3fffffffu;
        s = (x >> 30) & 1u;
    }
    for (u32 row = row1; row > row0;) {
        const u32 nr = row - row0 < 4u ? row - row0 : 4u;
        u32 f[4];
        chain_fetch4(pathbits, signbits, d, row, nr, c, f);
#pragma unroll
        for (u32 j = 0; j < 4; j++) {
            if (j >= nr) break;
            const u64 q = (u64)(row - 1u - j) * d.h + c;
            if (f[j] & 1u) k++;
            else { k = 0; s = f[j] >> 1; }
            if (q < d.N) code[q] = s ? 0xffffffffu - k : k;
        }
        row -= nr;
    }
}

// the code of position i: from the table when i is on path (its block has been walked), else k = 0 and the sign bit
__device__ __forceinline__ u32 chain_code_of(const u64 *__restrict__ pathbits, const u64 *__restrict__ signbits,
                                             const u32 *__restrict__ code, u32 i) {
    if ((pathbits[i >> 6] >> (i & 63)) & 1ull) return code[i];
    return ((signbits[i >> 6] >> (i & 63)) & 1ull) ? 0xffffffffu : 0u;
}

// TC_SA_TRACE: what the chain tables look like over the members -- [0] on path, [1] k = 0, [2] the largest k, [3] members whose
// group refers to itself
__global__ __launch_bounds__(256) void chain_diag_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ grp, u32 m,
                                                         const u64 *__restrict__ pathbits, const u64 *__restrict__ signbits,
                                                         const u32 *__restrict__ code, unsigned long long *__restrict__ out) {
    unsigned long long on = 0, k0 = 0, kmax = 0;
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < m; j += (u64)gridDim.x * 256) {
        const u32 i = idx[j];
        if ((pathbits[i >> 6] >> (i & 63)) & 1ull) on++;
        const u32 cd = chain_code_of(pathbits, signbits, code, i);
        const u32 k = (cd >> 31) ? ~cd : cd;
        if (k == 0) k0++;
        if (k > kmax) kmax = k;
    }
    atomicAdd(&out[0], on); atomicAdd(&out[1], k0); atomicMax(&out[2], kmax);
}

// pass 1: key2 = group << 32 | code of the member's position (vals_out: the suffix start sorted along -- dense ranks only)
__global__ __launch_bounds__(256) void chain_key1_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ grp,
                                                         const u64 *__restrict__ pathbits, const u64 *__restrict__ signbits,
                                                         const u32 *__restrict__ code, u32 m, u64 *__restrict__ keys,
                                                         u32 *__restrict__ vals_out) {
    const u64 k = (u64)blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    const u32 i = idx[k];
    keys[k] = ((u64)grp[k] << 32) | chain_code_of(pathbits, signbits, code, i);
    if (vals_out) vals_out[k] = i;
}

// pass 2: key2 = group << 32 | the rank the member's terminal sees: rank[i + (k + 1) h]
__global__ __launch_bounds__(256) void chain_key2_kernel(const u32 *__restrict__ idx, const u32 *__restrict__ grp,
                                                         const u64 *__restrict__ pathbits, const u64 *__restrict__ signbits,
                                                         const u32 *__restrict__ code, RankLookup r, u32 m, u32 h,
                                                         u64 *__restrict__ keys, u32 *__restrict__ vals_out) {
    __shared__ u16 s_lut[256];
    s_lut[threadIdx.x] = r.lut[threadIdx.x];
    __syncthreads();
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < m; k += (u64)gridDim.x * 256) {
        const u32 i = idx[k];
        const u32 cd = chain_code_of(pathbits, signbits, code, i);
        const u32 steps = (cd >> 31) ? ~cd : cd;
        keys[k] = ((u64)grp[k] << 32) | rank_of(r, s_lut, (u64)i + ((u64)steps + 1) * h);
        if (vals_out) vals_out[k] = i;
    }
}

#endif  // __HIPCC__
