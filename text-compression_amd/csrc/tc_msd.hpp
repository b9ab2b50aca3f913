// tc_msd.hpp -- round 0 of the suffix sort for small alphabets as an MSD radix sort.
//
// Replaces (for long texts over <= 15 byte values that look iid: the DNA benchmark record) the LSD
// passes of tc_radix.hpp + finish_kernel in front of `DS.unstableSortOn snd` (reference
// BWT/Internal.hs:130).
//
// Why MSD here.  An LSD pass must be STABLE, which on a GPU means: ranks by wave-ordered match,
// tiles in ticket order, a decoupled look-back per digit, and scattered segments of ~33 pairs that
// start at arbitrary alignment (every seam a partially written 128-byte line; measured in round 1:
// 7.8 ms for the access pattern alone against 5.0 ms for the same bytes in whole lines).  An MSD
// pass only has to PARTITION: the order inside a bucket is irrelevant, because the bucket is sorted
// again by the next field.  That removes the look-back, the ticket and the stable ranking (one LDS
// atomic per pair ranks it), and it allows software write combining: a workgroup keeps, per digit,
// the pairs that do not yet fill a 16-pair group (128 B of keys + 64 B of values) in LDS and only
// ever stores whole, aligned groups.  Positions are exact (no atomics, no holes): a counting kernel
// with the SAME static work split gives every (workgroup, parent bucket) segment its private range in
// every child.  Measured (1 GiB ACGTN): HBM traffic = 1.007 x the algorithmic bytes, and a level runs at
// ~0.8 of the device's copy rate -- its tile is a read burst and a write burst that HBM serves one
// after the other, with the LDS work hidden behind them.
//
//   level 1   text -> (key, idx) partitioned by field 0           1 B read, 12 B written per suffix
//   level 2   inside each level-1 bucket by field 1; its counting pass also gathers the joint
//             (field 1, field 2) counts, i.e. level 3's child counts  12 B read, 12 B written (+ 8 B counted)
//   level 3   inside each level-2 bucket by field 2; "aligned": every parent belongs to one
//             workgroup, so it needs no counting pass of its own   12 B read, 12 B written
//   finish    the level-3 buckets (9 symbols on DNA, ~550 suffixes at 1 GiB) are taken in chunks of
//             <= MSDF_CH consecutive buckets / MSDF_TILE pairs and ordered by the remaining 32 key bits
//             in LDS (counting sort by 17 key bits, then ranks inside the ~2-member bins); SA, last
//             column and the tied set leave from there (same contract as finish_kernel)
//                                                                 12 B read, 5 B written
// A text with a level-3 bucket above MSDF_CAP, or with more than 2^18 suffixes tied beyond the key
// (repeats, runs), takes the LSD way instead; the host does not even try when the byte entropy or the
// collision sample say the text is not iid-like (tc_encode_host.hpp).
#pragma once
#include "tc_sa.hpp"

#ifndef MSD_NT
#define MSD_NT 1024
#endif
#ifndef MSD_ITEMS
#define MSD_ITEMS 8
#endif
#define MSD_TILE (MSD_NT * MSD_ITEMS)
#ifndef MSD_GROUP
#define MSD_GROUP 16          // pairs per store group: one 128-byte line of keys, half a line of values
#endif
#define MSD_GLOG (MSD_GROUP == 32 ? 5 : MSD_GROUP == 16 ? 4 : 3)
#ifndef MSD_SUB
#define MSD_SUB 1             // ranking counters per digit (1, 2 or 4)
#endif
#ifndef MSD_BPC
#define MSD_BPC 1             // partition / counting workgroups per CU (LDS permitting)
#endif
#define MSD_LEVELS 3
#define MSDF_CAP_SMALL 2048                // pairs per chunk of the finish kernel's instance for small buckets
#ifndef MSDF_KO_NT
#define MSDF_KO_NT 256                     // threads of that instance in its key-only form
#endif
#ifndef MSDF_BIG_NT
#define MSDF_BIG_NT 512
#endif
#ifndef MSDF_BIG_ITEMS
#define MSDF_BIG_ITEMS 12
#endif
#define MSDF_CAP_BIG (MSDF_BIG_NT * MSDF_BIG_ITEMS)   // ... and of the one for buckets of a few thousand pairs

// One partition level.  Parents are numbered by their digit path (level 1: one parent; level 2:
// 256; level 3: 65536); an absent path is a parent with count 0.  Tiles never straddle parents.
// Workgroup b of G owns tiles [T*b/G, T*(b+1)/G); its run inside parent q is segment q + b.
struct MsdLevel {
    const u32 *pstart;   // [nparents] first position of the parent in the input arrays
    const u32 *pcnt;     // [nparents]
    const u32 *tpre;     // [nparents + 1] exclusive prefix of ceil(pcnt / MSD_TILE); tpre[nparents] = T
    u32 nparents;
    int shift;           // digit = (key >> shift) & 255
    u32 *seg;            // [(nparents + G) * 256] per-segment digit counts, then bases (in place)
    u32 *cstart;         // [nparents * 256] children: first position
    u32 *ccnt;           // [nparents * 256] children: count
    // "aligned" level: every parent belongs to ONE workgroup -- parent q to workgroup pstart[q] * G / ntot --
    // so a segment is a whole parent and its digit counts do not depend on where its pairs lie: they come
    // from cnt_in, filled by the PREVIOUS level's counting pass (joint counts of its digit and the next
    // one), and this level needs no counting pass of its own.
    int aligned;
    u32 ntot;
    const u32 *cnt_in;   // [nparents * 256] (aligned levels)
    u32 *flags;          // device word: bit 3 = the joint counts of an aligned level do not add up
    u64 *dbg;            // MSD_PROFILE builds: [16] cycles per phase of workgroup 0 (wave 0; [9]: the last wave)
};

#ifdef __HIPCC__

__device__ __forceinline__ u32 msd_tile_lo(u32 T, u32 b, u32 G) { return (u32)(((u64)T * b) / G); }
__device__ __forceinline__ u32 msd_block_of_tile(u32 T, u32 t, u32 G) {
    return (u32)((((u64)t + 1) * G - 1) / T);   // largest b with tile_lo(b) <= t
}

__global__ void msd_root_kernel(u32 *pstart, u32 *pcnt, u32 N, u32 *maxchild) {
    pstart[0] = 0;
    pcnt[0] = N;
    *maxchild = 0;
}

// aligned levels: workgroup of parent q, and the first parent of workgroup b
__device__ __forceinline__ u32 msd_block_of_parent(const MsdLevel &L, u32 q, u32 G) {
    return (u32)(((u64)L.pstart[q] * G) / L.ntot);
}
__device__ __forceinline__ u32 msd_first_parent(const MsdLevel &L, u32 b, u32 G) {
    if (b == 0) return 0;
    if (b >= G) return L.nparents;
    u32 lo = 0, hi = L.nparents;   // first q in [0, nparents] with block_of(q) >= b (pstart is monotone)
    while (lo < hi) {
        const u32 mid = (lo + hi) >> 1;
        if (msd_block_of_parent(L, mid, G) >= b) hi = mid; else lo = mid + 1;
    }
    return lo;
}
// tiles [t0, t1) of workgroup b (call from one thread)
__device__ __forceinline__ void msd_block_range(const MsdLevel &L, u32 b, u32 G, u32 *t0, u32 *t1) {
    if (L.aligned) {
        *t0 = L.tpre[msd_first_parent(L, b, G)];
        *t1 = L.tpre[msd_first_parent(L, b + 1, G)];
    } else {
        const u32 T = L.tpre[L.nparents];
        *t0 = (u32)(((u64)T * b) / G);
        *t1 = (u32)(((u64)T * (b + 1)) / G);
    }
}

// tiles per parent -> exclusive prefix.  One block; nparents <= 65536.
__global__ __launch_bounds__(1024) void msd_prep_kernel(const u32 *__restrict__ pcnt, u32 nparents,
                                                        u32 *__restrict__ tpre) {
    __shared__ u32 s_scan[1024 / 64 + 1];
    const u32 per = (nparents + 1023) / 1024;
    const u32 lo = threadIdx.x * per;
    u32 sum = 0;
    if (per == 64 && nparents == 65536u) {
        // level 3: a thread's 64 parents by sixteen 16-byte loads issued together (a load per loop turn, twice, was 128
        // latencies in a row: 130 us of a launch that does nothing else)
        uint4 v[16];
        const uint4 *pv = reinterpret_cast<const uint4 *>(pcnt + lo);
#pragma unroll
        for (int i = 0; i < 16; i++) v[i] = pv[i];
        u32 t[64];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            t[4 * i] = (v[i].x + MSD_TILE - 1) / MSD_TILE; t[4 * i + 1] = (v[i].y + MSD_TILE - 1) / MSD_TILE;
            t[4 * i + 2] = (v[i].z + MSD_TILE - 1) / MSD_TILE; t[4 * i + 3] = (v[i].w + MSD_TILE - 1) / MSD_TILE;
        }
#pragma unroll
        for (int i = 0; i < 64; i++) sum += t[i];
        u32 tot;
        u32 run = block_excl_sum<1024>(sum, s_scan, &tot);
        uint4 *ov = reinterpret_cast<uint4 *>(tpre + lo);
#pragma unroll
        for (int i = 0; i < 16; i++) {
            uint4 o;
            o.x = run; run += t[4 * i];
            o.y = run; run += t[4 * i + 1];
            o.z = run; run += t[4 * i + 2];
            o.w = run; run += t[4 * i + 3];
            ov[i] = o;
        }
        if (threadIdx.x == 0) tpre[nparents] = tot;
        return;
    }
    for (u32 i = 0; i < per; i++) {
        const u32 q = lo + i;
        if (q < nparents) sum += (pcnt[q] + MSD_TILE - 1) / MSD_TILE;
    }
    u32 tot;
    u32 run = block_excl_sum<1024>(sum, s_scan, &tot);
    for (u32 i = 0; i < per; i++) {
        const u32 q = lo + i;
        if (q < nparents) {
            tpre[q] = run;
            run += (pcnt[q] + MSD_TILE - 1) / MSD_TILE;
        }
    }
    if (threadIdx.x == 0) tpre[nparents] = tot;
}

// The walk of a workgroup over its tiles is the same code in the counting and the partition kernel,
// so both see the same segments.  Thread 0 advances it; the tile's description travels through LDS.
__device__ __forceinline__ u32 msd_find_parent(const u32 *tpre, u32 nparents, u32 t) {
    u32 lo = 0, hi = nparents;   // largest q with tpre[q] <= t
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (tpre[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;   // tpre[lo] <= t < tpre[lo + 1] because equal prefixes belong to empty parents before lo
}

struct MsdTileInfo {
    u32 base;     // first position of the tile in the input arrays
    u32 valid;    // pairs in the tile (0: no such tile)
    u32 q;        // parent
    u32 last;     // 1: last tile of its segment (parent changes or range ends)
};
// Thread 0's cursor: the parent of the tile last asked for stays in registers, so a tile's description
// is arithmetic; the tables are read only when the walk enters another parent.
struct MsdCur {
    u32 q, tq0, tq1, ps, pc;
};
__device__ __forceinline__ void msd_cur_init(const MsdLevel &L, MsdCur &c, u32 t) {
    c.q = msd_find_parent(L.tpre, L.nparents, t);
    c.tq0 = L.tpre[c.q]; c.tq1 = L.tpre[c.q + 1]; c.ps = L.pstart[c.q]; c.pc = L.pcnt[c.q];
}
__device__ __forceinline__ void msd_cur_info(const MsdLevel &L, MsdCur &c, u32 t, u32 t_end, MsdTileInfo *out) {
    if (t >= t_end) {
        out->base = 0; out->valid = 0; out->q = 0; out->last = 0;
        return;
    }
    if (t >= c.tq1) {
        u32 q = c.q + 1;
        while (L.tpre[q + 1] <= t) q++;
        c.q = q; c.tq0 = L.tpre[q]; c.tq1 = L.tpre[q + 1]; c.ps = L.pstart[q]; c.pc = L.pcnt[q];
    }
    const u32 off = (t - c.tq0) * MSD_TILE;
    out->base = c.ps + off;
    out->valid = c.pc - off < MSD_TILE ? c.pc - off : MSD_TILE;
    out->q = c.q;
    out->last = (t + 1 >= t_end || t + 1 >= c.tq1) ? 1u : 0u;
}

// field 0 of suffix i straight from the text (level 1 only): the first s symbols in base B
struct MsdTextDigit {
    const u8 *text;
    u32 n;
    u32 B, s;
    u16 lut[256];
    u32 hash_ok, hsh, tlo, thi;   // byte -> code by v_perm from a register table (RadixKeyGen, tc_radix.hpp)
};

// Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8; tc_dbg_dispatch_probe shows it), and the
// static work split hands workgroup b the b-th slice of the array, so every XCD touches every part of the
// 26 GB a level moves.  MSD_XCD_MAP = 1 gives XCD x the x-th eighth instead (levels 2 and 3 write where they
// read).  Measured at 1 GiB: no difference in either of the two step-time modes -- left off.
#ifndef MSD_XCD_MAP
#define MSD_XCD_MAP 0
#endif
__device__ __forceinline__ u32 msd_logical_wg(u32 b, u32 G) {
    return (MSD_XCD_MAP && (G & 7u) == 0) ? (b & 7u) * (G >> 3) + (b >> 3) : b;
}

// ---- per-segment digit counts ------------------------------------------------------------------
// keys: digit = (key >> shift) & 255.  TEXT (level 1): digit = field 0 of suffix i = its first s
// symbols in base B, straight from the text (two overlapping word loads per 4 suffixes where the
// segment lies inside the text; s <= 5 there, else bytes).  A segment = the run of a workgroup's tiles
// inside one parent = one contiguous range of positions: no per-tile synchronisation.
// JOINT: besides, the counts of (digit, next digit) per parent are accumulated into joint_out
// [(parent * 256 + digit) * 256 + next digit] -- the child counts of the NEXT level's parents, which is
// then an aligned level without a counting pass.  LDS table of 128 rows x 256 32-bit counters (no cell can
// overflow, whatever the text: a poly-A tract puts millions of pairs into one cell): the rows are the digits
// whose symbols are all real (sigma^s <= 128 of them; MsdJointRows maps digit -> row); a digit that
// contains the end marker belongs to one of the text's last suffixes and goes to joint_out directly.
struct MsdJointRows {
    u8 row[256];   // digit -> row of the LDS table, 0xff: none
    u8 dig[128];   // row -> digit
};
template <bool TEXT, bool JOINT>
__global__ __launch_bounds__(MSD_NT) void msd_count_kernel(MsdLevel L, const u64 *__restrict__ keys,
                                                           MsdTextDigit td, u32 *__restrict__ joint_out,
                                                           MsdJointRows jr) {
    __shared__ u32 s_cnt[256];
    __shared__ u16 s_lut[TEXT ? 256 : 1];
    __shared__ u32 s_seg[4];   // q, lo, hi, next tile
    __shared__ u32 s_joint[JOINT ? 32768 : 1];
    const u32 tid = threadIdx.x, G = gridDim.x, b = msd_logical_wg(blockIdx.x, G);
    if (tid < 256) s_cnt[tid] = 0;
    if (TEXT && tid < 256) s_lut[tid] = td.lut[tid];
    __shared__ u8 s_row[JOINT ? 256 : 1];
    if (JOINT) {
        for (u32 i = tid; i < 32768; i += MSD_NT) s_joint[i] = 0;
        if (tid < 256) s_row[tid] = jr.row[tid];
    }
    const u32 T = L.tpre[L.nparents];
    const u32 t0 = msd_tile_lo(T, b, G), t1 = msd_tile_lo(T, b + 1, G);
    if (t0 >= t1) return;
    u32 t = t0;
    u32 q = 0;
    if (tid == 0) q = msd_find_parent(L.tpre, L.nparents, t0);
    while (t < t1) {
        if (tid == 0) {
            while (L.tpre[q + 1] <= t) q++;
            const u32 tq0 = L.tpre[q], tq1 = L.tpre[q + 1];
            const u32 te = tq1 < t1 ? tq1 : t1;
            const u32 ps = L.pstart[q], pc = L.pcnt[q];
            const u32 lo = ps + (t - tq0) * MSD_TILE;
            const u64 hi = (u64)ps + (u64)(te - tq0) * MSD_TILE;
            s_seg[0] = q; s_seg[1] = lo; s_seg[2] = hi < (u64)ps + pc ? (u32)hi : ps + pc; s_seg[3] = te;
        }
        __syncthreads();
        const u32 sq = s_seg[0], lo = s_seg[1], hi = s_seg[2];
        t = s_seg[3];
        if (TEXT) {
            const bool words = td.s <= 5 && (u64)lo + 8 < td.n && ((((uintptr_t)td.text) + lo) & 15) == 0;
            u32 p = lo;
            if (words) {
                // rounds of 16 positions per thread: one 16-byte load + the 4 bytes behind it
                // (only rounds that end 8 bytes before the end of the text; the rest goes byte by byte)
                const u32 span = (u64)hi + 8 <= td.n ? hi - lo : td.n - 8 - lo;
                const u32 nfull = span / (16 * MSD_NT);
                const u32 *tw = reinterpret_cast<const u32 *>(td.text + lo);
                auto fetch = [&](u32 r, u32 *x) {
                    const u32 wi = (r * MSD_NT + tid) * 4;
#pragma unroll
                    for (int q = 0; q < 5; q++) x[q] = tw[wi + q];
                };
                auto tally = [&](const u32 *x) {
                    u32 cd[20];
                    if (td.hash_ok) {   // (block-uniform) four codes by one v_perm_b32 instead of four LDS reads
#pragma unroll
                        for (int q = 0; q < 5; q++) {
                            const u32 y = __builtin_amdgcn_perm(td.thi, td.tlo, (x[q] >> td.hsh) & 0x07070707u);
#pragma unroll
                            for (int j = 0; j < 4; j++) cd[4 * q + j] = (y >> (8 * j)) & 255u;
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < 20; j++) cd[j] = s_lut[(x[j >> 2] >> (8 * (j & 3))) & 255u];
                    }
#pragma unroll
                    for (int k = 0; k < 16; k++) {
                        u32 g = 0;
#pragma unroll
                        for (int j = 0; j < 5; j++)
                            if (j < (int)td.s) g = g * td.B + cd[k + j];
                        atomicAdd(&s_cnt[g], 1u);
                    }
                };
                // four rounds' loads in flight per thread (one round -- 20 bytes a lane -- left the pass at 2.1 TB/s:
                // one workgroup per CU, bound by the bytes in flight)
                u32 r = 0;
                for (; r + 4 <= nfull; r += 4) {
                    u32 xa[5], xb[5], xc[5], xd[5];
                    fetch(r, xa); fetch(r + 1, xb); fetch(r + 2, xc); fetch(r + 3, xd);
                    tally(xa); tally(xb); tally(xc); tally(xd);
                }
                for (; r < nfull; r++) {
                    u32 x[5];
                    fetch(r, x);
                    tally(x);
                }
                p = lo + nfull * 16 * MSD_NT;
            }
            for (u64 i0 = (u64)p + tid; i0 < hi; i0 += MSD_NT) {
                u64 i = i0;
                u32 g = 0;
                for (u32 j = 0; j < td.s; j++, i++) g = g * td.B + (i < td.n ? (u32)s_lut[td.text[i]] : 0u);
                atomicAdd(&s_cnt[g], 1u);
            }
        } else {
            auto add = [&](u64 k) {
                const u32 dd = (u32)(k >> (L.shift - 8)) & 0xffffu;   // digit, next digit
                if (JOINT) {
                    // (ONE atomic per key: the digit's own count is the sum of its row of the joint table, taken when
                    // the table is flushed at the segment's end; a digit without a row -- it holds the end marker: one
                    // of the text's last suffixes -- is counted directly)
                    const u32 r = s_row[dd >> 8];
                    if (r != 0xffu) atomicAdd(&s_joint[(r << 8) | (dd & 255u)], 1u);
                    else {
                        atomicAdd(&s_cnt[dd >> 8], 1u);
                        atomicAdd(&joint_out[((size_t)sq * 256 + (dd >> 8)) * 256 + (dd & 255u)], 1u);
                    }
                } else {
                    atomicAdd(&s_cnt[dd >> 8], 1u);
                }
            };
            u32 p = lo + tid;
            for (; p + 3 * MSD_NT < hi; p += 4 * MSD_NT) {
                const u64 k0 = keys[p], k1 = keys[p + MSD_NT], k2 = keys[p + 2 * MSD_NT], k3 = keys[p + 3 * MSD_NT];
                add(k0); add(k1); add(k2); add(k3);
            }
            for (; p < hi; p += MSD_NT) add(keys[p]);
        }
        __syncthreads();
        if (JOINT) {
            for (u32 w = tid; w < 32768; w += MSD_NT) {
                const u32 v = s_joint[w];
                if (v) {
                    atomicAdd(&joint_out[((size_t)sq * 256 + jr.dig[w >> 8]) * 256 + (w & 255u)], v);
                    atomicAdd(&s_cnt[jr.dig[w >> 8]], v);   // (the row's sum: the digit's count in this segment)
                    s_joint[w] = 0;
                }
            }
            __syncthreads();
        }
        if (tid < 256) {
            L.seg[((size_t)sq + b) * 256 + tid] = s_cnt[tid];
            s_cnt[tid] = 0;
        }
        // (the next round's barrier orders these resets before the next atomics)
    }
}

// ---- counts -> child ranges and per-segment bases ---------------------------------------------
// One block per parent.  scalars[0] (optional): atomicMax of the child counts.
__global__ __launch_bounds__(256) void msd_scan_kernel(MsdLevel L, u32 G, u32 *maxchild) {
    __shared__ u32 s_scan[256 / 64 + 1];
    const u32 q = blockIdx.x, d = threadIdx.x;
    const u32 cnt = L.pcnt[q];
    const size_t c = (size_t)q * 256 + d;
    if (cnt == 0) {
        L.ccnt[c] = 0;
        L.cstart[c] = L.pstart[q];
        return;
    }
    if (L.aligned) {   // one segment: the parent itself, counts from the previous level's joint table
        const u32 tot = L.cnt_in[c];
        u32 all;
        const u32 excl = block_excl_sum<256>(tot, s_scan, &all);
        const u32 start = L.pstart[q] + excl;
        L.cstart[c] = start;
        L.ccnt[c] = tot;
        L.seg[((size_t)q + msd_block_of_parent(L, q, G)) * 256 + d] = start;
        if (d == 0 && all != cnt) atomicOr(L.flags, 8u);
        if (maxchild) {
            u32 m = tot;
#pragma unroll
            for (int x = 32; x >= 1; x >>= 1) {
                const u32 o = __shfl_xor(m, x, 64);
                m = m > o ? m : o;
            }
            if ((d & 63) == 0 && m > MSDF_CAP_SMALL) atomicMax(maxchild, m);
        }
        return;
    }
    const u32 T = L.tpre[L.nparents];
    const u32 bf = msd_block_of_tile(T, L.tpre[q], G), bl = msd_block_of_tile(T, L.tpre[q + 1] - 1, G);
    // (workgroups between bf and bl whose tile range is empty -- fewer tiles than workgroups -- own no
    // segment and wrote nothing)
    u32 tot = 0;
    // (level 1 is ONE parent split over all G workgroups: a thread sums G counts.  The loads do not depend on each other --
    // read unconditionally, eight in flight -- only the adds do)
#pragma unroll 8
    for (u32 b = bf; b <= bl; b++) {
        const u32 v = L.seg[((size_t)q + b) * 256 + d];
        tot += msd_tile_lo(T, b, G) < msd_tile_lo(T, b + 1, G) ? v : 0u;
    }
    u32 all;
    const u32 excl = block_excl_sum<256>(tot, s_scan, &all);
    u32 run = L.pstart[q] + excl;
    L.cstart[c] = run;
    L.ccnt[c] = tot;
#pragma unroll 8
    for (u32 b = bf; b <= bl; b++) {
        const size_t o = ((size_t)q + b) * 256 + d;
        const u32 v = L.seg[o];
        if (msd_tile_lo(T, b, G) >= msd_tile_lo(T, b + 1, G)) continue;
        L.seg[o] = run;
        run += v;
    }
    if (maxchild) {
        u32 m = tot;
#pragma unroll
        for (int x = 32; x >= 1; x >>= 1) {
            const u32 o = __shfl_xor(m, x, 64);
            m = m > o ? m : o;
        }
        if ((d & 63) == 0 && m > MSDF_CAP_SMALL) atomicMax(maxchild, m);
    }
}

// ---- one partition pass -------------------------------------------------------------------------
// KEYGEN: level 1, keys are built from the text (layout of keybuild_kernel in tc_sa.hpp: P fields
// of s symbols in base B, w = 8 bits each, from bit 63 down; low byte = the preceding text byte).
//
// Per tile (8192 pairs, 8 per thread): rank inside the digit by one LDS atomic per pair (any order will
// do), the tile sorted by digit into the staging area, then for every digit the whole 16-pair groups of
// (carry ++ the tile's segment) are stored -- 128 B of keys and 64 B of values, aligned -- and what is
// left becomes the carry.  The loads of tile t + 1 are issued before tile t is ranked and land
// while it is staged; they are consumed before tile t's stores are issued, so no wave ever waits for a
// store.
typedef u32 msd_u32x4 __attribute__((ext_vector_type(4)));
// VALS = false (round 3): the level moves KEYS ONLY -- 8 + 8 instead of 12 + 12 bytes per suffix.  An encode
// (BWT -> MTF -> RLE) needs the last column, which rides in the key's low byte; the suffix START is needed only
// for the few suffixes that stay tied beyond the key, and those are found again afterwards by one pass over the
// text (tied_probe_kernel, tc_sa.hpp).  Callers that want the suffix array itself keep VALS = true.
template <bool KEYGEN, bool VALS = true>
__global__ __launch_bounds__(MSD_NT) void msd_partition_kernel(MsdLevel L, const u64 *kin, const u32 *vin,
                                                               u64 *kout, u32 *vout, const u8 *text,
                                                               RadixKeyGen kg) {
    // (no __restrict__ on purpose: loads that may alias the stores keep their place in program order)
    // staging of the tile sorted by digit; the key-generation image overlays it
    __shared__ __attribute__((aligned(16))) u64 s_keys[MSD_TILE];
    __shared__ __attribute__((aligned(16))) u32 s_vals[VALS ? MSD_TILE : 4];
    // pairs that do not fill a group yet, per digit (slots [0, r); the first ph slots of a
    // segment's first group are phantoms standing for the positions before the segment's range)
    __shared__ __attribute__((aligned(16))) u64 c_keys[256 * MSD_GROUP];
    __shared__ __attribute__((aligned(16))) u32 c_vals[VALS ? 256 * MSD_GROUP : 4];
    // MSD_SUB counters per digit (a lane uses counter lane % MSD_SUB): fewer lanes of a wave meet on one
    // LDS address in the ranking atomics; a digit's sub-segments lie side by side in the staging area
    __shared__ u32 s_cnt[256 * MSD_SUB], s_dstart[256 * MSD_SUB], s_r[256], s_cur[256], s_ph[256];
    __shared__ u32 s_scan[16];
    __shared__ u64 s_lmask[4];
#ifdef MSD_PROFILE
    __shared__ u64 s_dbg[2];
    if (threadIdx.x == 0) { s_dbg[0] = 0; s_dbg[1] = 0; }
#endif
    __shared__ u16 s_live[256];
    __shared__ u16 s_klut[KEYGEN ? 256 : 1];
    // KG_SPLIT (round 4; the key-only level 1, whose LDS has the room): the key-generation image -- the tile's text bytes
    // and its field bytes -- has its OWN region instead of lying over the staging area, so that the image of tile t + 1 is
    // made while tile t is still staged and stored: its text is written there between the landing of the prefetch and
    // barrier B3, its field bytes are computed beside the store phase (S4), and the keys are assembled right behind the
    // next B0 -- no barrier of its own (the overlaid image needed three per tile, in front of the ranking).
    constexpr bool KG_SPLIT = KEYGEN && !VALS && MSD_ITEMS == 8;
    __shared__ __attribute__((aligned(16))) u8 s_kr[KG_SPLIT ? MSD_TILE + 16 + 80 + 32 : 16];
    __shared__ __attribute__((aligned(16))) u8 s_kg8[KG_SPLIT ? MSD_TILE + 64 : 16];
    __shared__ u32 s_scan2[2];
    static_assert(MSD_GROUP == 8 || MSD_GROUP == 16 || MSD_GROUP == 32, "group = 8, 16 or 32 pairs");
    static_assert(((size_t)MSD_TILE * (VALS ? 12 : 8) + 256 * MSD_GROUP * (VALS ? 12 : 8) + 8192 + (KG_SPLIT ? 2 * MSD_TILE + 256 : 0)) * (VALS ? 1 : MSD_BPC) <= 163840, "LDS budget");
    __shared__ MsdTileInfo s_info[4];

    const u32 tid = threadIdx.x, G = gridDim.x, b = msd_logical_wg(blockIdx.x, G);
    if (KEYGEN && tid < 256) s_klut[tid] = kg.lut[tid];
    if (tid < 256) { s_r[tid] = 0; s_ph[tid] = 0; s_cur[tid] = 0; }
    for (u32 i = tid; i < 256 * MSD_SUB; i += MSD_NT) s_cnt[i] = 0;
    if (tid == 0) {
        u32 ta = 0, tb = 0;
        if (!(L.aligned && (*L.flags & 8u))) msd_block_range(L, b, G, &ta, &tb);   // bad joint counts: do nothing
        s_scan2[0] = ta; s_scan2[1] = tb;
    }
    __syncthreads();
    const u32 t0 = s_scan2[0], t1 = s_scan2[1];
    if (t0 >= t1) return;
    MsdCur cs = {};
    if (tid == 0) {
        msd_cur_init(L, cs, t0);
        msd_cur_info(L, cs, t0, t1, &s_info[0]);
        msd_cur_info(L, cs, t0 + 1, t1, &s_info[1]);
        msd_cur_info(L, cs, t0 + 2, t1, &s_info[2]);
    }
    __syncthreads();

    // key-generation image (overlays the staging area)
    constexpr int KG_PRE = 16, KG_SLOTS = MSD_TILE + KG_PRE + 80;
    u16 *k_c = reinterpret_cast<u16 *>(s_keys);
    u16 *k_g = k_c + KG_SLOTS;
    u8 *k_r = reinterpret_cast<u8 *>(k_g + KG_SLOTS);
    static_assert(KG_SLOTS * 5 <= MSD_TILE * 8, "key-generation image must fit under the staging keys");
    const u32 kg_units = KEYGEN ? (KG_PRE + MSD_TILE + kg.P * kg.s + kg.s + 15) / 16 : 0;

    u64 key[MSD_ITEMS];
    u32 val[MSD_ITEMS];
    msd_u32x4 raw = {0, 0, 0, 0}, nraw = {0, 0, 0, 0};
    // KEYGEN: 16 text bytes per thread from position base - KG_PRE + 16 * tid (bytes outside the text: 0).
    // A tile is an "edge" tile when some unit reaches outside the text or the text is not 16-byte
    // aligned (first / last tile only): those units are read byte by byte.
    auto kg_edge = [&](const MsdTileInfo ti) {
        return ti.base < (u32)KG_PRE || (u64)ti.base - KG_PRE + (u64)kg_units * 16 > (u64)kg.n_text ||
               ((((uintptr_t)text) + ti.base) & 15) != 0;
    };
    auto kg_load_plain = [&](const MsdTileInfo ti) {   // ordinary loads (edge tiles, first tile)
        msd_u32x4 rw = {0, 0, 0, 0};
        if (tid < kg_units) {
            const i64 p0 = (i64)ti.base - KG_PRE + (i64)tid * 16;
            const i64 nt = (i64)kg.n_text;
            if (p0 >= 0 && p0 + 16 <= nt && ((((uintptr_t)text) + (u64)p0) & 15) == 0) {
                rw = *reinterpret_cast<const msd_u32x4 *>(text + p0);
            } else {
                u32 x[4] = {0, 0, 0, 0};
#pragma unroll 1
                for (int j = 0; j < 16; j++) {
                    const i64 pp = p0 + j;
                    const u32 c = (pp >= 0 && pp < nt) ? (u32)text[pp] : 0u;
                    x[j >> 2] |= c << (8 * (j & 3));
                }
                rw.x = x[0]; rw.y = x[1]; rw.z = x[2]; rw.w = x[3];
            }
        }
        return rw;
    };
    // The prefetch of the next tile is issued by inline asm so that it stays where it is written (the
    // compiler sinks ordinary loads to their first use, which serialises load latency and tile work).
    // The compiler does not track these loads, so (i) they are unconditional -- every lane loads from a
    // clamped, valid address; a register they fill is defined by the asm alone, never merged with
    // another value before it has landed -- and (ii) every such register passes through land() before
    // anything reads it.
    // plain levels: thread t holds the pairs of two quads of 4 CONSECUTIVE positions (4 t .. 4 t + 3 of each
    // 4096-pair half): 16-byte loads -- 6 load instructions per thread and tile instead of 16, so the waves
    // get past the issue of the prefetch quickly (with 16 narrow loads per thread the load queue filled up
    // and every wave sat at the issue until HBM had taken the burst: ~10 000 of a tile's 24 000 cycles)
    msd_u32x4 nk4[MSD_ITEMS / 2], nv4[MSD_ITEMS / 4];
    auto pof = [&](int k) -> u32 {
        return KEYGEN ? (u32)k * MSD_NT + tid : (u32)(k >> 2) * (4 * MSD_NT) + 4 * tid + (u32)(k & 3);
    };
    // a tile whose whole extent lies inside the arrays can be fetched without bounds (pairs past `valid`
    // are read and ignored); the one tile that reaches past the arrays' end is loaded pair by pair
    auto tile_safe = [&](const MsdTileInfo x) { return x.valid != 0 && (u64)x.base + MSD_TILE <= (u64)L.ntot; };
    auto load_plain = [&](const MsdTileInfo f) {
#pragma unroll
        for (int k = 0; k < MSD_ITEMS; k++) {
            const u32 p = pof(k);
            key[k] = 0;
            val[k] = 0;
            if (p < f.valid) {
                key[k] = kin[f.base + p];
                if (VALS) val[k] = vin[f.base + p];
            }
        }
    };
    auto prefetch = [&](const MsdTileInfo cur_ti, const MsdTileInfo nx) {
        if (KEYGEN) {
            const bool use = nx.valid != 0 && !kg_edge(nx);
            const u32 base = use ? nx.base : (cur_ti.base >= (u32)KG_PRE && !kg_edge(cur_ti) ? cur_ti.base : 0u);
            const u32 unit = tid < kg_units ? tid : kg_units - 1;
            // (fallback address: the start of the text rounded up to 16 bytes -- any valid 16 bytes)
            const u8 *src = use || base ? text + ((u64)base - KG_PRE + (u64)unit * 16)
                                        : reinterpret_cast<const u8 *>((((uintptr_t)text) + 15) & ~(uintptr_t)15);
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nraw) : "v"(src) : "memory");
        } else {
            const bool safe = tile_safe(nx);
            const u32 nb = safe ? nx.base : 0u;   // (an unsafe or absent next tile: any valid tile)
            const u32 nv = safe ? nx.valid : (u32)MSD_TILE;
#pragma unroll
            for (int g = 0; g < MSD_ITEMS / 4; g++) {
                // (a quad wholly past the tile's last pair -- the last tile of a parent -- re-reads the
                // tile's first quad instead of the next parent's pairs: no HBM traffic for ignored data)
                u32 gs = (u32)g * (4 * MSD_NT) + 4 * tid;
                gs = gs < nv ? gs : 0u;
                const u64 *kp = kin + nb + gs;
                const u32 *vp = vin + nb + gs;
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nk4[2 * g]) : "v"(kp) : "memory");
                asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nk4[2 * g + 1]) : "v"(kp + 2) : "memory");
                if (VALS) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(nv4[g]) : "v"(vp) : "memory");
            }
        }
    };
    auto land = [&]() {
        if (KEYGEN) {
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(nraw) : : "memory");
        } else {
            static_assert(MSD_ITEMS == 4 || MSD_ITEMS == 8, "land() lists the prefetch registers");
            if (!VALS) {
                if (MSD_ITEMS == 8)
                    asm volatile("s_waitcnt vmcnt(0)"
                                 : "+v"(nk4[0]), "+v"(nk4[1]), "+v"(nk4[MSD_ITEMS / 2 - 2]), "+v"(nk4[MSD_ITEMS / 2 - 1])
                                 :
                                 : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(nk4[0]), "+v"(nk4[1]) : : "memory");
            } else if (MSD_ITEMS == 8)
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(nk4[0]), "+v"(nk4[1]), "+v"(nk4[MSD_ITEMS / 2 - 2]), "+v"(nk4[MSD_ITEMS / 2 - 1]), "+v"(nv4[0]),
                               "+v"(nv4[MSD_ITEMS / 4 - 1])
                             :
                             : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(nk4[0]), "+v"(nk4[1]), "+v"(nv4[0]) : : "memory");
        }
    };
    const u32 B = kg.B;
    (void)B;
    auto gen_group = [&](u8 *k_r, u8 *k_g8, u32 grp) {   // positions 8 grp .. 8 grp + 7
        const u32 *rp = reinterpret_cast<const u32 *>(k_r + KG_PRE + 8 * grp);
        const u32 x0 = rp[0], x1 = rp[1], x2 = rp[2];
        u32 c[10];
        if (kg.hash_ok) {   // (block-uniform) the codes of four bytes by one v_perm_b32 from a table in registers
            const u32 y0 = __builtin_amdgcn_perm(kg.thi, kg.tlo, (x0 >> kg.hsh) & 0x07070707u);
            const u32 y1 = __builtin_amdgcn_perm(kg.thi, kg.tlo, (x1 >> kg.hsh) & 0x07070707u);
            const u32 y2 = __builtin_amdgcn_perm(kg.thi, kg.tlo, (x2 >> kg.hsh) & 0x07070707u);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                c[j] = (y0 >> (8 * j)) & 255u;
                c[4 + j] = (y1 >> (8 * j)) & 255u;
            }
            c[8] = y2 & 255u;
            c[9] = (y2 >> 8) & 255u;
        } else {
#pragma unroll
        for (int j = 0; j < 4; j++) {
            c[j] = (u32)s_klut[(x0 >> (8 * j)) & 255u];
            c[4 + j] = (u32)s_klut[(x1 >> (8 * j)) & 255u];
        }
        c[8] = (u32)s_klut[x2 & 255u];
        c[9] = (u32)s_klut[(x2 >> 8) & 255u];
        }
        u32 lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const u32 g = kg.s == 3 ? (c[j] * B + c[j + 1]) * B + c[j + 2] : kg.s == 2 ? c[j] * B + c[j + 1] : c[j];
            if (j < 4) lo |= g << (8 * j);
            else hi |= g << (8 * (j - 4));
        }
        *reinterpret_cast<uint2 *>(k_g8 + 8 * grp) = make_uint2(lo, hi);
    };
    auto kg_fast = [&](const MsdTileInfo x) { return KEYGEN && kg.s <= 3 && MSD_ITEMS == 8 && x.valid != 0 && !kg_edge(x); };
    auto seg_init = [&](u32 qq) {   // threads < 256
        const u32 sb = L.seg[((size_t)qq + b) * 256 + tid];
        s_cur[tid] = sb & ~(u32)(MSD_GROUP - 1);
        s_ph[tid] = sb & (MSD_GROUP - 1);
        s_r[tid] = sb & (MSD_GROUP - 1);
        // the digits this parent has at all (a DNA field: 125 of 256): they are dealt to the lane groups of
        // the store phase in turn, so that every lane group serves the same number of digits
        const u64 m = __ballot(L.ccnt[(size_t)qq * 256 + tid] != 0);
        if ((tid & 63) == 0) s_lmask[tid >> 6] = m;
    };
    if (tid < 256) seg_init(s_info[0].q);
    // The first tile comes by ordinary loads.  Their registers then pass through an empty asm: the compiler
    // waits for them HERE and from then on regards them as asm-defined.  Otherwise its wait-count model
    // carries "may still be in flight" into the loop and puts a vmcnt wait in front of the first use of
    // every key register -- which, the asm-issued prefetch being invisible to that model, waits for the
    // whole prefetch of the next tile: the overlap was gone (measured: 13 000 of a tile's 24 000 cycles).
    auto detach = [&]() {
        if (KEYGEN) {
            asm volatile("" : "+v"(raw) : : "memory");
        } else if (!VALS) {
            asm volatile("" : "+v"(key[0]), "+v"(key[1]), "+v"(key[2]), "+v"(key[3]) : : "memory");
            if (MSD_ITEMS == 8)
                asm volatile("" : "+v"(key[MSD_ITEMS - 4]), "+v"(key[MSD_ITEMS - 3]), "+v"(key[MSD_ITEMS - 2]), "+v"(key[MSD_ITEMS - 1]) : : "memory");
        } else {
            asm volatile("" : "+v"(key[0]), "+v"(key[1]), "+v"(key[2]), "+v"(key[3]), "+v"(val[0]), "+v"(val[1]), "+v"(val[2]), "+v"(val[3]) : : "memory");
            if (MSD_ITEMS == 8)
                asm volatile("" : "+v"(key[MSD_ITEMS - 4]), "+v"(key[MSD_ITEMS - 3]), "+v"(key[MSD_ITEMS - 2]), "+v"(key[MSD_ITEMS - 1]),
                                  "+v"(val[MSD_ITEMS - 4]), "+v"(val[MSD_ITEMS - 3]), "+v"(val[MSD_ITEMS - 2]), "+v"(val[MSD_ITEMS - 1]) : : "memory");
        }
    };
    if (KEYGEN) {
        if (!kg_edge(s_info[0])) raw = kg_load_plain(s_info[0]);
    } else {
        load_plain(s_info[0]);
    }
    detach();
    if (KG_SPLIT && kg_fast(s_info[0])) {   // the first tile's image, before the loop
        if (tid < kg_units) *reinterpret_cast<uint4 *>(s_kr + tid * 16) = make_uint4(raw.x, raw.y, raw.z, raw.w);
        __syncthreads();
        gen_group(s_kr, s_kg8, tid);
        if (tid < 3) gen_group(s_kr, s_kg8, MSD_NT + tid);
    }

    for (u32 t = t0; t < t1; t++) {
        const u32 slot = (t - t0) & 3u;
        const MsdTileInfo ti = s_info[slot];
#ifdef MSD_PROFILE
        u64 tq[8];
        tq[0] = __builtin_readcyclecounter();
#endif
        if (tid == 0) msd_cur_info(L, cs, t + 3, t1, &s_info[(slot + 3) & 3u]);
        __syncthreads();   // (B0) s_cnt zeroed; staging free; carries / ranges of this segment in place
#ifdef MSD_PROFILE
        const u64 t_b0 = __builtin_readcyclecounter();
#endif
        const MsdTileInfo nx = s_info[(slot + 1) & 3u];
        prefetch(ti, nx);   // tile t + 1: in flight while tile t is ranked and staged
        if (KEYGEN && kg_edge(ti)) {   // first / last tile: ordinary loads, bounds-checked
            raw = kg_load_plain(ti);
            detach();
        }
        if (!KEYGEN && t != t0 && !tile_safe(ti)) {   // the tile that reaches past the arrays' end: pair by pair
            load_plain(ti);
            detach();
        }
        const bool fastkg = KEYGEN && kg.s <= 3 && MSD_ITEMS == 8 && !kg_edge(ti);
        if (fastkg) {
            // interior tile, fields of <= 3 symbols.  Two steps.  (A) every thread turns ITS 8 positions into field
            // values g(p) = the s symbols from p on (a byte each: B^s <= 256) and leaves them in LDS -- 10 LUT
            // reads and 16 multiply-adds per thread; (B) thread t assembles the keys of the 8 CONSECUTIVE suffixes
            // 8 t .. 8 t + 7 (which thread handles which pair is free: the tile is permuted anyway) from the 32
            // field bytes behind its first position: field f of suffix k is g(k + s f), so a key is seven byte
            // selections -- three v_perm_b32 per 32-bit half, constant selectors.  (The first version built all
            // 26 field values of a thread's window in registers, every thread for itself: 28 LUT reads, ~250
            // VALU per thread; level 1: 5.44 -> 5.12 ms.  Issuing the level's small prefetch after the keys instead of before
            // them -- right behind B0 the memory pipeline is still full of the previous tile's stores -- moves the wait,
            // it does not remove it: the level is bound by its 13 GB of scattered stores, 43 % of the fill rate.)
            u8 *kr = KG_SPLIT ? s_kr : k_r;
            u8 *k_g8 = KG_SPLIT ? s_kg8 : reinterpret_cast<u8 *>(k_c);   // field bytes of positions 0 .. MSD_TILE + 23 (the u16 images are unused here)
            if (!KG_SPLIT) {
                if (tid < kg_units) *reinterpret_cast<uint4 *>(kr + tid * 16) = make_uint4(raw.x, raw.y, raw.z, raw.w);
                __syncthreads();
                gen_group(kr, k_g8, tid);
                if (tid < 3) gen_group(kr, k_g8, MSD_NT + tid);   // the 6 s <= 18 positions behind the tile that its last suffixes reach
                __syncthreads();
            }
            // (KG_SPLIT: this tile's image was made during the previous tile -- or before the loop -- and B0 has published it)
            {
                const u32 *gp = reinterpret_cast<const u32 *>(k_g8 + 8 * tid);
                u32 D[8];
#pragma unroll
                for (int i = 0; i < 8; i++) D[i] = gp[i];
                // the byte before each suffix: text bytes p - 1 .. p + 6
                const u32 *rq = reinterpret_cast<const u32 *>(kr + KG_PRE + 8 * tid - 4);
                const u32 r0 = rq[0], r1 = rq[1], r2 = rq[2];
                const u32 PR[2] = {__builtin_amdgcn_alignbyte(r1, r0, 3), __builtin_amdgcn_alignbyte(r2, r1, 3)};
                // byte j of the field bytes / byte k of PR into the wanted byte lane
                auto pick2 = [&](u32 X, int bx, u32 Y, int by, bool upper) -> u32 {
                    // upper: X.byte[bx] -> result byte 3, Y.byte[by] -> result byte 2; else -> bytes 1, 0
                    const u32 sel = upper ? ((u32)(4 + bx) << 24) | ((u32)by << 16) : ((u32)(4 + bx) << 8) | (u32)by;
                    return __builtin_amdgcn_perm(X, Y, sel);
                };
                auto build = [&](auto S) {
                    constexpr int SS = decltype(S)::value;
#pragma unroll
                    for (int k = 0; k < 8; k++) {
                        const int j0 = k, j1 = k + SS, j2 = k + 2 * SS, j3 = k + 3 * SS, j4 = k + 4 * SS, j5 = k + 5 * SS, j6 = k + 6 * SS;
                        const u32 hA = pick2(D[j0 >> 2], j0 & 3, D[j1 >> 2], j1 & 3, true);
                        const u32 hB = pick2(D[j2 >> 2], j2 & 3, D[j3 >> 2], j3 & 3, false);
                        const u32 lA = pick2(D[j4 >> 2], j4 & 3, D[j5 >> 2], j5 & 3, true);
                        const u32 lB = pick2(D[j6 >> 2], j6 & 3, PR[k >> 2], k & 3, false);
                        const u32 hi = __builtin_amdgcn_perm(hA, hB, 0x07060100u);
                        const u32 lo = __builtin_amdgcn_perm(lA, lB, 0x07060100u);
                        key[k] = ((u64)hi << 32) | (u64)lo;
                    }
                };
                if (kg.s == 3) build(std::integral_constant<int, 3>{});
                else if (kg.s == 2) build(std::integral_constant<int, 2>{});
                else build(std::integral_constant<int, 1>{});
            }
            if (!KG_SPLIT) __syncthreads();   // image dead: the staging area may be written
        } else if (KEYGEN) {
            const i64 nt = (i64)kg.n_text;
            if (tid < kg_units) {
                const i64 p0 = (i64)ti.base - KG_PRE + (i64)tid * 16;
                const u32 xx[4] = {raw.x, raw.y, raw.z, raw.w};
                u32 cw[8];
                const bool inside = p0 >= 0 && p0 + 16 <= nt;
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    const u32 ba = (xx[j >> 2] >> (8 * (j & 3))) & 255u, bb = (xx[(j + 1) >> 2] >> (8 * ((j + 1) & 3))) & 255u;
                    u32 ca = s_klut[ba], cb = s_klut[bb];
                    if (!inside) {
                        const i64 pa = p0 + j, pb = p0 + j + 1;
                        if (pa < 0 || pa >= nt) ca = 0;
                        if (pb < 0 || pb >= nt) cb = 0;
                    }
                    cw[j >> 1] = ca | (cb << 16);
                }
                uint4 *dc = reinterpret_cast<uint4 *>(k_c + tid * 16);
                dc[0] = make_uint4(cw[0], cw[1], cw[2], cw[3]);
                dc[1] = make_uint4(cw[4], cw[5], cw[6], cw[7]);
                *reinterpret_cast<uint4 *>(k_r + tid * 16) = make_uint4(raw.x, raw.y, raw.z, raw.w);
            }
            __syncthreads();
            const u32 gslots = KG_PRE + MSD_TILE + kg.P * kg.s;
            const u32 B = kg.B;
            if (kg.s == 3) {
                for (u32 x = tid; x < gslots; x += MSD_NT) k_g[x] = (u16)((k_c[x] * B + k_c[x + 1]) * B + k_c[x + 2]);
            } else {
                for (u32 x = tid; x < gslots; x += MSD_NT) {
                    u32 g = 0;
                    for (u32 j = 0; j < kg.s; j++) g = g * B + k_c[x + j];
                    k_g[x] = (u16)g;
                }
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < MSD_ITEMS; k++) {
                const u32 p = k * MSD_NT + tid;
                if (p < ti.valid) {
                    const u32 x = KG_PRE + p;
                    u64 kk = 0;
#pragma unroll
                    for (int f = 0; f < 7; f++)
                        if (f < (int)kg.P) kk |= (u64)k_g[x + f * kg.s] << (56 - 8 * f);
                    key[k] = kk | (u64)k_r[x - 1];
                }
            }
            __syncthreads();   // image dead: the staging area may be written
        }
#ifdef MSD_PROFILE
        tq[1] = __builtin_readcyclecounter();
#endif
        // (S1) digit + rank inside the digit (any order: the partition need not be stable)
        u32 dig[MSD_ITEMS], rnk[MSD_ITEMS];
#pragma unroll
        for (int k = 0; k < MSD_ITEMS; k++) {
            const u32 p = pof(k);
            dig[k] = (u32)(key[k] >> L.shift) & 255u;
            if (p < ti.valid) rnk[k] = atomicAdd(&s_cnt[dig[k] * MSD_SUB + (tid % MSD_SUB)], 1u);
        }
#ifdef MSD_PROFILE
        const u64 t_s1 = __builtin_readcyclecounter();   // S1 issued and its LDS returns waited for (s_memtime is lgkm too)
        if ((tid & 63) == 0) {   // slowest wave from the end of B0 to its arrival at B1, and to the end of the prefetch issue
            atomicMax((unsigned long long *)&s_dbg[0], (unsigned long long)(t_s1 - t_b0));
            atomicMax((unsigned long long *)&s_dbg[1], (unsigned long long)(tq[1] - t_b0));
        }
#endif
        __syncthreads();   // (B1)
#ifdef MSD_PROFILE
        tq[2] = __builtin_readcyclecounter();
        if (b == 0 && L.dbg && (tid == 0 || tid == MSD_NT - 64)) L.dbg[tid == 0 ? 8 : 9] += t_s1 - tq[1];
        if (b == 0 && L.dbg && tid == 0) { L.dbg[10] += s_dbg[0]; L.dbg[11] += s_dbg[1]; s_dbg[0] = 0; s_dbg[1] = 0; }
#endif
        // (S2) start of every (digit, sub-counter) segment in the staging area
        {
            constexpr u32 NC = 256 * MSD_SUB;
            if (tid < NC) {
                const u32 c = s_cnt[tid];
                const u32 inc = wave_incl_sum(c);
                if ((tid & 63) == 63) s_scan[tid >> 6] = inc;
                s_dstart[tid] = inc - c;   // wave-relative for now
            }
            __syncthreads();   // (B2a)
            if (tid < NC) {
                u32 base = 0;
                for (u32 i = 0; i < (tid >> 6); i++) base += s_scan[i];
                s_dstart[tid] += base;
            }
            if (tid < 256) {   // list of the parent's digits (rebuilt per tile: a few instructions)
                const u32 w = tid >> 6;
                const u64 mine = s_lmask[w];
                if ((mine >> (tid & 63)) & 1ull) {
                    u32 pos = (u32)__popcll(mine & ((1ull << (tid & 63)) - 1ull));
                    for (u32 i = 0; i < w; i++) pos += (u32)__popcll(s_lmask[i]);
                    s_live[pos] = (u16)tid;
                }
            }
        }
        __syncthreads();   // (B2)
#ifdef MSD_PROFILE
        tq[3] = __builtin_readcyclecounter();
#endif
        // (S3) the tile, sorted by digit, into the staging area
#pragma unroll
        for (int k = 0; k < MSD_ITEMS; k++) {
            const u32 p = pof(k);
            if (p < ti.valid) {
                const u32 o = s_dstart[dig[k] * MSD_SUB + (tid % MSD_SUB)] + rnk[k];
                s_keys[o] = key[k];
                // (KEYGEN: the value is the suffix start -- which suffix slot k of this thread holds
                // depends on the key-generation form used for the tile)
                if (VALS) s_vals[o] = KEYGEN ? ti.base + (fastkg ? 8 * tid + k : p) : val[k];
            }
        }
#ifdef MSD_PROFILE
        tq[4] = __builtin_readcyclecounter();
#endif
        // the next tile's pairs take the registers over; nothing younger than their loads is outstanding
        land();
#ifdef MSD_PROFILE
        tq[5] = __builtin_readcyclecounter();
#endif
        const bool kg_next = KG_SPLIT && kg_fast(nx);
        if (KEYGEN) {
            raw = nraw;
            if (kg_next && tid < kg_units) *reinterpret_cast<uint4 *>(s_kr + tid * 16) = make_uint4(raw.x, raw.y, raw.z, raw.w);
        } else {
#pragma unroll
            for (int k = 0; k < MSD_ITEMS; k++) {
                const msd_u32x4 q4 = nk4[k >> 1];
                key[k] = (k & 1) ? ((u64)q4.w << 32 | q4.z) : ((u64)q4.y << 32 | q4.x);
                if (VALS) {
                    const msd_u32x4 v4 = nv4[k >> 2];
                    val[k] = (k & 3) == 0 ? v4.x : (k & 3) == 1 ? v4.y : (k & 3) == 2 ? v4.z : v4.w;
                }
            }
        }
        __syncthreads();   // (B3)
#ifdef MSD_PROFILE
        tq[6] = __builtin_readcyclecounter();
#endif
        if (kg_next) {   // the field bytes of tile t + 1, beside the stores of tile t
            gen_group(s_kr, s_kg8, tid);
            if (tid < 3) gen_group(s_kr, s_kg8, MSD_NT + tid);
        }
        // (S4) every digit belongs to one lane group of MSD_GROUP lanes (digits g, g + NG, ...): it stores
        // the whole groups of (carry ++ segment) -- group k = elements [G k, G k + G) -- and then keeps what
        // is left as the new carry.  No other lane group touches the digit's state: no barrier in between.
        {
            constexpr u32 NG = MSD_NT / MSD_GROUP;
            const u32 lg = tid / MSD_GROUP, l = tid % MSD_GROUP;
            const u32 nlive = (u32)(__popcll(s_lmask[0]) + __popcll(s_lmask[1]) + __popcll(s_lmask[2]) + __popcll(s_lmask[3]));
            for (u32 j = lg; j < nlive; j += NG) {
                const u32 d = s_live[j];
                u32 c = 0;
#pragma unroll
                for (int x = 0; x < MSD_SUB; x++) c += s_cnt[d * MSD_SUB + x];
                if (c == 0) continue;
                const u32 r = s_r[d], ds = s_dstart[d * MSD_SUB], cur = s_cur[d], ph = s_ph[d];
                const u32 ng = (r + c) >> MSD_GLOG;
                for (u32 k = 0; k < ng; k++) {
                    const u32 e = k * MSD_GROUP + l;
                    u64 kk;
                    u32 vv = 0;
                    if (e < r) {
                        kk = c_keys[d * MSD_GROUP + e];
                        if (VALS) vv = c_vals[d * MSD_GROUP + e];
                    } else {
                        kk = s_keys[ds + e - r];
                        if (VALS) vv = s_vals[ds + e - r];
                    }
                    if (k > 0 || l >= ph) {
                        kout[cur + e] = kk;
                        if (VALS) vout[cur + e] = vv;
                    }
                }
                const u32 newr = ng ? ((r + c) & (MSD_GROUP - 1)) : r + c;
                if (ng) {
                    if (l < newr) {
                        const u32 o = ds + (ng * MSD_GROUP - r) + l;
                        c_keys[d * MSD_GROUP + l] = s_keys[o];
                        if (VALS) c_vals[d * MSD_GROUP + l] = s_vals[o];
                    }
                } else if (l >= r && l < newr) {
                    c_keys[d * MSD_GROUP + l] = s_keys[ds + l - r];
                    if (VALS) c_vals[d * MSD_GROUP + l] = s_vals[ds + l - r];
                }
                if (l == 0) {
                    s_r[d] = newr;
                    if (ng) {
                        s_cur[d] = cur + ng * MSD_GROUP;
                        s_ph[d] = 0;
                    }
#pragma unroll
                    for (int x = 0; x < MSD_SUB; x++) s_cnt[d * MSD_SUB + x] = 0;
                }
            }
        }
#ifdef MSD_PROFILE
        tq[7] = __builtin_readcyclecounter();
        if (b == 0 && tid == 0 && L.dbg) {
            // [0] top..B0 (tile cursor, barrier)  [1] key generation  [2] S1 + B1  [3] S2 + barriers
            // [4] S3  [5] wait for the prefetch  [6] B3  [7] S4
            L.dbg[0] += tq[1] - tq[0]; L.dbg[2] += tq[2] - tq[1]; L.dbg[3] += tq[3] - tq[2]; L.dbg[4] += tq[4] - tq[3];
            L.dbg[5] += tq[5] - tq[4]; L.dbg[6] += tq[6] - tq[5]; L.dbg[7] += tq[7] - tq[6]; L.dbg[1] += 1;
        }
#endif
        if (ti.last) {
            // end of the segment: the carries go to their exact places (partial lines, once per
            // segment and digit); then the next segment's ranges are taken
            __syncthreads();
            for (u32 x = tid; x < 256 * MSD_GROUP; x += MSD_NT) {
                const u32 d = x / MSD_GROUP, j = x % MSD_GROUP;
                if (j >= s_ph[d] && j < s_r[d]) {
                    const u32 pos = s_cur[d] + j;
                    kout[pos] = c_keys[x];
                    if (VALS) vout[pos] = c_vals[x];
                }
            }
            __syncthreads();
            if (t + 1 < t1 && tid < 256) seg_init(s_info[(slot + 1) & 3u].q);
        }
    }
}

// ---- finish: every level-3 bucket ordered by its remaining key bits ----------------------------------
// (the equal-mass bin map of the finish kernels: see msd_finish_ko_kernel below)
struct MsdFinishLut {
    u32 a[256];      // cumulative share of the values below, in 1/64 bins (0 .. MSDK_BPC * 64)
    u32 w[256];      // a[v + 1] - a[v]
    u32 c[256];      // cumulative share in 1/65536 (<= 65535)
};
#define MSDK_BPC 1024
__global__ __launch_bounds__(256) void msd_finish_lut_kernel(const u32 *__restrict__ cnt, MsdFinishLut *out) {
    __shared__ u64 s_x[257];
    __shared__ u64 s_tot;
    const u32 t = threadIdx.x;
    s_x[t] = cnt[t];
    __syncthreads();
    if (t == 0) {
        u64 run = 0;
        for (int i = 0; i < 256; i++) { const u64 c = s_x[i]; s_x[i] = run; run += c; }
        s_x[256] = run;
        s_tot = run ? run : 1;
    }
    __syncthreads();
    const u64 tot = s_tot;
    // (65535, not 65536: a and w share a 32-bit word in the finish kernel, and the bin inside the child stays < MSDK_BPC)
    const u32 lo = (u32)(s_x[t] * (u64)(MSDK_BPC * 64 - 1) / tot), hi = t == 255 ? (u32)(MSDK_BPC * 64 - 1) : (u32)(s_x[t + 1] * (u64)(MSDK_BPC * 64 - 1) / tot);
    out->a[t] = lo;
    out->w[t] = hi - lo;
    const u64 c16 = s_x[t] * 65536ull / tot;
    out->c[t] = c16 > 65535 ? 65535u : (u32)c16;
}


struct MsdFinishArgs {
    const u64 *keys;
    const u32 *vals;
    const u32 *pcnt;     // level-3 parents (one block each)
    const u32 *cstart;   // [nparents * 256] level-3 buckets
    const u32 *ccnt;
    u32 *sa_out;
    u8 *L;
    u32 *out_slot, *out_idx, *out_grp;   // tied set, FIN_REGIONS regions of rcap entries
    u32 *rcount;
    u32 rcap;
    u32 *counters;       // [1] bit 2: a bucket above the instance's chunk was met (the caller takes the LSD path);
                         // bit 1: (SORTEDKEYS) such buckets were listed for msd_whole_kernel; [2]: how many
    u64 *kout;           // SORTEDKEYS: the keys in final order
    u32 *whole_list;     // SORTEDKEYS: (start, length) of the over-long buckets
    u32 whole_cap;
    u32 *out_khi;        // VALS = false: key bits 40..63 of every tied member (out_idx then holds bits 8..39)
    const MsdFinishLut *lut;   // bins = equal-mass intervals of (field 3, field 4) instead of key bits (null: key bits)
};

// Over-long buckets of the SORTEDKEYS instance (repeats, poly-A: no chunk holds them): msd_finish_kernel only
// lists them (start, length); here every member leaves in place as ONE tied group -- (slot, suffix, group =
// bucket start), dealt over the regions of the tied list -- for the doubling rounds to order, as
// finish_fix_kernel does for the over-long buckets of the LSD way.  Its own kernel so that the rare case does
// not set the register budget of the finish kernel (194 instead of 128 VGPRs when it was a branch there).
// Launched unconditionally with a fixed grid; the list length is read on the device.
#define MSDW_NT 512
__global__ __launch_bounds__(MSDW_NT) void msd_whole_kernel(MsdFinishArgs a) {
    __shared__ u32 s_o;
    const u32 tid = threadIdx.x;
    const u32 nlist = a.counters[2] < a.whole_cap ? a.counters[2] : a.whole_cap;
    for (u32 e = blockIdx.x; e < nlist; e += gridDim.x) {
        const u32 start = a.whole_list[2 * e], tot = a.whole_list[2 * e + 1];
        for (u32 base = 0; base < tot; base += MSDW_NT * 4) {
            const u32 cnt = tot - base < MSDW_NT * 4 ? tot - base : MSDW_NT * 4;
            const u32 reg = (e + base / (MSDW_NT * 4)) % FIN_REGIONS;   // (one bucket may hold more than a region)
            const u32 rbase = reg * a.rcap;
            __syncthreads();
            if (tid == 0) s_o = atomicAdd(a.rcount + reg * FIN_RSTRIDE, cnt);
            __syncthreads();
            const u32 o0 = s_o;
            for (u32 i = tid; i < cnt; i += MSDW_NT) {
                const u32 p = start + base + i;
                const u64 k = a.keys[p];
                const u32 v = a.vals[p];
                a.sa_out[p] = v;
                a.L[p] = (u8)(k & 0xff);
                a.kout[p] = k;
                if (o0 + i < a.rcap) {
                    a.out_slot[rbase + o0 + i] = p;
                    a.out_idx[rbase + o0 + i] = v;
                    a.out_grp[rbase + o0 + i] = start;
                }
            }
        }
    }
}

// One workgroup per level-3 parent; its buckets (the children) are taken in chunks of consecutive
// children -- at most MSDF_CH of them, at most MSDF_TILE pairs -- which are contiguous in memory.  A
// chunk is counting-sorted in LDS by (child, field 3, top MSDF_XB bits of field 4) = 16 + MSDF_XB key bits
// from bit 47 down, one bin per value (16-bit counters, two per LDS word): the bins lie in key order, so
// a pair's final place is its bin's start plus its rank inside the (~1-member) bin by all 32 remaining
// bits (39..8).  SA / last column leave coalesced.  Members with equal remaining bits are tied beyond
// the key: (slot, suffix, group = first slot of the equal run) go to the tied list exactly as
// finish_kernel emits them.
// Two instances: <256, 8, 4, 1> for buckets of a few hundred pairs (5-letter DNA at 1 GiB: ~550; chunks of <= 4
// buckets / 2048 pairs, several workgroups per CU) and <1024, 8, 1, 5> for buckets of a few thousand (4-letter
// DNA at 1 GiB: ~4096; one bucket per chunk of <= 8192 pairs, 8192 bins).  SORTEDKEYS: the keys are also
// written in their final order (kout), so that rank lookups of the doubling rounds are a binary search
// instead of a count inside an unsorted bucket.
// VALS = false (the key-only levels of an encode, round 3): no suffix starts come in and no suffix array goes
// out -- 8 bytes read and 1 written per suffix instead of 12 and 5; a tied member leaves with its KEY (56 bits
// in out_idx / out_khi) instead of its suffix start.
template <int MSDF_NT, int MSDF_ITEMS, int MSDF_CH, int MSDF_XB, bool SORTEDKEYS, bool VALS = true>
__global__ __launch_bounds__(MSDF_NT) void msd_finish_kernel(MsdFinishArgs a) {
    constexpr u32 MSDF_TILE = MSDF_NT * MSDF_ITEMS;
    constexpr u32 BINS_PER_CHILD = 256u << MSDF_XB;
    constexpr u32 MAXBINS = MSDF_CH * BINS_PER_CHILD;
    constexpr int BPT = MAXBINS / MSDF_NT;   // bins per thread in the scan (even)
    static_assert(BPT * MSDF_NT == MAXBINS && BPT % 2 == 0, "bins must divide among the threads in pairs");
    static_assert(MSDF_TILE < 65536, "16-bit bin counters");
    __shared__ u32 s_off[MAXBINS / 2 + 2];   // [bin] u16: count, then exclusive offset; [nbins] = pairs
    __shared__ u32 s_low[MSDF_TILE];   // remaining key bits, in bin order
    __shared__ u32 s_idx[VALS ? MSDF_TILE : 4];   // suffix starts, in final order
    __shared__ __attribute__((aligned(16))) u8 s_L[MSDF_TILE + 8];   // preceding bytes, in final order (VALS = false: shifted by start & 3,
                                                                      // so that aligned words of the image are aligned words of L)
    __shared__ u32 s_cc[256], s_cs[256];
    __shared__ u32 s_chunk[5];         // first child, children spanned, first position, pairs, next child
    __shared__ u32 s_scan[MSDF_NT / 64 + 1];
    const u32 q = blockIdx.x;
    if (a.pcnt[q] == 0 || (a.counters[1] & 8u)) return;   // (bit 3: the level-3 counts did not add up)
    const u32 tid = threadIdx.x, l = tid & 63;
    for (u32 i = tid; i < 256; i += MSDF_NT) {
        s_cc[i] = a.ccnt[(size_t)q * 256 + i];
        s_cs[i] = a.cstart[(size_t)q * 256 + i];
    }
    const u32 region = blockIdx.x % FIN_REGIONS;
    u32 *rctr = a.rcount + region * FIN_RSTRIDE;
    const u32 rbase = region * a.rcap;
    // bins by key BITS use a fraction of a child's bins only (a field is s symbols in base B inside 8 bits: 4-letter DNA has
    // 64 live values of 256), so a bin holds several keys and the ranking loop below runs as long as the wave's longest
    // bin; with the table of msd_finish_lut_kernel a bin is an equal-mass interval of (field 3, field 4) -- any monotone
    // map is correct, the ranking compares all of bits 39..8 -- and holds ~0.5 keys
    __shared__ u32 s_aw[256];
    __shared__ u16 s_cq[256];
    const bool use_lut = a.lut != nullptr;
    if (use_lut)
        for (u32 i = tid; i < 256; i += MSDF_NT) {
            s_aw[i] = a.lut->a[i] | (a.lut->w[i] << 16);
            s_cq[i] = (u16)a.lut->c[i];
        }
    auto bin_of = [&](u64 key, u32 c0) -> u32 {
        if (use_lut) {
            const u32 aw = s_aw[(u32)(key >> 32) & 255u];
            const u32 c = s_cq[(u32)(key >> 24) & 255u];
            const u32 inner = ((aw & 0xffffu) + ((u32)__umul24(aw >> 16, c) >> 16)) >> (8 - MSDF_XB);   // < BINS_PER_CHILD: a + w <= 65535
            return (((u32)(key >> 40) & 255u) - c0) * BINS_PER_CHILD + inner;
        }
        return ((u32)(key >> (32 - MSDF_XB)) & ((0x10000u << MSDF_XB) - 1u)) - c0 * BINS_PER_CHILD;
    };
    auto off_at = [&](u32 bin) { return (s_off[bin >> 1] >> (16 * (bin & 1))) & 0xffffu; };
    u32 ch = 0;
    while (true) {
        __syncthreads();   // (also: the previous chunk's copy-out has read the LDS images)
        if (tid == 0) {
            while (ch < 256 && s_cc[ch] == 0) ch++;
            u32 c0 = ch, tot = 0, span = 0;
            if (ch < 256 && s_cc[ch] > MSDF_TILE) {
                // a bucket no chunk can hold (repeats, poly-A).  SORTEDKEYS instance: its members leave as ONE
                // tied group, in place, for the doubling rounds to order (bit 1: they share only the levels'
                // symbols) -- what finish_fix_kernel does for over-long buckets of the LSD way.  Else: bit 2,
                // the caller takes the LSD way.
                if (SORTEDKEYS) {
                    const u32 e = atomicAdd(&a.counters[2], 1u);
                    if (e < a.whole_cap) {
                        a.whole_list[2 * e] = s_cs[ch];
                        a.whole_list[2 * e + 1] = s_cc[ch];
                    }
                    atomicOr(&a.counters[1], e < a.whole_cap ? 2u : 4u);
                } else {
                    atomicOr(&a.counters[1], 4u);
                }
                ch++;
                span = 1;   // (tot = 0: nothing to do here)
            } else {
                while (ch < 256 && span < MSDF_CH && tot + s_cc[ch] <= MSDF_TILE) {
                    tot += s_cc[ch];
                    ch++;
                    span++;
                }
            }
            s_chunk[0] = c0; s_chunk[1] = span; s_chunk[2] = c0 < 256 ? s_cs[c0] : 0; s_chunk[3] = tot; s_chunk[4] = ch;
        }
        __syncthreads();
        const u32 c0 = s_chunk[0], span = s_chunk[1], start = s_chunk[2], tot = s_chunk[3];
        ch = s_chunk[4];
        if (c0 >= 256) break;
        if (tot == 0) continue;
        const u32 nbins = span * BINS_PER_CHILD;
        const u32 lsh = (u32)((uintptr_t)(a.L + start) & 3u);   // (VALS = false) where the chunk starts inside its first word of L
        for (u32 i = tid; i <= nbins / 2; i += MSDF_NT) s_off[i] = 0;
        __syncthreads();
        u64 key[MSDF_ITEMS];
        u32 val[MSDF_ITEMS], pos[MSDF_ITEMS];
        const u64 *kp = a.keys + start;
        const u32 *vp = a.vals + start;
#pragma unroll
        for (int k = 0; k < MSDF_ITEMS; k++) {
            const u32 i = k * MSDF_NT + tid;
            if (i < tot) {
                key[k] = kp[i];
                if (VALS) val[k] = vp[i];
            }
        }
#pragma unroll
        for (int k = 0; k < MSDF_ITEMS; k++) {
            const u32 i = k * MSDF_NT + tid;
            if (i < tot) {   // (the packed result is taken apart later: no wait between the atomics)
                const u32 bin = bin_of(key[k], c0);
                pos[k] = atomicAdd(&s_off[bin >> 1], (bin & 1) ? 0x10000u : 1u);
            }
        }
        __syncthreads();
        {   // counts -> exclusive offsets; entry nbins = tot
            const u32 w0 = tid * (BPT / 2);
            u32 wv[BPT / 2], sum = 0;
#pragma unroll
            for (int j = 0; j < BPT / 2; j++) {
                wv[j] = (w0 + j) * 2 < nbins ? s_off[w0 + j] : 0u;
                sum += (wv[j] & 0xffffu) + (wv[j] >> 16);
            }
            u32 all;
            u32 ex = block_excl_sum<MSDF_NT>(sum, s_scan, &all);
#pragma unroll
            for (int j = 0; j < BPT / 2; j++) {
                const u32 lo = wv[j] & 0xffffu, hi = wv[j] >> 16;
                if ((w0 + j) * 2 < nbins) s_off[w0 + j] = ex | ((ex + lo) << 16);
                ex += lo + hi;
            }
            if (tid == 0) s_off[nbins / 2] = tot;   // (nbins is even)
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MSDF_ITEMS; k++) {
            const u32 i = k * MSDF_NT + tid;
            if (i < tot) {
                const u32 bin = bin_of(key[k], c0);
                pos[k] = ((pos[k] >> (16 * (bin & 1))) & 0xffffu) + off_at(bin);
                s_low[pos[k]] = (u32)(key[k] >> 8);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MSDF_ITEMS; k++) {
            if (k * MSDF_NT < tot) {   // (block-uniform)
                const u32 i = k * MSDF_NT + tid;
                const bool in = i < tot;
                u32 s = 0, e = 0, lt = 0, eq = 0;
                const u32 mine = (u32)(key[k] >> 8);
                if (in) {
                    const u32 bin = bin_of(key[k], c0);
                    s = off_at(bin);
                    e = off_at(bin + 1);
                }
                for (u32 u = s; __any(u < e); u++) {
                    if (u < e) {
                        const u32 y = s_low[u];
                        lt += y < mine;
                        eq += y == mine;
                    }
                }
                u32 rank = s + lt;
                const bool td = in && eq > 1;   // equal on all key bits: tied beyond the key
                const u64 tb = __ballot(td);
                if (tb) {
                    // (rare) members of an equal run take consecutive places in the order of their bin slots
                    if (td)
                        for (u32 u = s; u < pos[k]; u++) rank += s_low[u] == mine;
                    u32 base = 0;
                    if (l == 0) base = atomicAdd(rctr, (u32)__popcll(tb));
                    base = __shfl(base, 0, 64);
                    if (td) {
                        const u32 o = base + (u32)__popcll(tb & lanemask_lt());
                        if (o < a.rcap) {
                            a.out_slot[rbase + o] = start + rank;
                            a.out_idx[rbase + o] = VALS ? val[k] : mine;
                            if (!VALS) a.out_khi[rbase + o] = (u32)(key[k] >> 40);
                            a.out_grp[rbase + o] = start + s + lt;
                        }
                    }
                }
                if (in) {
                    if (VALS) s_idx[rank] = val[k];
                    s_L[rank + (VALS ? 0u : lsh)] = (u8)(key[k] & 0xff);
                    if (SORTEDKEYS) a.kout[start + rank] = key[k];
                }
            }
        }
        __syncthreads();
        if (VALS) {
            for (u32 i = tid; i < tot; i += MSDF_NT) {
                a.sa_out[start + i] = s_idx[i];
                a.L[start + i] = s_L[i];
            }
        } else {
            // the last column is all that leaves: whole 32-bit words where the chunk covers them, bytes at its two ends
            const u32 sh = lsh, nw = (sh + tot + 3u) >> 2;
            u8 *Lw = a.L + (start - sh);
            const u32 *sw = reinterpret_cast<const u32 *>(s_L);
            for (u32 w = tid; w < nw; w += MSDF_NT) {
                const u32 lo = 4u * w, hi = lo + 4u;
                if (lo >= sh && hi <= sh + tot) {
                    reinterpret_cast<u32 *>(Lw)[w] = sw[w];
                } else {
                    for (u32 q = lo; q < hi; q++)
                        if (q >= sh && q < sh + tot) Lw[q] = s_L[q];
                }
            }
        }
    }
}

// ---- the finish of the key-only levels, small buckets (round 3) -------------------------------------------------
// msd_finish_kernel<256, 8, 4, 1, false, false> spends ~200 instructions per key, most of them in the ranking loop:
// its bins are (child, field 3, top bit of field 4) taken as key BITS, but a field is s symbols in base B inside 8
// bits (5-letter DNA: 125 live values of 256), so only ~210 of a child's 512 bins are ever used and a bin holds 2.6
// keys on average -- the loop over the bin (as long as the longest bin of the wave) runs ~8 times.  Here
//  * a bin is an equal-MASS interval of (field 3, field 4): bin = (A[f3] + ((W[f3] * C[f4]) >> 16)) >> 6 with A / W /
//    C the cumulative / own share of a field value among the text's field-0 digits (the level-1 counts; a text that
//    takes the MSD way looks iid at this depth, and any monotone map is CORRECT -- the ranking inside a bin compares
//    all of bits 39..8).  MSDK_BPC bins per child, ~0.55 keys per bin;
//  * the rank inside the bin: the first four slots of the bin straight-line, a loop only for a longer bin (rare);
//  * 32-bit bins and offsets (one ds_read2 gives a bin's start and end), the chunk walk done by every thread from LDS
//    (no single-thread section), key halves in registers throughout.
// Same results as the generic instance (final places, last column, the tied set by (slot, key, group)).
template <int CH>
__global__ __launch_bounds__(256, 6) void msd_finish_ko_kernel(MsdFinishArgs a, const MsdFinishLut *__restrict__ lut) {
    constexpr u32 NT = 256, ITEMS = 8, TILE = NT * ITEMS, MAXBINS = CH * MSDK_BPC;
    constexpr int BPT = MAXBINS / NT;
    static_assert(CH == 3, "the chunk's live children are c0, l1, l2");
    static_assert(TILE == MSDF_CAP_SMALL && TILE <= 2048 && MAXBINS <= 4096 && BPT * NT == MAXBINS && BPT % 4 == 0,
                  "chunk of the small instance; bin | byte | slot in one word; whole 16-byte groups of bins per thread");
    __shared__ __attribute__((aligned(16))) u32 s_off[MAXBINS + 4];   // [bin] count, then exclusive offset; [MAXBINS] = pairs
    __shared__ u32 s_low[TILE + 4];                                    // key bits 39..8, in bin order
    __shared__ __attribute__((aligned(16))) u8 s_L[TILE + 8];          // preceding bytes, in final order, shifted by start & 3
    __shared__ u32 s_aw[256];                                          // a | w << 16
    __shared__ u16 s_c[256];
    __shared__ u32 s_live[256 + 4];                                    // the parent's non-empty children: child << 16 | min(pairs, 65535)
    __shared__ u32 s_scan[NT / 64 + 1];
    const u32 q = blockIdx.x;
    if (a.pcnt[q] == 0 || (a.counters[1] & 8u)) return;   // (bit 3: the level-3 counts did not add up)
    const u32 tid = threadIdx.x, l = tid & 63;
    u32 nlive;
    {
        const u32 c = a.ccnt[(size_t)q * 256 + tid];
        s_aw[tid] = lut->a[tid] | (lut->w[tid] << 16);
        s_c[tid] = (u16)lut->c[tid];
        // the non-empty children, in order, as one list: a chunk is then three consecutive entries (one LDS round
        // trip instead of a walk over the counts, a dependent read per child)
        const u64 m = __ballot(c != 0);
        if (l == 0) s_scan[tid >> 6] = (u32)__popcll(m);
        __syncthreads();
        u32 before = 0, all = 0;
        for (u32 w = 0; w < NT / 64; w++) {
            const u32 x = s_scan[w];
            before += w < (tid >> 6) ? x : 0u;
            all += x;
        }
        nlive = all;
        if (c) s_live[before + (u32)__popcll(m & lanemask_lt())] = (tid << 16) | (c > 65535u ? 65535u : c);   // (> TILE is all that matters of a long one)
        if (tid < 4) s_live[all + tid] = (256u << 16) | 0xffffu;      // (ends the list: fits no chunk)
    }
    const u32 region = blockIdx.x % FIN_REGIONS;
    u32 *rctr = a.rcount + region * FIN_RSTRIDE;
    const u32 rbase = region * a.rcap;
    const u32 pstart = a.cstart[(size_t)q * 256];    // (the children of a parent lie one behind the other)
    __syncthreads();
    // a chunk: up to CH non-empty children (c0 < l1 < l2; 256: none) of at most TILE pairs together; `ch` walks the
    // children, `run` the pairs before them.  Every thread walks the same counts (LDS broadcasts), the results scalar.
    struct Chunk { u32 c0, l1, l2, tot, off; };
    u32 lj = 0, run = 0;
    auto walk = [&](Chunk &k) -> bool {
        while (true) {
            if (lj >= nlive) return false;
            const u32 e0 = s_live[lj], e1 = s_live[lj + 1], e2 = s_live[lj + 2];
            const u32 n0 = e0 & 0xffffu, n1 = e1 & 0xffffu, n2 = e2 & 0xffffu;
            if (n0 > TILE) {
                // a bucket no chunk can hold (repeats, poly-A): bit 2, the caller takes the LSD way
                if (tid == 0) atomicOr(&a.counters[1], 4u);
                run += a.ccnt[(size_t)q * 256 + (e0 >> 16)];
                lj++;
                continue;
            }
            const bool t1 = n0 + n1 <= TILE, t2 = t1 && n0 + n1 + n2 <= TILE;
            const u32 tot = n0 + (t1 ? n1 : 0u) + (t2 ? n2 : 0u);
            k.c0 = (u32)__builtin_amdgcn_readfirstlane((int)(e0 >> 16));
            k.l1 = (u32)__builtin_amdgcn_readfirstlane((int)(t1 ? e1 >> 16 : 256u));
            k.l2 = (u32)__builtin_amdgcn_readfirstlane((int)(t2 ? e2 >> 16 : 256u));
            k.tot = (u32)__builtin_amdgcn_readfirstlane((int)tot);
            k.off = (u32)__builtin_amdgcn_readfirstlane((int)run);
            run += tot;
            lj += 1u + (u32)t1 + (u32)t2;
            return true;
        }
    };
    // the keys of the NEXT chunk are loaded while this one is ranked and copied out (the barriers in between are
    // LDS-only, so nothing waits for them before their first use); a slot past the chunk re-reads its first key.
    // Plain loads, not asm-issued ones: the register allocator is free to MOVE the destination of an asm load
    // before its data has arrived (it did, in one build of this kernel: stale register contents became keys).
    u64 nk[ITEMS];
    auto prefetch = [&](const Chunk &k) {
        const u64 *kp = a.keys + pstart + k.off;
#pragma unroll
        for (int j = 0; j < (int)ITEMS; j++) {
            const u32 i = j * NT + tid;
            nk[j] = kp[i < k.tot ? i : 0u];
        }
    };
    Chunk cur, nxt;
    bool have = walk(cur);
    if (have) prefetch(cur);
#ifdef MSDK_PROFILE
    u32 tq[10], acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define MSDK_T(i) tq[i] = (u32)__builtin_readcyclecounter()
#else
#define MSDK_T(i)
#endif
    while (have) {
        MSDK_T(0);
        const u32 c0 = cur.c0, l1 = cur.l1, l2 = cur.l2, tot = cur.tot, start = pstart + cur.off;
        const u32 lsh = (u32)((uintptr_t)(a.L + start) & 3u);   // where the chunk starts inside its first word of L
        u32 klo[ITEMS], khi[ITEMS];
#pragma unroll
        for (int k = 0; k < (int)ITEMS; k++) {
            klo[k] = (u32)nk[k];
            khi[k] = (u32)(nk[k] >> 32);
        }
        MSDK_T(1);
        lds_barrier();   // (the previous chunk's copy-out has read the LDS images)
        MSDK_T(2);
        {
            uint4 *z = reinterpret_cast<uint4 *>(s_off + tid * BPT);
#pragma unroll
            for (int j = 0; j < BPT / 4; j++) z[j] = make_uint4(0u, 0u, 0u, 0u);
        }
        lds_barrier();
        MSDK_T(3);
        u32 bp[ITEMS];   // bin | preceding byte << 12 | slot << 20 (the slot: first the arrival number in the bin, then the place in bin order)
#pragma unroll
        for (int k = 0; k < (int)ITEMS; k++) {
            const u32 i = k * NT + tid;
            if (i < tot) {
                const u32 aw = s_aw[khi[k] & 255u];                    // field 3: key bits 39..32
                const u32 c = s_c[klo[k] >> 24];                       // field 4: 31..24
                // (HIP declares __umul24 as returning int: without the cast the shift is arithmetic, and a share above
                // one half times a cumulative share above one half came back negative -- the bin left the table)
                u32 inner = ((aw & 0xffffu) + ((u32)__umul24(aw >> 16, c) >> 16)) >> 6;   // < MSDK_BPC: a + w <= 65535
                inner = inner < MSDK_BPC ? inner : MSDK_BPC - 1u;                          // (whatever the table holds)
                const u32 child = (khi[k] >> 8) & 255u;
                const u32 bin = ((u32)(child >= l1) + (u32)(child >= l2)) * MSDK_BPC + inner;
                bp[k] = bin | ((klo[k] & 255u) << 12) | (atomicAdd(&s_off[bin], 1u) << 20);
            }
        }
        __syncthreads();
        MSDK_T(4);
        {   // counts -> exclusive offsets; entry MAXBINS = pairs
            uint4 *z = reinterpret_cast<uint4 *>(s_off + tid * BPT);
            uint4 v[BPT / 4];
            u32 sum = 0;
#pragma unroll
            for (int j = 0; j < BPT / 4; j++) {
                v[j] = z[j];
                sum += v[j].x + v[j].y + v[j].z + v[j].w;
            }
            u32 all;
            u32 ex = block_excl_sum<NT>(sum, s_scan, &all);
#pragma unroll
            for (int j = 0; j < BPT / 4; j++) {
                uint4 o;
                o.x = ex; ex += v[j].x;
                o.y = ex; ex += v[j].y;
                o.z = ex; ex += v[j].z;
                o.w = ex; ex += v[j].w;
                z[j] = o;
            }
            if (tid == NT - 1) s_off[MAXBINS] = ex;
        }
        __syncthreads();
        MSDK_T(5);
#pragma unroll
        for (int k = 0; k < (int)ITEMS; k++) {
            const u32 i = k * NT + tid;
            if (i < tot) {
                bp[k] += s_off[bp[k] & 0xfffu] << 20;
                s_low[bp[k] >> 20] = __builtin_amdgcn_alignbit(khi[k], klo[k], 8);
            }
        }
        __syncthreads();   // (the key registers are dead from here: a tied member's upper bits are the parent and its child)
        MSDK_T(6);
        have = walk(nxt);
        if (have) prefetch(nxt);
#pragma unroll
        for (int k = 0; k < (int)ITEMS; k++) {
            if (k * NT < tot) {   // (block-uniform)
                const u32 i = k * NT + tid;
                const bool in = i < tot;
                u32 s = 0, e = 0, mine = 0;
                if (in) {
                    s = s_off[bp[k] & 0xfffu];
                    e = s_off[(bp[k] & 0xfffu) + 1];
                }
                const u32 cnt = e - s;
                u32 lt = 0, eq = cnt ? 1u : 0u;
                // (the kernel is bound by its LDS traffic: a key alone in its bin -- six in ten -- reads nothing more, the
                // others their own bits and the bin's first two slots, one in ten slots 2 and 3 as well)
                if (cnt > 1) {
                    mine = s_low[bp[k] >> 20];
                    const u32 y0 = s_low[s], y1 = s_low[s + 1];
                    u32 y2 = 0, y3 = 0;
                    if (cnt > 2) {
                        y2 = s_low[s + 2];
                        y3 = s_low[s + 3];
                    }
                    lt = (u32)(y0 < mine) + (u32)(y1 < mine) + (u32)(cnt > 2 && y2 < mine) + (u32)(cnt > 3 && y3 < mine);
                    eq = (u32)(y0 == mine) + (u32)(y1 == mine) + (u32)(cnt > 2 && y2 == mine) + (u32)(cnt > 3 && y3 == mine);
                }
                if (__any(cnt > 4)) {
                    for (u32 u = s + 4; __any(u < e); u++) {
                        if (u < e) {
                            const u32 y = s_low[u];
                            lt += y < mine;
                            eq += y == mine;
                        }
                    }
                }
                u32 rank = s + lt;
                const bool td = in && eq > 1;   // equal on all key bits: tied beyond the key
                const u64 tb = __ballot(td);
                if (tb) {
                    // (rare) members of an equal run take consecutive places in the order of their bin slots
                    if (td)
                        for (u32 u = s; u < (bp[k] >> 20); u++) rank += s_low[u] == mine;
                    u32 base = 0;
                    if (l == 0) base = atomicAdd(rctr, (u32)__popcll(tb));
                    base = __shfl(base, 0, 64);
                    if (td) {
                        const u32 o = base + (u32)__popcll(tb & lanemask_lt());
                        if (o < a.rcap) {
                            const u32 nb = (bp[k] & 0xfffu) / MSDK_BPC;   // which of the chunk's children
                            a.out_slot[rbase + o] = start + rank;
                            a.out_idx[rbase + o] = mine;
                            a.out_khi[rbase + o] = (q << 8) | (nb == 0 ? c0 : (nb == 1 ? l1 : l2));   // key bits 63..40
                            a.out_grp[rbase + o] = start + s + lt;
                        }
                    }
                }
                if (in) s_L[rank + lsh] = (u8)((bp[k] >> 12) & 0xffu);
            }
        }
        lds_barrier();
        MSDK_T(7);
        {
            // the last column is all that leaves: whole 32-bit words where the chunk covers them, bytes at its two ends
            const u32 nw = (lsh + tot + 3u) >> 2;
            u8 *Lw = a.L + (start - lsh);
            const u32 *sw = reinterpret_cast<const u32 *>(s_L);
            for (u32 w = tid; w < nw; w += NT) {
                const u32 lo = 4u * w, hi = lo + 4u;
                if (lo >= lsh && hi <= lsh + tot) {
                    reinterpret_cast<u32 *>(Lw)[w] = sw[w];
                } else {
                    for (u32 x = lo; x < hi; x++)
                        if (x >= lsh && x < lsh + tot) Lw[x] = s_L[x];
                }
            }
        }
#ifdef MSDK_PROFILE
        MSDK_T(8);
        for (int i = 0; i < 8; i++) acc[i] += tq[i + 1] - tq[i];
        acc[8] += 1; acc[9] += tot;
#endif
        cur = nxt;
    }
#ifdef MSDK_PROFILE
    if (tid == 0 && (q & 255u) == 43u) {
        u64 *dbg = reinterpret_cast<u64 *>(a.counters) - 12 + 112;   // (counters = d_scalars + 12 words of 64 bits)
        for (int i = 0; i < 10; i++) atomicAdd((unsigned long long *)&dbg[i], (unsigned long long)acc[i]);
    }
#endif
}

#endif  // __HIPCC__
