// tc_msd.hpp -- round 0 of the suffix sort for small alphabets as an MSD radix sort.
//
// Replaces (for long texts over <= 15 byte values: DNA records) the LSD passes of tc_radix.hpp +
// finish_kernel in front of `DS.unstableSortOn snd` (reference BWT/Internal.hs:130).
//
// Why MSD here.  An LSD pass must be STABLE, which on a GPU means: ranks by wave-ordered match,
// tiles in ticket order, a decoupled look-back per digit, and scattered segments of ~33 pairs that
// start at arbitrary alignment (every seam a partially written 128-byte line; measured in round 1:
// 7.8 ms for the access pattern alone against 5.0 ms for the same bytes in whole lines).  An MSD
// pass only has to PARTITION: the order inside a bucket is irrelevant, because the bucket is sorted
// again by the next field.  That removes the look-back, the ticket and the stable ranking, and it
// allows software write combining: a workgroup keeps, per digit, the pairs that do not yet fill a
// 32-pair group (256 B of keys + 128 B of values) in LDS and only ever stores whole, line-aligned
// groups.  Positions are exact (no atomics, no holes): a counting kernel with the SAME static
// work split gives every (workgroup, parent bucket) segment its private range in every child.
//
//   level 1   text -> (key, idx) partitioned by field 0           1 B read, 12 B written per suffix
//   level 2,3 partitioned inside each parent by field 1, 2        12 B read, 12 B written
//   finish    each level-3 bucket (9 symbols on DNA, ~550 suffixes at 1 GiB) is ordered by the
//             remaining 32 key bits inside one wave's LDS image (counting sort by field 3, then
//             ranks inside the ~4-member bins); SA, last column and the tied set leave from there
//             (same contract as finish_kernel)                    12 B read, 5 B written
// A text whose level-3 buckets exceed MSDF_CAP (repeats, runs) takes the LSD path instead.
#pragma once
#include "tc_sa.hpp"

#define MSD_NT 1024
#define MSD_ITEMS 4
#define MSD_TILE (MSD_NT * MSD_ITEMS)
#define MSD_GROUP 32          // pairs per store group: 2 lines of keys, 1 line of values
#define MSD_LEVELS 3
#define MSDF_NT 256
#define MSDF_CAP 1024         // largest level-3 bucket the finish kernel orders
#define MSDF_ITEMS (MSDF_CAP / 64)

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

// tiles per parent -> exclusive prefix.  One block; nparents <= 65536.
__global__ __launch_bounds__(1024) void msd_prep_kernel(const u32 *__restrict__ pcnt, u32 nparents,
                                                        u32 *__restrict__ tpre) {
    __shared__ u32 s_scan[1024 / 64 + 1];
    const u32 per = (nparents + 1023) / 1024;
    const u32 lo = threadIdx.x * per;
    u32 sum = 0;
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
    u32 lo = 0, hi = nparents;   // largest q with tpre[q] <= t  (then skip empties forward)
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (tpre[mid] <= t) lo = mid; else hi = mid;
    }
    return lo;   // tpre[lo] <= t < tpre[lo + 1] because equal prefixes belong to empty parents before lo
}

struct MsdTileInfo {
    u32 base;     // first position of the tile in the input arrays
    u32 valid;    // pairs in the tile
    u32 q;        // parent
    u32 last;     // 1: last tile of its segment (parent changes or range ends)
};
// info of tile t (parent hint q, advanced as needed); returns the parent found
__device__ __forceinline__ u32 msd_tile_info(const MsdLevel &L, u32 t, u32 t_end, u32 q, MsdTileInfo *out) {
    while (L.tpre[q + 1] <= t) q++;
    const u32 rel = t - L.tpre[q];
    const u32 cnt = L.pcnt[q];
    const u32 off = rel * MSD_TILE;
    out->base = L.pstart[q] + off;
    out->valid = cnt - off < MSD_TILE ? cnt - off : MSD_TILE;
    out->q = q;
    out->last = (t + 1 >= t_end || L.tpre[q + 1] <= t + 1) ? 1u : 0u;
    return q;
}

// field 0 of suffix i straight from the text (level 1 only): the first s symbols in base B
struct MsdTextDigit {
    const u8 *text;
    u32 n;
    u32 B, s;
    u16 lut[256];
};

// ---- per-segment digit counts ------------------------------------------------------------------
// keys: digit = (key >> shift) & 255.  TEXT (level 1): digit = field 0 of suffix i = its first s
// symbols in base B, straight from the text (two overlapping word loads per 4 suffixes where the
// tile lies inside the text; s <= 5 there, else bytes).
template <bool TEXT>
__global__ __launch_bounds__(MSD_NT) void msd_count_kernel(MsdLevel L, const u64 *__restrict__ keys,
                                                           MsdTextDigit td) {
    __shared__ u32 s_cnt[256];
    __shared__ u16 s_lut[TEXT ? 256 : 1];
    __shared__ MsdTileInfo s_info;
    const u32 tid = threadIdx.x, b = blockIdx.x, G = gridDim.x;
    if (tid < 256) s_cnt[tid] = 0;
    if (TEXT && tid < 256) s_lut[tid] = td.lut[tid];
    const u32 T = L.tpre[L.nparents];
    const u32 t0 = msd_tile_lo(T, b, G), t1 = msd_tile_lo(T, b + 1, G);
    if (t0 >= t1) return;
    u32 q = 0;
    if (tid == 0) q = msd_find_parent(L.tpre, L.nparents, t0);
    for (u32 t = t0; t < t1; t++) {
        __syncthreads();
        if (tid == 0) q = msd_tile_info(L, t, t1, q, &s_info);
        __syncthreads();
        const MsdTileInfo ti = s_info;
        if (TEXT) {
            const bool words = td.s <= 5 && ti.valid == MSD_TILE && (u64)ti.base + MSD_TILE + 8 <= td.n &&
                               ((((uintptr_t)td.text) + ti.base) & 3) == 0;
            if (words) {
                const u32 *tw = reinterpret_cast<const u32 *>(td.text + ti.base);
                const u64 x = (u64)tw[tid] | ((u64)tw[tid + 1] << 32);   // bytes 4 tid .. 4 tid + 7
                u32 cd[8];
#pragma unroll
                for (int j = 0; j < 8; j++) cd[j] = s_lut[(u32)(x >> (8 * j)) & 255u];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    u32 g = 0;
#pragma unroll
                    for (int j = 0; j < 5; j++)
                        if (j < (int)td.s) g = g * td.B + cd[k + j];
                    atomicAdd(&s_cnt[g], 1u);
                }
            } else {
#pragma unroll
                for (int k = 0; k < MSD_ITEMS; k++) {
                    const u32 p = k * MSD_NT + tid;
                    if (p < ti.valid) {
                        u64 i = (u64)ti.base + p;
                        u32 g = 0;
                        for (u32 j = 0; j < td.s; j++, i++) g = g * td.B + (i < td.n ? (u32)s_lut[td.text[i]] : 0u);
                        atomicAdd(&s_cnt[g], 1u);
                    }
                }
            }
        } else {
            const u64 *kt = keys + ti.base;
#pragma unroll
            for (int k = 0; k < MSD_ITEMS; k++) {
                const u32 p = k * MSD_NT + tid;
                if (p < ti.valid) atomicAdd(&s_cnt[(u32)(kt[p] >> L.shift) & 255u], 1u);
            }
        }
        if (ti.last) {
            __syncthreads();
            if (tid < 256) {
                L.seg[((size_t)ti.q + b) * 256 + tid] = s_cnt[tid];
                s_cnt[tid] = 0;
            }
        }
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
    const u32 T = L.tpre[L.nparents];
    const u32 bf = msd_block_of_tile(T, L.tpre[q], G), bl = msd_block_of_tile(T, L.tpre[q + 1] - 1, G);
    u32 tot = 0;
    for (u32 b = bf; b <= bl; b++) tot += L.seg[((size_t)q + b) * 256 + d];
    u32 all;
    const u32 excl = block_excl_sum<256>(tot, s_scan, &all);
    u32 run = L.pstart[q] + excl;
    L.cstart[c] = run;
    L.ccnt[c] = tot;
    for (u32 b = bf; b <= bl; b++) {
        const size_t o = ((size_t)q + b) * 256 + d;
        const u32 v = L.seg[o];
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
        if ((d & 63) == 0 && m > MSDF_CAP) atomicMax(maxchild, m);
    }
}

// ---- one partition pass -------------------------------------------------------------------------
// KEYGEN: level 1, keys are built from the text (layout of keybuild_kernel in tc_sa.hpp: P fields
// of s symbols in base B, w = 8 bits each, from bit 63 down; low byte = the preceding text byte).
template <bool KEYGEN>
__global__ __launch_bounds__(MSD_NT) void msd_partition_kernel(MsdLevel L, const u64 *__restrict__ kin,
                                                               const u32 *__restrict__ vin,
                                                               u64 *__restrict__ kout, u32 *__restrict__ vout,
                                                               const u8 *__restrict__ text, RadixKeyGen kg) {
    // staging of the tile sorted by digit; the key-generation image overlays it
    __shared__ __attribute__((aligned(16))) u64 s_keys[MSD_TILE];
    __shared__ __attribute__((aligned(16))) u32 s_vals[MSD_TILE];
    // pairs that do not fill a group yet, per digit (slots [0, r); the first ph slots of a
    // segment's first group are phantoms standing for the positions before the segment's range)
    __shared__ __attribute__((aligned(16))) u64 c_keys[256 * MSD_GROUP];
    __shared__ __attribute__((aligned(16))) u32 c_vals[256 * MSD_GROUP];
    __shared__ u32 s_cnt[256], s_dstart[256], s_r[256], s_ng[256], s_goff[256], s_cur[256], s_ph[256];
    __shared__ u16 s_gmap[512];
    __shared__ u16 s_jmap[256];
    __shared__ u32 s_scan[8];
    __shared__ u32 s_tot[2];
    __shared__ u16 s_klut[KEYGEN ? 256 : 1];
    __shared__ MsdTileInfo s_info[2];

    const u32 tid = threadIdx.x, b = blockIdx.x, G = gridDim.x;
    const u32 hw = tid >> 5, l5 = tid & 31;
    if (KEYGEN && tid < 256) s_klut[tid] = kg.lut[tid];
    if (tid < 256) { s_cnt[tid] = 0; s_r[tid] = 0; s_ph[tid] = 0; s_cur[tid] = 0; }
    const u32 T = L.tpre[L.nparents];
    const u32 t0 = msd_tile_lo(T, b, G), t1 = msd_tile_lo(T, b + 1, G);
    if (t0 >= t1) return;
    u32 q = 0;
    if (tid == 0) {
        q = msd_find_parent(L.tpre, L.nparents, t0);
        q = msd_tile_info(L, t0, t1, q, &s_info[0]);
    }
    __syncthreads();

    // key-generation image (overlays the staging area)
    constexpr int KG_PRE = 16, KG_SLOTS = MSD_TILE + KG_PRE + 80;
    u16 *k_c = reinterpret_cast<u16 *>(s_keys);
    u16 *k_g = k_c + KG_SLOTS;
    u8 *k_r = reinterpret_cast<u8 *>(k_g + KG_SLOTS);
    static_assert(KG_SLOTS * 5 <= MSD_TILE * 8, "key-generation image must fit under the staging keys");
    const u32 kg_units = KEYGEN ? (KG_PRE + MSD_TILE + kg.P * kg.s + kg.s + 15) / 16 : 0;

    u64 key[MSD_ITEMS], nkey[MSD_ITEMS];
    u32 val[MSD_ITEMS], nval[MSD_ITEMS];
    uint4 raw = make_uint4(0, 0, 0, 0), nraw = make_uint4(0, 0, 0, 0);
    auto load_tile = [&](const MsdTileInfo &ti, u64 *kk, u32 *vv, uint4 &rw) {
        if (KEYGEN) {
            // 16 text bytes per thread from position base - KG_PRE + 16 * tid (bytes outside the text: 0)
            if (tid < kg_units) {
                const i64 p0 = (i64)ti.base - KG_PRE + (i64)tid * 16;
                const i64 nt = (i64)kg.n_text;
                if (p0 >= 0 && p0 + 16 <= nt && ((((uintptr_t)text) + (u64)p0) & 15) == 0) {
                    rw = *reinterpret_cast<const uint4 *>(text + p0);
                } else {
                    u32 x[4] = {0, 0, 0, 0};
                    for (int j = 0; j < 16; j++) {
                        const i64 pp = p0 + j;
                        const u32 c = (pp >= 0 && pp < nt) ? (u32)text[pp] : 0u;
                        x[j >> 2] |= c << (8 * (j & 3));
                    }
                    rw = make_uint4(x[0], x[1], x[2], x[3]);
                }
            }
        } else {
            const u64 *kt = kin + ti.base;
            const u32 *vt = vin + ti.base;
#pragma unroll
            for (int k = 0; k < MSD_ITEMS; k++) {
                const u32 p = k * MSD_NT + tid;
                if (p < ti.valid) {
                    kk[k] = kt[p];
                    vv[k] = vt[p];
                }
            }
        }
    };
    auto seg_init = [&](u32 qq) {   // threads < 256
        const u32 sb = L.seg[((size_t)qq + b) * 256 + tid];
        s_cur[tid] = sb & ~(u32)(MSD_GROUP - 1);
        s_ph[tid] = sb & (MSD_GROUP - 1);
        s_r[tid] = sb & (MSD_GROUP - 1);
    };
    if (tid < 256) seg_init(s_info[0].q);
    load_tile(s_info[0], key, val, raw);

    int cur = 0;
    for (u32 t = t0; t < t1; t++, cur ^= 1) {
        const MsdTileInfo ti = s_info[cur];
        const bool more = t + 1 < t1;
        if (tid == 0 && more) q = msd_tile_info(L, t + 1, t1, q, &s_info[cur ^ 1]);
        __syncthreads();   // (B0) next tile's info visible; s_cnt zeroed; staging free
        if (more) load_tile(s_info[cur ^ 1], nkey, nval, nraw);   // in flight while this tile is ranked
        if (KEYGEN) {
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
                *reinterpret_cast<uint4 *>(k_r + tid * 16) = raw;
            }
            __syncthreads();
            const u32 gslots = KG_PRE + MSD_TILE + kg.P * kg.s;
            const u32 B = kg.B;
            for (u32 x = tid; x < gslots; x += MSD_NT) {
                u32 g = 0;
                for (u32 j = 0; j < kg.s; j++) g = g * B + k_c[x + j];
                k_g[x] = (u16)g;
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
                    val[k] = ti.base + p;
                }
            }
            __syncthreads();   // image dead: the staging area may be written
        }
        // (S1) digit + rank inside the digit (any order: the partition need not be stable)
        u32 dig[MSD_ITEMS], rnk[MSD_ITEMS];
#pragma unroll
        for (int k = 0; k < MSD_ITEMS; k++) {
            const u32 p = k * MSD_NT + tid;
            dig[k] = (u32)(key[k] >> L.shift) & 255u;
            if (p < ti.valid) rnk[k] = atomicAdd(&s_cnt[dig[k]], 1u);
        }
        __syncthreads();   // (B1)
        // (S2) per digit: start in the staging area, whole groups to store, carry jobs
        if (tid < 256) {
            const u32 c = s_cnt[tid], r = s_r[tid];
            const u32 ng = (r + c) >> 5;
            const u32 packed = c | (ng << 13) | ((c ? 1u : 0u) << 23);
            const u32 inc = wave_incl_sum(packed);
            if ((tid & 63) == 63) s_scan[tid >> 6] = inc;
            s_dstart[tid] = inc - packed;   // wave-relative for now
            s_ng[tid] = ng;
        }
        __syncthreads();   // (B2a)
        if (tid < 256) {
            u32 base = 0;
            for (u32 i = 0; i < (tid >> 6); i++) base += s_scan[i];
            const u32 ex = s_dstart[tid] + base;
            const u32 ds = ex & 0x1fffu, go = (ex >> 13) & 0x3ffu, jo = ex >> 23;
            s_dstart[tid] = ds;
            s_goff[tid] = go;
            const u32 ng = s_ng[tid];
            for (u32 k = 0; k < ng; k++) s_gmap[go + k] = (u16)tid;
            if (s_cnt[tid]) s_jmap[jo] = (u16)tid;
            if (tid == 255) {
                s_tot[0] = go + ng;
                s_tot[1] = jo + (s_cnt[tid] ? 1u : 0u);
            }
        }
        __syncthreads();   // (B2)
        // (S3) the tile, sorted by digit, into the staging area
#pragma unroll
        for (int k = 0; k < MSD_ITEMS; k++) {
            const u32 p = k * MSD_NT + tid;
            if (p < ti.valid) {
                const u32 o = s_dstart[dig[k]] + rnk[k];
                s_keys[o] = key[k];
                s_vals[o] = val[k];
            }
        }
        __syncthreads();   // (B3)
        // (S4) whole groups leave: group k of digit d = elements [32k, 32k + 32) of (carry ++ segment)
        {
            const u32 Gt = s_tot[0];
            for (u32 g = hw; g < Gt; g += MSD_NT / 32) {
                const u32 d = s_gmap[g];
                const u32 k = g - s_goff[d], r = s_r[d];
                const u32 e = k * MSD_GROUP + l5;
                u64 kk;
                u32 vv;
                if (e < r) {
                    kk = c_keys[d * MSD_GROUP + e];
                    vv = c_vals[d * MSD_GROUP + e];
                } else {
                    const u32 o = s_dstart[d] + e - r;
                    kk = s_keys[o];
                    vv = s_vals[o];
                }
                if (k > 0 || l5 >= s_ph[d]) {
                    const u32 pos = s_cur[d] + e;
                    kout[pos] = kk;
                    vout[pos] = vv;
                }
            }
        }
        __syncthreads();   // (B4) carry read before it is rewritten
        // (S5) what is left of every touched digit becomes its carry
        {
            const u32 nj = s_tot[1];
            for (u32 j = hw; j < nj; j += MSD_NT / 32) {
                const u32 d = s_jmap[j];
                const u32 c = s_cnt[d], r = s_r[d], ng = s_ng[d], ds = s_dstart[d];
                const u32 newr = ng ? ((r + c) & (MSD_GROUP - 1)) : r + c;
                if (ng) {
                    if (l5 < newr) {
                        const u32 o = ds + (ng * MSD_GROUP - r) + l5;
                        c_keys[d * MSD_GROUP + l5] = s_keys[o];
                        c_vals[d * MSD_GROUP + l5] = s_vals[o];
                    }
                } else if (l5 >= r && l5 < newr) {
                    const u32 o = ds + l5 - r;
                    c_keys[d * MSD_GROUP + l5] = s_keys[o];
                    c_vals[d * MSD_GROUP + l5] = s_vals[o];
                }
                if (l5 == 0) {
                    s_r[d] = newr;
                    if (ng) {
                        s_cur[d] += ng * MSD_GROUP;
                        s_ph[d] = 0;
                    }
                    s_cnt[d] = 0;
                }
            }
        }
        if (ti.last) {
            // end of the segment: the carries go to their exact places (partial lines, once per
            // segment and digit); then the next segment's ranges are taken
            __syncthreads();
            for (u32 x = tid; x < 256 * MSD_GROUP; x += MSD_NT) {
                const u32 d = x >> 5, j = x & 31;
                if (j >= s_ph[d] && j < s_r[d]) {
                    const u32 pos = s_cur[d] + j;
                    kout[pos] = c_keys[x];
                    vout[pos] = c_vals[x];
                }
            }
            __syncthreads();
            if (more && tid < 256) seg_init(s_info[cur ^ 1].q);
        }
        if (more) {
#pragma unroll
            for (int k = 0; k < MSD_ITEMS; k++) {
                key[k] = nkey[k];
                val[k] = nval[k];
            }
            raw = nraw;
        }
    }
}

// ---- finish: every level-3 bucket ordered by its remaining key bits ----------------------------------
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
    u32 *counters;       // [1] bit 2: a bucket above MSDF_CAP was met (the caller takes the LSD path)
};

// One wave per bucket: counting sort by field 3 (bits 39..32) into the wave's LDS image, ranks inside
// the bins by all 32 remaining bits (39..8), then SA / last column leave coalesced.  Members with equal
// remaining bits are tied beyond the key: (slot, suffix, group = first slot of the equal run) go to
// the tied list exactly as finish_kernel emits them.
__global__ __launch_bounds__(MSDF_NT) void msd_finish_kernel(MsdFinishArgs a) {
    constexpr int NW = MSDF_NT / 64;
    __shared__ u32 s_off[NW][256];
    __shared__ u32 s_low[NW][MSDF_CAP];   // remaining key bits, in bin order
    __shared__ u32 s_idx[NW][MSDF_CAP];   // suffix starts, in final order
    __shared__ u8 s_L[NW][MSDF_CAP];      // preceding bytes, in final order
    const u32 q = blockIdx.x;
    if (a.pcnt[q] == 0) return;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    u32 *off = s_off[w], *low = s_low[w], *sidx = s_idx[w];
    u8 *sl = s_L[w];
    const u32 region = blockIdx.x % FIN_REGIONS;
    u32 *rctr = a.rcount + region * FIN_RSTRIDE;
    const u32 rbase = region * a.rcap;
    for (u32 ch = w; ch < 256; ch += NW) {
        const size_t cid = (size_t)q * 256 + ch;
        const u32 c = a.ccnt[cid];
        if (c == 0) continue;
        if (c > MSDF_CAP) {
            if (l == 0) atomicOr(&a.counters[1], 4u);
            continue;
        }
        const u32 bs = a.cstart[cid];
        const u64 *kp = a.keys + bs;
        const u32 *vp = a.vals + bs;
        u64 key[MSDF_ITEMS];
        u32 val[MSDF_ITEMS], pos[MSDF_ITEMS];
#pragma unroll
        for (int k = 0; k < MSDF_ITEMS; k++) {
            const u32 i = k * 64 + l;
            if (k * 64 < (int)c && i < c) {
                key[k] = kp[i];
                val[k] = vp[i];
            }
        }
        off[l] = 0; off[64 + l] = 0; off[128 + l] = 0; off[192 + l] = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < MSDF_ITEMS; k++) {
            const u32 i = k * 64 + l;
            if (k * 64 < (int)c && i < c) pos[k] = atomicAdd(&off[(u32)(key[k] >> 32) & 255u], 1u);
        }
        __builtin_amdgcn_wave_barrier();
        {   // counts -> exclusive offsets (4 bins per lane); the end of bin 255 is c
            const u32 h0 = off[4 * l], h1 = off[4 * l + 1], h2 = off[4 * l + 2], h3 = off[4 * l + 3];
            const u32 sm = h0 + h1 + h2 + h3;
            const u32 ex = wave_incl_sum(sm) - sm;
            __builtin_amdgcn_wave_barrier();
            off[4 * l] = ex; off[4 * l + 1] = ex + h0; off[4 * l + 2] = ex + h0 + h1; off[4 * l + 3] = ex + h0 + h1 + h2;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < MSDF_ITEMS; k++) {
            const u32 i = k * 64 + l;
            if (k * 64 < (int)c && i < c) {
                pos[k] += off[(u32)(key[k] >> 32) & 255u];
                low[pos[k]] = (u32)(key[k] >> 8);
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int k = 0; k < MSDF_ITEMS; k++) {
            if (k * 64 < (int)c) {   // (wave-uniform)
                const u32 i = k * 64 + l;
                const bool in = i < c;
                u32 s = 0, e = 0, lt = 0, eqb = 0, eq = 0;
                const u32 mine = (u32)(key[k] >> 8);
                if (in) {
                    const u32 f = (u32)(key[k] >> 32) & 255u;
                    s = off[f];
                    e = f == 255u ? c : off[f + 1];
                }
                for (u32 u = s; __any(u < e); u++) {
                    if (u < e) {
                        const u32 y = low[u];
                        lt += y < mine;
                        eq += y == mine;
                        eqb += (y == mine) & (u < pos[k]);
                    }
                }
                const u32 rank = s + lt + eqb;
                if (in) {
                    sidx[rank] = val[k];
                    sl[rank] = (u8)(key[k] & 0xff);
                }
                const bool td = in && eq > 1;   // equal on all key bits: tied beyond the key
                const u64 tb = __ballot(td);
                if (tb) {
                    u32 base = 0;
                    if (l == 0) base = atomicAdd(rctr, (u32)__popcll(tb));
                    base = __shfl(base, 0, 64);
                    if (td) {
                        const u32 o = base + (u32)__popcll(tb & lanemask_lt());
                        if (o < a.rcap) {
                            a.out_slot[rbase + o] = bs + rank;
                            a.out_idx[rbase + o] = val[k];
                            a.out_grp[rbase + o] = bs + s + lt;
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (u32 i = l; i < c; i += 64) {
            a.sa_out[bs + i] = sidx[i];
            a.L[bs + i] = sl[i];
        }
        __builtin_amdgcn_wave_barrier();
    }
}

#endif  // __HIPCC__
