// tc_decode_host.hpp -- decode path: inverse RLE, inverse MTF, inverse BWT.
//
// Inverse BWT replaces fromBWT (reference BWT.hs:93-104: sort (symbol, position))
// + magicInverseBWT (BWT/Internal.hs:163-200: an n-step dependent pointer chase
//   f <- snd (index ys f) from the Nothing row until it returns to row e).
// On the device:
//   1. stable counting sort of positions by symbol (Nothing first) = one/two radix
//      passes -> spos[d] (the `snd` column of the sorted Seq); the `fst` column is
//      implied by the symbol boundaries C[].
//   2. the chase becomes list ranking: one row per block of S rows is a splitter (its offset in
//      the block is a byte-sum hash of the block number, so that periodic inputs -- whose walks
//      keep a constant row mod S -- cannot miss every splitter);
//      each splitter walks to the next splitter (independent walks, in parallel),
//      the splitter chain is ranked by pointer jumping (log K rounds), then every
//      splitter on row e's chain re-walks its segment writing text bytes at their
//      final offsets.  Works for ANY Seq (Maybe Word8): no Nothing => empty output;
//      a second Nothing met on the chain => TC_ERR_MALFORMED (fromJust, :195).
#pragma once
#include <type_traits>
#include <algorithm>
#include "tc_encode_host.hpp"

#ifndef IBWT_S
#define IBWT_S 256  // splitter spacing (rows), a power of two
#endif
#define IBWT_SBITS (IBWT_S == 256 ? 8 : IBWT_S == 512 ? 9 : IBWT_S == 1024 ? 10 : 7)

#ifdef __HIPCC__

// the splitter of block q = r / 256 sits at a HASHED offset (the top bits of q times an odd constant); block 0: row 0.
// (Rounds 1-3: -(byte sum of q) mod 256 -- enough against walks that keep their row mod 256 (periodic text: 211 s for
// 256 MiB before it), but an arithmetic progression in q itself: a walk that advances by a constant 17 rows per step -- the
// rows of N^k x, k = 1, 2, .., of an assembly with 17 gaps lie 17 apart -- shifts its residues by -1 per block exactly as the
// offsets did and missed the splitters for tens of thousands of steps: the 2^28-byte record's walk kernel 38 ms instead of 5.)
__device__ __forceinline__ u32 ibwt_split_off(u32 q) {
    return (q * 0x9E3779B1u) >> (32 - IBWT_SBITS);
}
__device__ __forceinline__ u32 ibwt_split_row(u32 q) { return q * IBWT_S + ibwt_split_off(q); }
__device__ __forceinline__ bool ibwt_is_splitter(u32 r) {
    return (r & (u32)(IBWT_S - 1)) == ibwt_split_off(r >> IBWT_SBITS);
}

template <class Acc>
__global__ __launch_bounds__(256) void ibwt_keys_kernel(Acc acc, u32 N, Lut16 lut,
                                                        u64 *__restrict__ keys) {
    u32 j = blockIdx.x * 256 + threadIdx.x;
    if (j < N) keys[j] = (u64)lut.v[acc(j) + 1];
}

struct CTable {
    u32 c[260];  // c[code] = first sorted row of that code; c[sigma] = N
    i16 sym[260];
    u32 sigma;
};

__device__ __forceinline__ int ibwt_sym_of_row(const u32 *s_c, const i16 *s_sym, u32 sigma, u32 r) {
    // largest code with c[code] <= r
    u32 lo = 0, hi = sigma;  // invariant: c[lo] <= r < c[hi]
    while (hi - lo > 1) {
        u32 mid = (lo + hi) >> 1;
        if (s_c[mid] <= r) lo = mid; else hi = mid;
    }
    return (int)s_sym[lo];
}

// Stable counting sort of the positions by symbol in ONE dedicated pass (sigma <= 256): the keys are
// the symbols themselves, so nothing but the position array is written -- 2 bytes read and 4 written
// per symbol instead of a key-building launch plus a generic (key, value) radix pass (30 bytes).
// Same scheme as radix_pass_kernel: 4096-position tiles from an atomic ticket, wave-striped stable
// ranking through per-wave LDS match masks, per-code decoupled look-back, LDS staging, coalesced
// write-out.  cbase[c] = first sorted row of code c.
#define ISC_NT 256
#define ISC_ITEMS 16
#define ISC_TILE (ISC_NT * ISC_ITEMS)
template <class Acc>
__global__ __launch_bounds__(ISC_NT) void ibwt_scatter_kernel(Acc acc, u32 N, Lut16 lut, CTable ct,
                                                             u32 *__restrict__ spos, u64 *status,
                                                             u32 *ticket, u32 *err) {
    constexpr int NW = ISC_NT / 64;
    __shared__ u32 s_hist[NW * 256];
    __shared__ u64 s_mask[NW * 256];
    __shared__ u32 s_vals[ISC_TILE];
    __shared__ u8 s_dig[ISC_TILE];
    __shared__ u32 s_dbase[256], s_gbase[256];
    __shared__ u16 s_lut[260];
    __shared__ u32 s_scan[NW + 1];
    __shared__ u32 s_tile;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    for (int i = tid; i < 257; i += ISC_NT) s_lut[i] = lut.v[i];
    if (tid == 0) s_tile = atomicAdd(ticket, 1u);
    for (int i = tid; i < NW * 256; i += ISC_NT) {
        s_hist[i] = 0;
        s_mask[i] = 0ull;
    }
    __syncthreads();
    const u32 tile = s_tile;
    const u64 base = (u64)tile * ISC_TILE;
    const u32 valid = (N - base) < (u64)ISC_TILE ? (u32)(N - base) : (u32)ISC_TILE;
    const u32 wofs = w * 64 * ISC_ITEMS;
    u32 *wh = s_hist + w * 256;
    u64 *wm = s_mask + w * 256;
    const u64 mybit = 1ull << l;
    u32 dig[ISC_ITEMS], rnk[ISC_ITEMS];
#pragma unroll
    for (int k = 0; k < ISC_ITEMS; k++) {
        const u32 p = wofs + k * 64 + l;
        const u32 d = p < valid ? (u32)s_lut[acc(base + p) + 1] : 255u;   // pads: last code, last
        dig[k] = d;
        atomicOr((unsigned long long *)&wm[d], (unsigned long long)mybit);
        __builtin_amdgcn_wave_barrier();
        const u64 m = wm[d];
        const u32 old = wh[d];
        const u32 prior = __popcll(m & (mybit - 1ull));
        __builtin_amdgcn_wave_barrier();
        if (prior == 0) {
            wh[d] = old + __popcll(m);
            wm[d] = 0ull;
        }
        __builtin_amdgcn_wave_barrier();
        rnk[k] = old + prior;
    }
    __syncthreads();
    u32 tot = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {   // one owner thread per code (ISC_NT == 256 codes)
        const u32 c = s_hist[i * 256 + tid];
        s_hist[i * 256 + tid] = tot;
        tot += c;
    }
    u32 tot_real = tot;
    if (tid == 255) tot_real = tot - (ISC_TILE - valid);
    u64 *st = status + (u64)tile * 256 + tid;
    lb_store(st, (tile == 0 ? LB_FLAG_INC : LB_FLAG_AGG) | (u64)tot_real);
    u32 dtot;
    const u32 dbase = block_excl_sum<ISC_NT>(tot, s_scan, &dtot);
    s_dbase[tid] = dbase;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ISC_ITEMS; k++) {
        const u32 pos = s_dbase[dig[k]] + wh[dig[k]] + rnk[k];
        s_vals[pos] = (u32)base + wofs + k * 64 + l;
        s_dig[pos] = (u8)dig[k];
    }
    u32 excl = 0;
    if (tile > 0) {
        i64 t = (i64)tile - 1;
        u32 spins = 0;
        for (;;) {
            const u64 sv = lb_load(status + (u64)t * 256 + tid);
            const u32 f = (u32)(sv >> 62);
            if (f == 0) {
                if (++spins > LB_SPIN_LIMIT) {
                    atomicOr(err, 2u);
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                continue;
            }
            excl += (u32)LB_VALUE(sv);
            if (f == 2 || t == 0) break;
            t--;
        }
        lb_store(st, LB_FLAG_INC | (u64)(excl + tot_real));
    }
    s_gbase[tid] = ct.c[tid] + excl - dbase;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ISC_ITEMS; k++) {
        const u32 p = tid + k * ISC_NT;
        if (p < valid) spos[s_gbase[s_dig[p]] + p] = s_vals[p];
    }
}

// Splitter q walks to the next splitter row (independent walks, one per lane) and records what it
// passes: the symbols of its segment go to seg + q * IBWT_SEGCAP, their
// number to seglen[q].  Flags: bit 0 = the segment is longer than IBWT_SEGCAP (listed, re-walked by
// ibwt_walk2_kernel), bit 1 = a Nothing row other than row 0 was met (an error only if the segment
// turns out to lie on the chain of row e).
#define IBWT_SEGCAP (16 * IBWT_S)   // segment lengths are geometric with mean IBWT_S: one in 10^7 is longer
// A record that fills up before a splitter row is met is closed where the walk stands and the walk goes
// on as a NEW splitter (slot K + e, e from the counter `nextra`, at most extra_cap of them): no segment is
// ever walked twice -- a second, serial walk of one long segment costs ~1 us per row, 4 ms for a full
// record, whatever the rest of the grid does.
__global__ __launch_bounds__(256) void ibwt_extra_init_kernel(u32 K, u32 Ktot, u32 *__restrict__ nxt,
                                                              u32 *__restrict__ dist, u32 *__restrict__ seglen,
                                                              u8 *__restrict__ segflag) {
    const u32 q = K + blockIdx.x * 256 + threadIdx.x;
    if (q >= Ktot) return;
    nxt[q] = q;      // inert until a walk takes the slot
    dist[q] = 0;
    seglen[q] = 0;
    segflag[q] = 0;
}
__global__ __launch_bounds__(256) void ibwt_walk1_kernel(const u32 *__restrict__ spos, u32 N, u32 K,
                                                         u32 *__restrict__ nxt,
                                                         u32 *__restrict__ dist, CTable ct,
                                                         u8 *__restrict__ seg, u32 *__restrict__ seglen,
                                                         u8 *__restrict__ segflag, u32 force_rewalk,
                                                         u32 *__restrict__ noverflow, u32 *__restrict__ ovlist,
                                                         u32 *__restrict__ nextra, u32 extra_cap, u32 segcap) {
    __shared__ u32 s_c[260];
    __shared__ i16 s_sym[260];
    for (int i = threadIdx.x; i < 260; i += 256) {
        s_c[i] = ct.c[i];
        s_sym[i] = ct.sym[i];
    }
    __syncthreads();
    u32 q = blockIdx.x * 256 + threadIdx.x;
    if (q >= K) return;
    u32 r = ibwt_split_row(q), steps = 0;
    if (r >= N) {  // the last, partial block may not hold its splitter row: inert slot
        nxt[q] = q;
        dist[q] = 0;
        seglen[q] = 0;
        segflag[q] = 0;
        return;
    }
    u8 *out = seg + (u64)q * IBWT_SEGCAP;
    u32 word = 0;
    uint4 w4 = make_uint4(0, 0, 0, 0);
    u32 emitted = 0, flag = force_rewalk ? 1u : 0u;
    // the load of the NEXT row is issued before this row's symbol is looked up and stored, so the
    // lookup hides behind the (dependent, random) load; the one load past the last row is harmless
    r = spos[r];
    for (;;) {
        steps++;
        const u32 rn = spos[r];
        if (r != 0) {
            const int sym = ibwt_sym_of_row(s_c, s_sym, ct.sigma, r);
            if (sym < 0) flag |= 2u;
            if (emitted < IBWT_SEGCAP) {
                word |= (u32)(u8)sym << (8 * (emitted & 3u));
                if ((emitted & 3u) == 3u) {   // sixteen symbols per store
                    const u32 k4 = (emitted >> 2) & 3u;
                    if (k4 == 0) w4.x = word;
                    else if (k4 == 1) w4.y = word;
                    else if (k4 == 2) w4.z = word;
                    else {
                        w4.w = word;
                        reinterpret_cast<uint4 *>(out)[emitted >> 4] = w4;
                    }
                    word = 0;
                }
            } else {
                flag |= 1u;
            }
            emitted++;
        }
        if (ibwt_is_splitter(r) || steps > N) break;
        if (emitted == segcap && extra_cap) {   // record full: close it here, go on as a new splitter
            const u32 e = atomicAdd(nextra, 1u);
            if (e < extra_cap) {
                nxt[q] = K + e;
                dist[q] = steps;
                seglen[q] = emitted;
                segflag[q] = (u8)flag;
                q = K + e;
                out = seg + (u64)q * IBWT_SEGCAP;
                steps = 0;
                emitted = 0;
                word = 0;
                flag = 0;
            }   // (else: slots used up -- the host repeats the walk the old way)
        }
        r = rn;
    }
    if (emitted < IBWT_SEGCAP && (emitted & 15u)) {   // the last, partial group of sixteen
        const u32 k4 = (emitted >> 2) & 3u;             // words already complete in w4: k4
        if (k4 == 0) w4.x = word;
        else if (k4 == 1) w4.y = word;
        else if (k4 == 2) w4.z = word;
        else w4.w = word;
        reinterpret_cast<uint4 *>(out)[emitted >> 4] = w4;
    }
    nxt[q] = r / IBWT_S;
    dist[q] = steps;
    seglen[q] = emitted;
    segflag[q] = (u8)flag;
    if (flag & 1u) ovlist[atomicAdd(noverflow, 1u)] = q;   // rare: re-walked by ibwt_walk2_kernel
}

// keep splitter 0's real successor aside and make it the terminal of the chain
__global__ void ibwt_terminal_kernel(u32 *nxt, u32 *dist, u64 *scalars) {
    scalars[4] = nxt[0];
    scalars[5] = dist[0];
    nxt[0] = 0;
    dist[0] = 0;
}

__global__ __launch_bounds__(256) void ibwt_jump_kernel(const u32 *__restrict__ nxt,
                                                        const u32 *__restrict__ dist,
                                                        u32 *__restrict__ nxt2,
                                                        u32 *__restrict__ dist2, u32 K) {
    u32 q = blockIdx.x * 256 + threadIdx.x;
    if (q >= K) return;
    u32 n = nxt[q];
    nxt2[q] = nxt[n];
    dist2[q] = dist[q] + dist[n];
}

// Lc = length of row e's cycle; scalars[6] = Lc
__global__ void ibwt_len_kernel(const u32 *nxt, const u32 *dist, u64 *scalars, u32 *err) {
    u32 n0 = (u32)scalars[4], d0 = (u32)scalars[5];
    if (n0 != 0 && nxt[n0] != 0) atomicOr(err, 4u);  // e's chain must come back to e
    scalars[6] = (u64)d0 + (n0 == 0 ? 0u : dist[n0]);
}

// second walk, only for on-chain segments that did not fit their buffer: bytes at final offsets
__global__ __launch_bounds__(256) void ibwt_walk2_kernel(const u32 *__restrict__ spos, u32 N, u32 K,
                                                         const u32 *__restrict__ nxt,
                                                         const u32 *__restrict__ dist,
                                                         const u64 *__restrict__ scalars,
                                                         CTable ct, u8 *__restrict__ text,
                                                         const u32 *__restrict__ ovlist, u32 nov, u32 *err) {
    __shared__ u32 s_c[260];
    __shared__ i16 s_sym[260];
    for (int i = threadIdx.x; i < 260; i += 256) {
        s_c[i] = ct.c[i];
        s_sym[i] = ct.sym[i];
    }
    __syncthreads();
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nov) return;
    const u32 q = ovlist[i];            // a segment that did not fit its record buffer
    if (q >= K || (q != 0 && nxt[q] != 0)) return;  // not on row e's chain
    const u32 Lc = (u32)scalars[6];
    u32 p = (q == 0) ? 0u : Lc - dist[q];
    u32 r = ibwt_split_row(q), steps = 0;
    while (steps++ <= N) {
        r = spos[r];
        if (r == 0) break;
        int sym = ibwt_sym_of_row(s_c, s_sym, ct.sigma, r);
        if (sym < 0) {
            atomicOr(err, 0x200u);  // fromJust Nothing (BWT/Internal.hs:195)
            break;
        }
        text[p++] = (u8)sym;
        if (ibwt_is_splitter(r)) break;
    }
}

// every splitter on row e's chain copies its recorded segment to its final offset (one wave per
// splitter, 64 bytes per step)
__global__ __launch_bounds__(256) void ibwt_copy_kernel(u32 K, const u32 *__restrict__ nxt,
                                                        const u32 *__restrict__ dist,
                                                        const u64 *__restrict__ scalars,
                                                        const u8 *__restrict__ seg,
                                                        const u32 *__restrict__ seglen,
                                                        const u8 *__restrict__ segflag,
                                                        u8 *__restrict__ text, u32 *err) {
    const u32 q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= K) return;
    if (q != 0 && nxt[q] != 0) return;  // not on row e's chain
    const u32 fl = segflag[q];
    if (fl & 2u) {
        if (lane_id() == 0) atomicOr(err, 0x200u);  // fromJust Nothing (BWT/Internal.hs:195)
        return;
    }
    if (fl & 1u) return;                // too long for its buffer: ibwt_walk2_kernel
    const u32 Lc = (u32)scalars[6];
    const u32 p0 = (q == 0) ? 0u : Lc - dist[q];
    const u32 len = seglen[q];
    const u8 *src = seg + (u64)q * IBWT_SEGCAP;
    for (u32 i = lane_id(); i < len; i += 64) text[p0 + i] = src[i];
}

// ---- small alphabets (<= 5 byte values, one Nothing): the walk by LF instead of by positions ------
// The position array of the sort above is 4 N bytes and every step of a walk reads 4 of them at a
// random place.  For DNA-like records the same step can be computed from the last column itself:
// LF(r) = C[c] + (occurrences of c = L[r] above row r) is the inverse of the step the reference
// takes (sorted[f].position), so walking it from row e = 0 visits the same cycle backwards and
// emits the text reversed.  The last column is kept as 64-byte lines of 128 rows: {occurrences of
// codes 0..3 above the line, three bit planes of the 3-bit codes}; a step reads ONE line of a
// structure of N / 2 bytes (0.5 GB for a 1 GiB record, half of it inside the 256 MB MALL) and
// needs no sort of the positions at all.  The Nothing row p is stored as code 0 and taken back out
// of code 0's counts; LF(p) = 0 closes the cycle.  Code 4's count is what is left of the line start.
#define LF_ROWS 128                 // rows per line
#define LF_BLOCK (256 * LF_ROWS)    // rows per block of the build kernels
#define LF_STRIDE 132               // staged bytes per line in LDS (33 words: conflict-free)
struct LfTable {
    u32 C[8];     // first sorted row of code c
    u8 sym[8];    // byte value of code c
};

// the symbols as MTF codes, one byte each (fused decode of a small alphabet): code 0 = Nothing,
// code c = the (c-1)-th byte value; as an accessor it yields c - 1, i.e. -1 or the LF code itself
struct CodeAcc {
    const u8 *c;
    __device__ __forceinline__ int operator()(u64 j) const { return (int)c[j] - 1; }
};
// pass 1: per block, occurrences of codes 0..3; the row of the Nothing -> *prim
// prim[0] = row of the (first) Nothing, prim[1] = number of Nothings, prim[2] != 0: a symbol outside the
// expected alphabet (lut value 0xff) was met
template <class Acc>
__global__ __launch_bounds__(256) void lf_count_kernel(Acc acc, u32 N, Lut8 lut, u32 *__restrict__ bcnt,
                                                       u32 *__restrict__ prim) {
    __shared__ u8 s_lut[260];
    __shared__ u32 s_c[4];
    for (int i = threadIdx.x; i < 257; i += 256) s_lut[i] = lut.v[i];
    if (threadIdx.x < 4) s_c[threadIdx.x] = 0;
    __syncthreads();
    const u64 base = (u64)blockIdx.x * LF_BLOCK;
    u32 c[4] = {0, 0, 0, 0};
    auto one = [&](u64 j, int sym) {
        if (sym < 0) {
            atomicMin(prim, (u32)j);
            atomicAdd(prim + 1, 1u);
        }
        const u32 code = s_lut[sym + 1];
        if (code == 0xffu && !prim[2]) atomicOr(prim + 2, 1u);
        c[0] += code == 0; c[1] += code == 1; c[2] += code == 2; c[3] += code == 3;
    };
    bool done = false;
    if constexpr (std::is_same<Acc, SymAcc>::value) {
        // eight symbols per 16-byte load (2-byte loads run at a fraction of the HBM rate)
        if (base + LF_BLOCK <= N && (((uintptr_t)(acc.s + base)) & 15) == 0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(acc.s + base);
            for (u32 g = threadIdx.x; g < LF_BLOCK / 8; g += 256) {
                const uint4 t = src[g];
                const u32 x[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int q = 0; q < 8; q++) one(base + 8 * g + q, (int)(i16)((x[q >> 1] >> (16 * (q & 1))) & 0xffffu));
            }
            done = true;
        }
    }
    if constexpr (std::is_same<Acc, CodeAcc>::value) {
        if (base + LF_BLOCK <= N && (((uintptr_t)(acc.c + base)) & 15) == 0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(acc.c + base);
            for (u32 g = threadIdx.x; g < LF_BLOCK / 16; g += 256) {
                const uint4 t = src[g];
                const u32 x[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int q = 0; q < 16; q++) one(base + 16 * g + q, (int)((x[q >> 2] >> (8 * (q & 3))) & 0xffu) - 1);
            }
            done = true;
        }
    }
    if (!done)
        for (u32 i = threadIdx.x; i < LF_BLOCK; i += 256) {
            const u64 j = base + i;
            if (j >= N) break;
            one(j, acc(j));
        }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        u32 v = c[k];
        for (int d = 32; d >= 1; d >>= 1) v += (u32)__shfl_xor((int)v, d, 64);
        if ((threadIdx.x & 63) == 0) atomicAdd(&s_c[k], v);
    }
    __syncthreads();
    if (threadIdx.x < 4) bcnt[(u64)blockIdx.x * 4 + threadIdx.x] = s_c[threadIdx.x];
}
// exclusive scan of the block counts, four counters side by side (one block)
__global__ __launch_bounds__(1024) void lf_scan_kernel(u32 *bcnt, u32 nb, u32 *__restrict__ totals) {
    __shared__ u32 s_part[1024][4];
    const u32 per = (nb + 1023) / 1024;
    const u32 lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    u32 v[4] = {0, 0, 0, 0};
    for (u32 t = lo; t < hi; t++)
        for (int k = 0; k < 4; k++) v[k] += bcnt[(u64)t * 4 + k];
    for (int k = 0; k < 4; k++) s_part[threadIdx.x][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 4) {
        u32 run = 0;
        for (int i = 0; i < 1024; i++) {
            const u32 c = s_part[i][threadIdx.x];
            s_part[i][threadIdx.x] = run;
            run += c;
        }
        totals[threadIdx.x] = run;   // occurrences of codes 0..3 in the whole column
    }
    __syncthreads();
    u32 run[4];
    for (int k = 0; k < 4; k++) run[k] = s_part[threadIdx.x][k];
    for (u32 t = lo; t < hi; t++)
        for (int k = 0; k < 4; k++) {
            const u32 c = bcnt[(u64)t * 4 + k];
            bcnt[(u64)t * 4 + k] = run[k];
            run[k] += c;
        }
}
// pass 2: the lines
template <class Acc>
__global__ __launch_bounds__(256) void lf_build_kernel(Acc acc, u32 N, Lut8 lut, const u32 *__restrict__ bcnt,
                                                       uint4 *__restrict__ lines) {
    __shared__ u8 s_lut[260];
    __shared__ u32 s_code[256 * LF_STRIDE / 4];
    __shared__ u64 s_wsum[4];
    for (int i = threadIdx.x; i < 257; i += 256) s_lut[i] = lut.v[i];
    __syncthreads();
    const u64 base = (u64)blockIdx.x * LF_BLOCK;
    u8 *sc = reinterpret_cast<u8 *>(s_code);
    bool staged = false;
    if constexpr (std::is_same<Acc, SymAcc>::value) {
        if (base + LF_BLOCK <= N && (((uintptr_t)(acc.s + base)) & 15) == 0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(acc.s + base);
            for (u32 g = threadIdx.x; g < LF_BLOCK / 8; g += 256) {
                const uint4 t = src[g];
                const u32 x[4] = {t.x, t.y, t.z, t.w};
                u32 w2[2] = {0, 0};
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int sym = (int)(i16)((x[q >> 1] >> (16 * (q & 1))) & 0xffffu);
                    w2[q >> 2] |= (u32)s_lut[sym + 1] << (8 * (q & 3));
                }
                const u32 i = 8 * g;   // eight codes of one line: two aligned words
                u32 *dst = reinterpret_cast<u32 *>(sc + (i / LF_ROWS) * LF_STRIDE + (i % LF_ROWS));
                dst[0] = w2[0];
                dst[1] = w2[1];
            }
            staged = true;
        }
    }
    if constexpr (std::is_same<Acc, CodeAcc>::value) {
        if (base + LF_BLOCK <= N && (((uintptr_t)(acc.c + base)) & 15) == 0) {
            const uint4 *src = reinterpret_cast<const uint4 *>(acc.c + base);
            for (u32 g = threadIdx.x; g < LF_BLOCK / 16; g += 256) {
                const uint4 t = src[g];
                const u32 x[4] = {t.x, t.y, t.z, t.w};
                u32 w4[4];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    w4[q] = 0;
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        w4[q] |= (u32)s_lut[(x[q] >> (8 * b)) & 0xffu] << (8 * b);   // lut index = (code - 1) + 1
                }
                const u32 i = 16 * g;   // sixteen codes of one line: four aligned words
                u32 *dst = reinterpret_cast<u32 *>(sc + (i / LF_ROWS) * LF_STRIDE + (i % LF_ROWS));
                dst[0] = w4[0]; dst[1] = w4[1]; dst[2] = w4[2]; dst[3] = w4[3];
            }
            staged = true;
        }
    }
    if (!staged)
        for (u32 i = threadIdx.x; i < LF_BLOCK; i += 256) {
            const u64 j = base + i;
            sc[(i / LF_ROWS) * LF_STRIDE + (i % LF_ROWS)] = j < N ? s_lut[acc(j) + 1] : (u8)0;
        }
    __syncthreads();
    const u32 *mine = s_code + threadIdx.x * (LF_STRIDE / 4);
    u64 pl[3][2] = {{0, 0}, {0, 0}, {0, 0}};
#pragma unroll
    for (int h = 0; h < 2; h++)
        for (int q = 0; q < 16; q++) {
            const u32 w = mine[h * 16 + q];
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const u32 code = (w >> (8 * b)) & 7u;
                const int bit = q * 4 + b;
                pl[0][h] |= (u64)(code & 1u) << bit;
                pl[1][h] |= (u64)((code >> 1) & 1u) << bit;
                pl[2][h] |= (u64)((code >> 2) & 1u) << bit;
            }
        }
    // occurrences of codes 0..3 in this line, 16 bits each (a block holds 32768 rows)
    u64 cnt = 0;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        u32 k = 0;
#pragma unroll
        for (int h = 0; h < 2; h++)
            k += (u32)__popcll(((c & 1) ? pl[0][h] : ~pl[0][h]) & ((c & 2) ? pl[1][h] : ~pl[1][h]) & ~pl[2][h]);
        cnt |= (u64)k << (16 * c);
    }
    // rows past N were staged as code 0: take them back out (only the last line of the text)
    {
        const u64 lstart = base + (u64)threadIdx.x * LF_ROWS;
        if (lstart + LF_ROWS > N) cnt -= (lstart >= N ? (u64)LF_ROWS : lstart + LF_ROWS - N);
    }
    u64 inc = cnt;
    for (int d = 1; d < 64; d <<= 1) {
        const u64 t = __shfl_up(inc, d, 64);
        if ((int)(threadIdx.x & 63) >= d) inc += t;
    }
    if ((threadIdx.x & 63) == 63) s_wsum[threadIdx.x >> 6] = inc;
    __syncthreads();
    u64 pre = inc - cnt;
    for (int i = 0; i < (int)(threadIdx.x >> 6); i++) pre += s_wsum[i];
    const u64 lstart = base + (u64)threadIdx.x * LF_ROWS;
    if (lstart < N) {
        const u32 *bc = bcnt + (u64)blockIdx.x * 4;
        uint4 *o = lines + (lstart / LF_ROWS) * 4;
        o[0] = make_uint4(bc[0] + (u32)(pre & 0xffff), bc[1] + (u32)((pre >> 16) & 0xffff),
                          bc[2] + (u32)((pre >> 32) & 0xffff), bc[3] + (u32)(pre >> 48));
        o[1] = make_uint4((u32)pl[0][0], (u32)(pl[0][0] >> 32), (u32)pl[0][1], (u32)(pl[0][1] >> 32));
        o[2] = make_uint4((u32)pl[1][0], (u32)(pl[1][0] >> 32), (u32)pl[1][1], (u32)(pl[1][1] >> 32));
        o[3] = make_uint4((u32)pl[2][0], (u32)(pl[2][0] >> 32), (u32)pl[2][1], (u32)(pl[2][1] >> 32));
    }
}

// one LF step; `code` = the code of L[r].  r == p (the Nothing row): LF = 0
__device__ __forceinline__ u32 lf_step(const uint4 *__restrict__ lines, const u32 *s_C, u32 p, u32 r, u32 &code) {
    const uint4 *ln = lines + (u64)(r / LF_ROWS) * 4;
    const uint4 a = ln[0], b0 = ln[1], b1 = ln[2], b2 = ln[3];
    const u32 o = r % LF_ROWS, hi = o >> 6, sh = o & 63u;
    const u64 p0[2] = {(u64)b0.x | ((u64)b0.y << 32), (u64)b0.z | ((u64)b0.w << 32)};
    const u64 p1[2] = {(u64)b1.x | ((u64)b1.y << 32), (u64)b1.z | ((u64)b1.w << 32)};
    const u64 p2[2] = {(u64)b2.x | ((u64)b2.y << 32), (u64)b2.z | ((u64)b2.w << 32)};
    const u32 c0 = (u32)((hi ? p0[1] : p0[0]) >> sh) & 1u, c1 = (u32)((hi ? p1[1] : p1[0]) >> sh) & 1u,
              c2 = (u32)((hi ? p2[1] : p2[0]) >> sh) & 1u;
    code = c0 | (c1 << 1) | (c2 << 2);
    const u64 m0 = (c0 ? p0[0] : ~p0[0]) & (c1 ? p1[0] : ~p1[0]) & (c2 ? p2[0] : ~p2[0]);
    const u64 m1 = (c0 ? p0[1] : ~p0[1]) & (c1 ? p1[1] : ~p1[1]) & (c2 ? p2[1] : ~p2[1]);
    const u64 below = (1ull << sh) - 1ull;
    const u32 in_line = hi ? (u32)__popcll(m0) + (u32)__popcll(m1 & below) : (u32)__popcll(m0 & below);
    const u32 lstart = r - o;
    const u32 above = code == 0 ? a.x : code == 1 ? a.y : code == 2 ? a.z : code == 3 ? a.w
                                                                              : lstart - (a.x + a.y + a.z + a.w);
    if (r == p) return 0u;
    return s_C[code] + above + in_line - ((code == 0 && r > p) ? 1u : 0u);
}

// the walks of ibwt_walk1_kernel with the LF step (same records, same chain bookkeeping)
__global__ __launch_bounds__(256) void ibwt_lfwalk1_kernel(const uint4 *__restrict__ lines, u32 N, u32 K,
                                                           const u32 *__restrict__ prim, LfTable tb,
                                                           u32 *__restrict__ nxt, u32 *__restrict__ dist,
                                                           u8 *__restrict__ seg, u32 *__restrict__ seglen,
                                                           u8 *__restrict__ segflag, u32 force_rewalk,
                                                           u32 *__restrict__ noverflow, u32 *__restrict__ ovlist,
                                                           u32 *__restrict__ nextra, u32 extra_cap, u32 segcap) {
    __shared__ u32 s_C[8];
    __shared__ u8 s_sym[8];
    if (threadIdx.x < 8) {
        s_C[threadIdx.x] = tb.C[threadIdx.x];
        s_sym[threadIdx.x] = tb.sym[threadIdx.x];
    }
    __syncthreads();
    const u32 p = *prim;
    u32 q = blockIdx.x * 256 + threadIdx.x;
    if (q >= K) return;
    u32 r = ibwt_split_row(q), steps = 0;
    if (r >= N) {
        nxt[q] = q;
        dist[q] = 0;
        seglen[q] = 0;
        segflag[q] = 0;
        return;
    }
    u8 *out = seg + (u64)q * IBWT_SEGCAP;
    u32 word = 0;
    uint4 w4 = make_uint4(0, 0, 0, 0);
    u32 emitted = 0, flag = force_rewalk ? 1u : 0u;
    for (;;) {
        steps++;
        u32 code;
        const u32 rn = lf_step(lines, s_C, p, r, code);
        if (rn != 0) {   // (rn == 0: r was the Nothing row -- nothing to emit)
            if (emitted < IBWT_SEGCAP) {
                word |= (u32)s_sym[code] << (8 * (emitted & 3u));
                if ((emitted & 3u) == 3u) {
                    const u32 k4 = (emitted >> 2) & 3u;
                    if (k4 == 0) w4.x = word;
                    else if (k4 == 1) w4.y = word;
                    else if (k4 == 2) w4.z = word;
                    else {
                        w4.w = word;
                        reinterpret_cast<uint4 *>(out)[emitted >> 4] = w4;
                    }
                    word = 0;
                }
            } else {
                flag |= 1u;
            }
            emitted++;
        }
        r = rn;
        if (ibwt_is_splitter(r) || steps > N) break;
        if (emitted == segcap && extra_cap) {   // record full: go on as a new splitter (ibwt_walk1_kernel)
            const u32 e = atomicAdd(nextra, 1u);
            if (e < extra_cap) {
                nxt[q] = K + e;
                dist[q] = steps;
                seglen[q] = emitted;
                segflag[q] = (u8)flag;
                q = K + e;
                out = seg + (u64)q * IBWT_SEGCAP;
                steps = 0;
                emitted = 0;
                word = 0;
                flag = 0;
            }
        }
    }
    if (emitted < IBWT_SEGCAP && (emitted & 15u)) {
        const u32 k4 = (emitted >> 2) & 3u;
        if (k4 == 0) w4.x = word;
        else if (k4 == 1) w4.y = word;
        else if (k4 == 2) w4.z = word;
        else w4.w = word;
        reinterpret_cast<uint4 *>(out)[emitted >> 4] = w4;
    }
    nxt[q] = r / IBWT_S;
    dist[q] = steps;
    seglen[q] = emitted;
    segflag[q] = (u8)flag;
    if (flag & 1u) ovlist[atomicAdd(noverflow, 1u)] = q;
}
// segments that did not fit their record buffer: walked again, bytes at their final (reversed) offsets
__global__ __launch_bounds__(256) void ibwt_lfwalk2_kernel(const uint4 *__restrict__ lines, u32 N, u32 K,
                                                           const u32 *__restrict__ prim, LfTable tb,
                                                           const u32 *__restrict__ nxt, const u32 *__restrict__ dist,
                                                           const u64 *__restrict__ scalars, u8 *__restrict__ text,
                                                           const u32 *__restrict__ ovlist, u32 nov) {
    __shared__ u32 s_C[8];
    __shared__ u8 s_sym[8];
    if (threadIdx.x < 8) {
        s_C[threadIdx.x] = tb.C[threadIdx.x];
        s_sym[threadIdx.x] = tb.sym[threadIdx.x];
    }
    __syncthreads();
    const u32 i = blockIdx.x * 256 + threadIdx.x;
    if (i >= nov) return;
    const u32 q = ovlist[i];
    if (q >= K || (q != 0 && nxt[q] != 0)) return;
    const u32 p = *prim, Lc = (u32)scalars[6];
    u32 k = (q == 0) ? 0u : Lc - dist[q];   // offset in the reversed text
    u32 r = ibwt_split_row(q), steps = 0;
    while (steps++ <= N) {
        u32 code;
        r = lf_step(lines, s_C, p, r, code);
        if (r == 0) break;
        if (k + 2 <= Lc) text[Lc - 2 - k] = s_sym[code];
        k++;
        if (ibwt_is_splitter(r)) break;
    }
}
// ibwt_copy_kernel for the reversed walk.  A wave per splitter: the record comes in with 16-byte
// loads (into LDS), leaves mirrored with 4-byte stores (one-byte copies ran at 1.2 TB/s).
__global__ __launch_bounds__(256) void ibwt_copy_rev_kernel(u32 K, const u32 *__restrict__ nxt,
                                                            const u32 *__restrict__ dist,
                                                            const u64 *__restrict__ scalars,
                                                            const u8 *__restrict__ seg,
                                                            const u32 *__restrict__ seglen,
                                                            const u8 *__restrict__ segflag,
                                                            u8 *__restrict__ text) {
    __shared__ __attribute__((aligned(16))) u8 s_seg[4][IBWT_SEGCAP];
    const u32 w = threadIdx.x >> 6, l = lane_id();
    const u32 q = blockIdx.x * 4 + w;
    if (q >= K) return;
    if (q != 0 && nxt[q] != 0) return;
    if (segflag[q] & 1u) return;
    const u32 Lc = (u32)scalars[6];
    const u32 k0 = (q == 0) ? 0u : Lc - dist[q];
    u32 len = seglen[q];
    if (k0 + 2 > Lc) return;
    if (len > Lc - 1 - k0) len = Lc - 1 - k0;       // (a valid chain never needs this)
    if (len == 0) return;
    const u8 *src = seg + (u64)q * IBWT_SEGCAP;
    u8 *sb = s_seg[w];
    for (u32 i = l * 16; i < len; i += 64 * 16)
        *reinterpret_cast<uint4 *>(sb + i) = *reinterpret_cast<const uint4 *>(src + i);
    __builtin_amdgcn_wave_barrier();
    // text[D - i] = src[i]; the destination bytes are [D - len + 1, D]
    const u64 D = (u64)Lc - 2 - k0;
    u8 *lo = text + (D - len + 1), *hi = text + D + 1;                 // [lo, hi)
    u8 *alo = reinterpret_cast<u8 *>((reinterpret_cast<uintptr_t>(lo) + 3) & ~(uintptr_t)3);
    u8 *ahi = reinterpret_cast<u8 *>(reinterpret_cast<uintptr_t>(hi) & ~(uintptr_t)3);
    if (alo >= ahi) {   // shorter than an aligned word: bytes
        for (u32 i = l; i < len; i += 64) text[D - i] = sb[i];
        return;
    }
    const u32 head = (u32)(alo - lo), tail = (u32)(hi - ahi), words = (u32)(ahi - alo) / 4;
    if (l < head) lo[l] = sb[len - 1 - l];
    if (l < tail) ahi[l] = sb[(u32)(hi - ahi) - 1 - l];
    for (u32 j = l; j < words; j += 64) {
        const u32 i3 = (u32)(text + D - (alo + 4 * j));   // source index of the word's first byte
        const u32 v = (u32)sb[i3] | ((u32)sb[i3 - 1] << 8) | ((u32)sb[i3 - 2] << 16) | ((u32)sb[i3 - 3] << 24);
        *reinterpret_cast<u32 *>(alo + 4 * j) = v;
    }
}

#endif  // __HIPCC__

// Inverse BWT of an accessor stream; d_text receives *n_out bytes (<= N).
template <class Acc>
// alphabet (optional, fused decode): the sorted symbols the stream can hold (the block's MTF list);
// lets the small-alphabet path count on the device instead of taking a histogram first
static void ibwt_device(tc_ctx *ctx, Arena &A, Acc acc, u64 N, const u32 *counts257, u8 *d_text,
                        u64 *n_out, bool dry, const i16 *alphabet = nullptr, u32 nalphabet = 0) {
    const u32 K = tc_cdiv(N, IBWT_S);
    u32 *d_counts = A.get<u32>(260);
    // (k0 doubles as the segment record buffer after the sort: K * IBWT_SEGCAP bytes)
    // splitters: one per IBWT_S rows + slots for walks that outgrow a record (ibwt_walk1_kernel)
    const u32 Kx = K / 32 + 256, Kt = K + Kx;
    u64 *k0 = A.get<u64>(N > (u64)Kt * (IBWT_SEGCAP / 8) ? N : (u64)Kt * (IBWT_SEGCAP / 8));
    u64 *k1 = A.get<u64>(N);
    u32 *v0 = A.get<u32>(N);
    u32 *v1 = A.get<u32>(N);
    u32 *hist = A.get<u32>(RDX_MAX_PASSES * RDX_BINS);
    u64 *rstatus = A.get<u64>(radix_status_words(N));
    u32 *nx[2] = {A.get<u32>(Kt + 1), A.get<u32>(Kt + 1)};
    u32 *ds[2] = {A.get<u32>(Kt + 1), A.get<u32>(Kt + 1)};
    u32 *seglen = A.get<u32>(Kt + 1);
    u8 *segflag = A.get<u8>(Kt + 1);
    u32 *ovlist = A.get<u32>(Kt + 1);
    uint4 *lf_lines = A.get<uint4>(((size_t)(N / LF_ROWS) + 1) * 4);   // LF walk: N / 2 bytes
    u32 *lf_bcnt = A.get<u32>(((size_t)tc_cdiv(N, LF_BLOCK) + 1) * 4);
    if (dry) return;
    hipStream_t s = ctx->stream;
    *n_out = 0;
    // walks (recording) -> chain ranking -> copy-out, shared by both step functions.  First with slots for
    // walks that outgrow their record; if those run out (or TC_IBWT_REWALK asks for the old way) once more
    // without, the long segments then being walked a second time.
    u8 *seg = reinterpret_cast<u8 *>(k0);
    auto walk_rank_copy = [&](auto launch_walk1, auto launch_walk2, bool reversed) {
        const u32 rewalk = (u32)env_int("TC_IBWT_REWALK", 0);
        // (tests: records "full" after fewer symbols, so that small inputs reach the extra slots and their exhaustion)
        u32 segcap = (u32)env_int("TC_IBWT_SEGCAP", IBWT_SEGCAP) & ~15u;
        if (segcap < 16 || segcap > IBWT_SEGCAP) segcap = IBWT_SEGCAP;
        u32 *nov_ctr = reinterpret_cast<u32 *>(ctx->d_scalars + 17);   // [0] overflowed records, [1] extra slots taken
        for (int attempt = rewalk ? 1 : 0; attempt < 2; attempt++) {
            const u32 xcap = attempt == 0 ? Kx : 0u;
            const u32 Ka = K + xcap;    // slots that take part in the ranking
            tc_memset_async(ctx, ctx->d_scalars + 17, 0, sizeof(u64));
            if (xcap) {
                ibwt_extra_init_kernel<<<tc_cdiv(xcap, 256), 256, 0, s>>>(K, Ka, nx[0], ds[0], seglen, segflag);
                TC_LAUNCH_CHECK(ctx);
            }
            launch_walk1(rewalk, nov_ctr, nov_ctr + 1, xcap, segcap);
            TC_LAUNCH_CHECK(ctx);
            ibwt_terminal_kernel<<<1, 1, 0, s>>>(nx[0], ds[0], ctx->d_scalars);
            TC_LAUNCH_CHECK(ctx);
            int cur = 0;
            for (int r = 0; r < ceil_log2_u64(Ka) + 1; r++) {
                ibwt_jump_kernel<<<tc_cdiv(Ka, 256), 256, 0, s>>>(nx[cur], ds[cur], nx[cur ^ 1], ds[cur ^ 1], Ka);
                TC_LAUNCH_CHECK(ctx);
                cur ^= 1;
            }
            ibwt_len_kernel<<<1, 1, 0, s>>>(nx[cur], ds[cur], ctx->d_scalars, ctx->d_err);
            TC_LAUNCH_CHECK(ctx);
            tc_d2h(ctx, &ctx->h_scalars[6], ctx->d_scalars + 6, sizeof(u64));
            tc_d2h(ctx, &ctx->h_scalars[17], ctx->d_scalars + 17, sizeof(u64));
            TC_HIP(ctx, hipStreamSynchronize(s));
            const u32 nov = (u32)(ctx->h_scalars[17] & 0xffffffffu), taken = (u32)(ctx->h_scalars[17] >> 32);
            if (xcap && taken > xcap) continue;   // slots ran out: the old way
            if (reversed)
                ibwt_copy_rev_kernel<<<tc_cdiv(Ka, 4), 256, 0, s>>>(Ka, nx[cur], ds[cur], ctx->d_scalars, seg, seglen,
                                                                   segflag, d_text);
            else
                ibwt_copy_kernel<<<tc_cdiv(Ka, 4), 256, 0, s>>>(Ka, nx[cur], ds[cur], ctx->d_scalars, seg, seglen,
                                                               segflag, d_text, ctx->d_err);
            TC_LAUNCH_CHECK(ctx);
            if (nov) {
                launch_walk2(nx[cur], ds[cur], nov);
                TC_LAUNCH_CHECK(ctx);
            }
            TC_HIP(ctx, hipStreamSynchronize(s));
            break;
        }
        const u64 Lc = ctx->h_scalars[6];
        *n_out = Lc ? Lc - 1 : 0;
    };
    // small alphabet, one Nothing: walk by LF over the packed last column (see above).  syms = the
    // byte values in order (codes 0..nsym-1); cnt = their occurrences, or null: counted on the
    // device (then the walk is only taken if the column holds exactly one Nothing and nothing else
    // than these symbols -- returns false otherwise, nothing written)
    auto lf_path = [&](const u8 *syms, u32 nsym, const u32 *cnt) -> bool {
        Lut8 l8;
        LfTable tb = {};
        for (int v = 0; v < 257; v++) l8.v[v] = 0xff;
        l8.v[0] = 0;
        for (u32 c = 0; c < nsym; c++) {
            // (a CodeAcc stream already holds the codes: its "symbol" c is code c)
            l8.v[(std::is_same<Acc, CodeAcc>::value ? c : (u32)syms[c]) + 1] = (u8)c;
            tb.sym[c] = syms[c];
        }
        uint4 *lines = lf_lines;
        u32 *bcnt = lf_bcnt;
        u32 *prim = reinterpret_cast<u32 *>(ctx->d_scalars + 24);   // 4 words + 4 totals
        const u32 nb = tc_cdiv(N, LF_BLOCK);
        tc_memset_async(ctx, prim, 0, 4 * sizeof(u64));
        tc_memset_async(ctx, prim, 0xff, sizeof(u32));
        lf_count_kernel<Acc><<<nb, 256, 0, s>>>(acc, (u32)N, l8, bcnt, prim);
        TC_LAUNCH_CHECK(ctx);
        lf_scan_kernel<<<1, 1024, 0, s>>>(bcnt, nb, prim + 4);
        TC_LAUNCH_CHECK(ctx);
        u32 tot[5] = {0, 0, 0, 0, 0};
        if (cnt) {
            for (u32 c = 0; c < nsym; c++) tot[c] = cnt[c];
        } else {
            tc_d2h(ctx, &ctx->h_scalars[24], ctx->d_scalars + 24, 4 * sizeof(u64));
            TC_HIP(ctx, hipStreamSynchronize(s));
            const u32 *hp = reinterpret_cast<const u32 *>(&ctx->h_scalars[24]);
            if (hp[1] != 1 || hp[2] != 0) return false;
            u64 sum4 = 0;
            for (int c = 0; c < 4; c++) { tot[c] = hp[4 + c]; sum4 += hp[4 + c]; }
            tot[0] -= 1;                       // the Nothing was counted as code 0
            tot[4] = (u32)(N - sum4);
        }
        u32 crow = 1;   // row 0 is the Nothing's
        for (u32 c = 0; c < 5; c++) {
            tb.C[c] = crow;
            crow += tot[c];
        }
        lf_build_kernel<Acc><<<nb, 256, 0, s>>>(acc, (u32)N, l8, bcnt, lines);
        TC_LAUNCH_CHECK(ctx);
        walk_rank_copy(
            [&](u32 rewalk, u32 *nov_ctr, u32 *nextra, u32 xcap, u32 segcap) {
                ibwt_lfwalk1_kernel<<<tc_cdiv(K, 256), 256, 0, s>>>(lines, (u32)N, K, prim, tb, nx[0], ds[0], seg, seglen,
                                                                   segflag, rewalk, nov_ctr, ovlist, nextra, xcap, segcap);
            },
            [&](const u32 *nxf, const u32 *dsf, u32 nov) {
                ibwt_lfwalk2_kernel<<<tc_cdiv(nov, 256), 256, 0, s>>>(lines, (u32)N, K, prim, tb, nxf, dsf, ctx->d_scalars,
                                                                     d_text, ovlist, nov);
            },
            true);
        return true;
    };
    const bool lf_on = N > 1 && env_int("TC_IBWT_LF", 1) != 0;
    if (lf_on && !counts257 && alphabet && nalphabet >= 2 && nalphabet <= 6 && alphabet[0] == -1) {
        // the caller knows the alphabet: no histogram pass
        u8 syms[5];
        bool ok = true;
        for (u32 i = 1; i < nalphabet; i++) {
            ok = ok && alphabet[i] > alphabet[i - 1] && alphabet[i] <= 255;
            syms[i - 1] = (u8)alphabet[i];
        }
        if (ok && lf_path(syms, nalphabet - 1, nullptr)) return;
    }
    if constexpr (std::is_same<Acc, CodeAcc>::value) {
        *n_out = ~0ull;   // a code stream only takes the LF walk; the caller falls back to symbols
        return;
    } else {
    u32 local[257];
    if (!counts257) {
        sym_hist_host<Acc>(ctx, acc, N, d_counts, local);
        counts257 = local;
    }
    if (counts257[0] == 0) return;  // no Nothing: magicInverseBWT returns empty (:173-174)
    Alphabet al;
    al.build(counts257);
    if (lf_on && counts257[0] == 1 && al.sigma <= 6) {
        u8 syms[5];
        u32 cnt[5];
        for (u32 c = 1; c < al.sigma; c++) {
            syms[c - 1] = (u8)al.sym_of_code[c];
            cnt[c - 1] = counts257[al.sym_of_code[c] + 1];
        }
        if (lf_path(syms, al.sigma - 1, cnt)) return;
    }
    Lut16 lut;
    CTable ct;
    u32 acc_c = 0;
    for (int v = 0; v < 257; v++) lut.v[v] = al.code_of_sym[v];
    for (u32 c = 0; c < al.sigma; c++) {
        ct.c[c] = acc_c;
        ct.sym[c] = al.sym_of_code[c];
        acc_c += counts257[al.sym_of_code[c] + 1];
    }
    ct.c[al.sigma] = (u32)N;
    ct.sigma = al.sigma;
    // 1. sorted (symbol, position): spos
    const u32 *spos = nullptr;
    if (al.sigma <= 256 && env_int("TC_IBWT_SCATTER", 1) != 0) {
        for (u32 c = al.sigma; c < 260; c++) ct.c[c] = (u32)N;   // codes that do not occur
        const u32 stiles = tc_cdiv(N, ISC_TILE);
        u32 *sticket = reinterpret_cast<u32 *>(rstatus + (size_t)stiles * 256);
        tc_memset_async(ctx, rstatus, 0, ((size_t)stiles * 256 + 2) * sizeof(u64));
        ibwt_scatter_kernel<Acc><<<stiles, ISC_NT, 0, s>>>(acc, (u32)N, lut, ct, v0, rstatus, sticket, ctx->d_err);
        TC_LAUNCH_CHECK(ctx);
        spos = v0;
    } else {
        ibwt_keys_kernel<Acc><<<tc_cdiv(N, 256), 256, 0, s>>>(acc, (u32)N, lut, k0);
        TC_LAUNCH_CHECK(ctx);
        RadixPlan plan;
        plan.add_range(0, ceil_log2_u64(al.sigma) > 0 ? ceil_log2_u64(al.sigma) : 1);
        RadixBuffers rb;
        rb.keys = k0; rb.keys_alt = k1; rb.vals = v0; rb.vals_alt = v1;
        rb.hist = hist; rb.status = rstatus;
        radix_sort_pairs(ctx, rb, (u32)N, plan, /*gen_idx=*/true, /*hist_ready=*/false);
        spos = rb.vals;
    }
    // k0 holds the segment records ((K + Kx) * IBWT_SEGCAP ~ 16.5 N bytes; in the scatter path it is not used for keys at all)
    // 2. splitter walks (recording the symbols passed), chain ranking, copy-out
    walk_rank_copy(
        [&](u32 rewalk, u32 *nov_ctr, u32 *nextra, u32 xcap, u32 segcap) {
            ibwt_walk1_kernel<<<tc_cdiv(K, 256), 256, 0, s>>>(spos, (u32)N, K, nx[0], ds[0], ct, seg, seglen, segflag,
                                                             rewalk, nov_ctr, ovlist, nextra, xcap, segcap);
        },
        [&](const u32 *nxf, const u32 *dsf, u32 nov) {
            ibwt_walk2_kernel<<<tc_cdiv(nov, 256), 256, 0, s>>>(spos, (u32)N, K, nxf, dsf, ctx->d_scalars, ct, d_text,
                                                               ovlist, nov, ctx->d_err);
        },
        false);
    }   // (!CodeAcc)
}

// ---- inverse MTF ---------------------------------------------------------------
template <int ROWS>
static void imtf_launch(tc_ctx *ctx, const u16 *d_idx, u64 N, u32 sigma, u16 *perms, u32 chunks,
                        const SymTab &tab, i16 *d_out) {
    hipStream_t s = ctx->stream;
    imtf_summary_kernel<ROWS><<<chunks, 64, 0, s>>>(d_idx, N, sigma, perms, ctx->d_err);
    TC_LAUNCH_CHECK(ctx);
    imtf_scan_kernel<ROWS><<<1, 64 * MTFG_SCAN_WAVES, 0, s>>>(perms, chunks);
    TC_LAUNCH_CHECK(ctx);
    imtf_apply_kernel<ROWS><<<chunks, 64, 0, s>>>(d_idx, N, sigma, perms, tab, d_out);
    TC_LAUNCH_CHECK(ctx);
}

// sigma <= 256: one chunk per lane (tc_mtf.hpp, "inverse MTF, lane chunks")
template <int ROWS>
static void imtf_lane_launch(tc_ctx *ctx, const u16 *d_idx, u64 N, u32 sigma, u16 *perms,
                             const SymTab &tab, i16 *d_out) {
    hipStream_t s = ctx->stream;
    GmiArgs a;
    a.N = N; a.sigma = sigma; a.ls = ((sigma + 3) / 4) | 1u;
    a.idx = d_idx; a.perms = perms; a.tab = tab; a.out = d_out; a.err = ctx->d_err;
    const u32 tiles = tc_cdiv(N, GM_TILE);
    const size_t lds = gm_lds_bytes(a.ls);
    TC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(imtf_gm_kernel<ROWS, false>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    TC_HIP(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(imtf_gm_kernel<ROWS, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    imtf_gm_kernel<ROWS, false><<<tiles, GM_NT, lds, s>>>(a);
    TC_LAUNCH_CHECK(ctx);
    imtf_scan_kernel<ROWS><<<1, 64 * MTFG_SCAN_WAVES, 0, s>>>(perms, tiles);
    TC_LAUNCH_CHECK(ctx);
    imtf_gm_kernel<ROWS, true><<<tiles, GM_NT, lds, s>>>(a);
    TC_LAUNCH_CHECK(ctx);
}

// seqFromMTF: initial list = sort(unique(list)) (MTF/Internal.hs:214).
// list = 16 nibbles in a register (tc_mtf.hpp, "inverse MTF, sigma <= 16"); t_perm: tiles + 1 words
template <class IT, class OT>
static void imtf_nib_device(tc_ctx *ctx, u64 *t_perm, const IT *d_idx, u64 N, u32 sigma, const SymTab &tab,
                            OT *d_out) {
    hipStream_t s = ctx->stream;
    const u32 tiles = tc_cdiv(N, MTF_TILE);
    imtf_nib_summary_kernel<IT><<<tiles, MTF_NT, 0, s>>>(d_idx, N, sigma, t_perm, ctx->d_err);
    TC_LAUNCH_CHECK(ctx);
    imtf_nib_scan_kernel<<<1, MTF_NT, 0, s>>>(t_perm, tiles);
    TC_LAUNCH_CHECK(ctx);
    imtf_nib_apply_kernel<IT, OT><<<tiles, MTF_NT, 0, s>>>(d_idx, N, sigma, t_perm, tab, d_out, ctx->d_err);
    TC_LAUNCH_CHECK(ctx);
}
__global__ __launch_bounds__(256) void codes_to_syms_kernel(const u8 *__restrict__ codes, u64 N, SymTab tab,
                                                            i16 *__restrict__ out) {
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < N; j += (u64)gridDim.x * 256) out[j] = tab.v[codes[j]];
}

// primary / d_idx_rw (fused decode only): the row of the one sentinel and the same index buffer,
// writable -- lets sigma = 257 take the 256-symbol lane chunks (tc_mtf.hpp, "sigma = 257")
static void mtf_decode_device(tc_ctx *ctx, Arena &A, const u16 *d_idx, u64 N, const i16 *list,
                              u32 nlist, i16 *d_out, bool dry, i64 primary = -1, u16 *d_idx_rw = nullptr) {
    const u32 chunks = tc_cdiv(N, MTFG_CH);
    u16 *perms = A.get<u16>(((size_t)chunks + 1) * 320);
    const u32 tiles257 = tc_cdiv(N, M257_TILE);
    u16 *tmax = A.get<u16>((size_t)tiles257 + 8);
    u32 *fix = A.get<u32>(512);
    if (dry) return;
    bool seen[257] = {false};
    for (u32 i = 0; i < nlist; i++) {
        if (list[i] < -1 || list[i] > 255) TC_FAIL(ctx, TC_ERR_ARG, "list symbol out of range");
        seen[list[i] + 1] = true;
    }
    SymTab tab;
    u32 sigma = 0;
    for (int v = 0; v < 257; v++)
        if (seen[v]) tab.v[sigma++] = (i16)(v - 1);
    for (u32 v = sigma; v < 260; v++) tab.v[v] = 0;
    if (sigma == 257 && d_idx_rw && primary > 0 && (u64)primary < N && env_int("TC_MTF_WAVE_CHUNKS", 0) == 0 &&
        env_int("TC_MTF_SENTINEL_SPLIT", 1) != 0) {
        hipStream_t s = ctx->stream;
        u64 *res = ctx->d_scalars + 20;
        tc_memset_async(ctx, res, 0, 3 * sizeof(u64));
        imtf257_tmax_kernel<<<tiles257, 256, 0, s>>>(d_idx, N, tmax);
        TC_LAUNCH_CHECK(ctx);
        imtf257_chain_kernel<<<1, 256, 0, s>>>(d_idx, N, (u64)primary, tmax, fix, res);
        TC_LAUNCH_CHECK(ctx);
        u32 grid = tc_cdiv(N, 256 * 16);
        if (grid > 8192) grid = 8192;
        imtf257_check_kernel<<<grid, 256, 0, s>>>(d_idx, N, (u64)primary, fix, res);
        TC_LAUNCH_CHECK(ctx);
        tc_d2h(ctx, &ctx->h_scalars[20], res, 3 * sizeof(u64));
        TC_HIP(ctx, hipStreamSynchronize(s));
        if (ctx->h_scalars[22] == 0) {
            imtf257_apply_kernel<<<1, 512, 0, s>>>(d_idx_rw, (u64)primary, fix, res);
            TC_LAUNCH_CHECK(ctx);
            SymTab bytes;
            for (int v = 0; v < 260; v++) bytes.v[v] = (i16)(v < 256 ? v : 0);
            imtf_lane_launch<4>(ctx, d_idx, N, 256, perms, bytes, d_out);
            imtf257_sentinel_kernel<<<1, 1, 0, s>>>(d_out, (u64)primary);
            TC_LAUNCH_CHECK(ctx);
            return;
        }
        // not the index stream of a BWT with its sentinel at `primary`: the nine-bit path below
    }
    if (sigma <= 16 && env_int("TC_MTF_FORCE_GENERAL", 0) == 0 && env_int("TC_MTF_WAVE_CHUNKS", 0) == 0) {
        imtf_nib_device<u16, i16>(ctx, reinterpret_cast<u64 *>(perms), d_idx, N, sigma, tab, d_out);
        return;
    }
    if (sigma <= 256 && env_int("TC_MTF_WAVE_CHUNKS", 0) == 0) {
        const u32 rows = (sigma + 63) / 64;
        if (rows <= 1) imtf_lane_launch<1>(ctx, d_idx, N, sigma, perms, tab, d_out);
        else if (rows == 2) imtf_lane_launch<2>(ctx, d_idx, N, sigma, perms, tab, d_out);
        else if (rows == 3) imtf_lane_launch<3>(ctx, d_idx, N, sigma, perms, tab, d_out);
        else imtf_lane_launch<4>(ctx, d_idx, N, sigma, perms, tab, d_out);
    } else if (sigma <= 64) imtf_launch<1>(ctx, d_idx, N, sigma, perms, chunks, tab, d_out);
    else if (sigma <= 128) imtf_launch<2>(ctx, d_idx, N, sigma, perms, chunks, tab, d_out);
    else imtf_launch<5>(ctx, d_idx, N, sigma, perms, chunks, tab, d_out);
}

// ---- inverse RLE ---------------------------------------------------------------
template <class SymT, class OutT = SymT>
static void rle_decode_device(tc_ctx *ctx, Arena &A, const u32 *d_counts, const SymT *d_syms,
                              u64 nruns, bool has_nothing, OutT *d_out, u64 cap, u64 *N_out,
                              bool dry) {
    const u32 ntiles = tc_cdiv(nruns, RLD_TILE);
    u64 *status = A.get<u64>((size_t)ntiles + 4);
    const u32 huge_cap = (u32)(cap / RLE_HUGE + 2);
    HugeRun *huge = A.get<HugeRun>(huge_cap);
    u32 *nhuge = A.get<u32>(4);
    if (dry) return;
    hipStream_t s = ctx->stream;
    tc_memset_async(ctx, nhuge, 0, 4 * sizeof(u32));
    tc_memset_async(ctx, status, 0, ((size_t)ntiles + 4) * sizeof(u64));
    tc_memset_async(ctx, ctx->d_scalars + 7, 0, sizeof(u64));
    RleDecArgs a;
    a.counts = d_counts; a.syms = d_syms; a.nruns = nruns; a.has_nothing = has_nothing ? 1 : 0;
    a.cap = cap; a.out = d_out; a.status = status;
    a.ticket = reinterpret_cast<u32 *>(status + ntiles + 2);
    a.total = ctx->d_scalars + 7; a.err = ctx->d_err;
    a.huge = huge; a.nhuge = nhuge; a.huge_cap = huge_cap; a.ntiles = ntiles;
    u32 grid = tc_persistent_grid_for(ctx, rle_decode_fused_kernel<SymT, OutT>, RLD_NT, 8);
    if (grid > ntiles) grid = ntiles;
    rle_decode_fused_kernel<SymT, OutT><<<grid, RLD_NT, 0, s>>>(a);
    TC_LAUNCH_CHECK(ctx);
    rle_fill_huge_kernel<OutT><<<tc_persistent_grid(ctx, 4), 256, 0, s>>>(huge, nhuge, huge_cap, cap, d_out);
    TC_LAUNCH_CHECK(ctx);
    tc_d2h(ctx, &ctx->h_scalars[7], ctx->d_scalars + 7, sizeof(u64));
    TC_HIP(ctx, hipStreamSynchronize(s));
    *N_out = ctx->h_scalars[7];   // > cap: the caller reports TC_ERR_CAPACITY (nothing past cap was written)
}

// ---- fused decode: runs -> MTF indices -> BWT symbols -> text -------------------
static void decode_device(tc_ctx *ctx, const tc_block *blk, u8 *d_text) {
    const u64 n = blk->n, N = n + 1;
    u16 *d_idx = nullptr;
    i16 *d_sym = nullptr;
    u64 got = 0, n_out = 0;
    i16 sorted_list[TC_MAX_SIGMA];
    {
        const u32 sg = blk->sigma <= TC_MAX_SIGMA ? blk->sigma : 0;
        for (u32 i = 0; i < sg; i++) sorted_list[i] = blk->final_list[i];
        std::sort(sorted_list, sorted_list + sg);
    }
    // small alphabets (a proper list: distinct symbols in range): the byte-wide pipeline
    bool small = blk->sigma >= 1 && blk->sigma <= 16 && env_int("TC_MTF_FORCE_GENERAL", 0) == 0 &&
                 env_int("TC_MTF_WAVE_CHUNKS", 0) == 0 && env_int("TC_DECODE_BYTES", 1) != 0;
    for (u32 i = 0; small && i < blk->sigma; i++)
        small = sorted_list[i] >= -1 && sorted_list[i] <= 255 && (i == 0 || sorted_list[i] > sorted_list[i - 1]);
    const bool small_lf = small && blk->sigma >= 2 && blk->sigma <= 6 && sorted_list[0] == -1 && N > 1 &&
                          env_int("TC_IBWT_LF", 1) != 0;
    auto plan = [&](Arena &A, bool dry) {
        d_idx = A.get<u16>(N + 64);
        d_sym = A.get<i16>(N + 64);
        size_t mark = A.off, hi = mark;
        if (small) {
            // sigma <= 16: indices and codes one byte each all the way (half the traffic of the u16 / i16 forms)
            u8 *d_idx8 = reinterpret_cast<u8 *>(d_idx);
            u8 *d_code8 = reinterpret_cast<u8 *>(d_idx) + ((N + 64 + 15) & ~(u64)15);   // second half of d_idx
            rle_decode_device<u16, u8>(ctx, A, blk->run_count, blk->run_value, blk->nruns, false, d_idx8, N, &got, dry);
            if (!dry && got != N)
                TC_FAIL(ctx, TC_ERR_MALFORMED, "runs expand to %llu symbols, block says %llu",
                        (unsigned long long)got, (unsigned long long)N);
            hi = A.off > hi ? A.off : hi;
            A.off = mark;
            u64 *t_perm = A.get<u64>((size_t)tc_cdiv(N, MTF_TILE) + 2);
            SymTab tab;
            for (u32 v = 0; v < 260; v++) tab.v[v] = v < blk->sigma ? sorted_list[v] : (i16)0;
            bool walked = false;
            if (small_lf) {
                if (!dry) imtf_nib_device<u8, u8>(ctx, t_perm, d_idx8, N, blk->sigma, tab, d_code8);
                hi = A.off > hi ? A.off : hi;
                A.off = mark;
                CodeAcc cacc{d_code8};
                ibwt_device<CodeAcc>(ctx, A, cacc, N, nullptr, d_text, &n_out, dry, sorted_list, blk->sigma);
                walked = dry || n_out != ~0ull;
                if (!walked) {   // not one Nothing / a code outside the list: the symbol path decides
                    u32 g = tc_cdiv(N, 256 * 8);
                    codes_to_syms_kernel<<<g > 8192 ? 8192 : g, 256, 0, ctx->stream>>>(d_code8, N, tab, d_sym);
                    TC_LAUNCH_CHECK(ctx);
                }
            } else if (!dry) {
                imtf_nib_device<u8, i16>(ctx, t_perm, d_idx8, N, blk->sigma, tab, d_sym);
            }
            hi = A.off > hi ? A.off : hi;
            A.off = mark;
            if (!walked || dry) {
                SymAcc sacc{d_sym};
                u64 n2 = 0;
                ibwt_device<SymAcc>(ctx, A, sacc, N, nullptr, d_text, &n2, dry);
                if (!walked) n_out = n2;
            }
            hi = A.off > hi ? A.off : hi;
            A.off = hi;
            return;
        }
        rle_decode_device<u16>(ctx, A, blk->run_count, blk->run_value, blk->nruns, false, d_idx, N,
                               &got, dry);
        if (!dry && got != N)
            TC_FAIL(ctx, TC_ERR_MALFORMED, "runs expand to %llu symbols, block says %llu",
                    (unsigned long long)got, (unsigned long long)N);
        hi = A.off > hi ? A.off : hi;
        A.off = mark;
        mtf_decode_device(ctx, A, d_idx, N, blk->final_list, blk->sigma, d_sym, dry, (i64)blk->primary, d_idx);
        hi = A.off > hi ? A.off : hi;
        A.off = mark;
        SymAcc acc{d_sym};
        // the symbols the stream can hold: the block's MTF list, sorted
        ibwt_device<SymAcc>(ctx, A, acc, N, nullptr, d_text, &n_out, dry, sorted_list, blk->sigma);
        hi = A.off > hi ? A.off : hi;
        A.off = hi;
    };
    Arena dry(nullptr);
    plan(dry, true);
    tc_ws_reserve(ctx, dry.off);
    Arena A(ctx->ws);
    plan(A, false);
    tc_sync_check(ctx);
    if (n_out != n)
        TC_FAIL(ctx, TC_ERR_MALFORMED, "block decodes to %llu bytes, header says %llu",
                (unsigned long long)n_out, (unsigned long long)n);
}
