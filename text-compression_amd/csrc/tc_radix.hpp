// tc_radix.hpp -- device LSD radix sort of (u64 key, u32 value) pairs.
//
// One "onesweep"-style pass = ONE read and ONE write of every pair: global digit
// histograms for all passes are taken up front (or fused into the producer of the
// keys), and each pass resolves a tile's position inside its digit bucket with a
// decoupled look-back over per-tile digit counts while the tile sits in LDS.
// Ranking inside the tile is stable: wave-striped items, match-any by ballots.
//
// Replaces (together with tc_sa.hip) `DS.unstableSortOn snd` over the suffixes,
// reference BWT/Internal.hs:130.
#pragma once
#include <stdlib.h>

#include "tc_common.hpp"

#define RDX_BITS 8
#define RDX_BINS 256
#define RDX_MAX_PASSES 16
// tile shape: 4096 pairs per tile; 256 threads x 16 items measured best on MI355X
// (512x8 9.0, 1024x8 8.7, 512x16 8.6, 256x16 8.5 ms per 2^30-pair pass)
#ifndef RDX_NT
#define RDX_NT 256
#endif
#ifndef RDX_ITEMS
#define RDX_ITEMS 16
#endif
#define RDX_TILE (RDX_NT * RDX_ITEMS)
#ifndef RDX_XCD_RUN
#define RDX_XCD_RUN 8
#endif
#ifndef RDX_MINW
#define RDX_MINW 3
#endif
#define RDX_LB_WAVES(persist) ((persist) ? 4 : RDX_MINW)

struct RadixPlan {
    int npass = 0;
    int shift[RDX_MAX_PASSES];
    u32 mask[RDX_MAX_PASSES];
    static int digit_bits() {  // TC_RADIX_DIGIT_BITS: fan-out experiment knob (default 8)
        const char *e = getenv("TC_RADIX_DIGIT_BITS");
        int v = e && *e ? atoi(e) : RDX_BITS;
        return v < 1 ? 1 : (v > RDX_BITS ? RDX_BITS : v);
    }
    void add_range(int lo_bit, int hi_bit) {  // passes over bits [lo_bit, hi_bit), LSD order
        const int db = digit_bits();
        for (int b = lo_bit; b < hi_bit; b += db) {
            int w = hi_bit - b < db ? hi_bit - b : db;
            shift[npass] = b;
            mask[npass] = (1u << w) - 1u;
            npass++;
        }
    }
};

// first pass of the suffix sort: keys are built from the text inside the pass (no key array
// is ever written or read for it); see keybuild_kernel in tc_sa.hpp for the key layout
struct RadixKeyGen {
    u32 n_text;   // text length; pairs sorted = n_text + 1
    u32 B, w, s, P;
    u16 lut[256];
    // byte -> code without a table read, for alphabets of at most 8 bytes whose values differ in three adjacent bits
    // (ACGTN: bits 1..3): code = byte ((byte >> hsh) & 7) of the 8-byte table (tlo, thi), four bytes per v_perm_b32
    u32 hash_ok, hsh, tlo, thi;
};
// (host) fill the hash fields of a key generator from its lut
static inline void radix_keygen_hash(RadixKeyGen &kg) {
    kg.hash_ok = 0; kg.hsh = 0; kg.tlo = 0; kg.thi = 0;
    int nb = 0, bytes[256];
    for (int v = 0; v < 256; v++)
        if (kg.lut[v]) bytes[nb++] = v;
    if (nb == 0 || nb > 8) return;
    for (u32 sh = 0; sh <= 5; sh++) {
        u32 seen = 0;
        bool ok = true;
        for (int i = 0; i < nb && ok; i++) {
            const u32 h = ((u32)bytes[i] >> sh) & 7u;
            ok = !((seen >> h) & 1u) && kg.lut[bytes[i]] < 256;
            seen |= 1u << h;
        }
        if (!ok) continue;
        unsigned long long t = 0;
        for (int i = 0; i < nb; i++) t |= (unsigned long long)(kg.lut[bytes[i]] & 0xffu) << (8 * (((u32)bytes[i] >> sh) & 7u));
        kg.hash_ok = 1; kg.hsh = sh; kg.tlo = (u32)t; kg.thi = (u32)(t >> 32);
        return;
    }
}

struct RadixPlanDev {
    int npass;
    int shift[RDX_MAX_PASSES];
    u32 mask[RDX_MAX_PASSES];
};

#ifdef __HIPCC__

// ---- up-front digit histograms for every pass (one read of the keys) --------
__global__ __launch_bounds__(256) void radix_hist_kernel(const u64 *__restrict__ keys, u32 n,
                                                         RadixPlanDev plan,
                                                         u32 *__restrict__ hist) {
    __shared__ u32 s_h[RDX_MAX_PASSES * RDX_BINS];
    for (int i = threadIdx.x; i < plan.npass * RDX_BINS; i += 256) s_h[i] = 0;
    __syncthreads();
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) {
        u64 k = keys[i];
        for (int p = 0; p < plan.npass; p++)
            atomicAdd(&s_h[p * RDX_BINS + (u32)((k >> plan.shift[p]) & plan.mask[p])], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < plan.npass * RDX_BINS; i += 256) {
        u32 c = s_h[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

// counts[npass][256] -> exclusive bucket bases, in place.  grid = npass blocks.
__global__ __launch_bounds__(256) void radix_scan_hist_kernel(u32 *hist) {
    __shared__ u32 s[8];
    u32 *h = hist + blockIdx.x * RDX_BINS;
    u32 v = h[threadIdx.x], tot;
    u32 e = block_excl_sum<256>(v, s, &tot);
    h[threadIdx.x] = e;
}

// ---- one pass ---------------------------------------------------------------
// Shipped instance: PERSIST = false (one tile per block, tile id from an atomic ticket: a
// tile only ever waits on tiles whose blocks are already running), LBB = 1 (look-back
// fetches one status word per round trip), SPLIT = false.  The other settings are the
// round-1 experiments (persistent blocks with register prefetch, batched look-back, offsets
// from a pre-scanned matrix); none was faster -- profiles/r01_radix_ablation.txt.
// KEYGEN = true: first pass of the suffix sort, keys built from the text inside the pass.
template <bool GEN_IDX, bool PERSIST, int LBB, bool SPLIT, bool KEYGEN>
__global__ __launch_bounds__(RDX_NT, RDX_LB_WAVES(PERSIST)) void radix_pass_kernel(
    const u64 *__restrict__ kin, const u32 *__restrict__ vin, u64 *__restrict__ kout,
    u32 *__restrict__ vout, u32 n, int shift, u32 mask, const u32 *__restrict__ bucket_base,
    u64 *status, u32 *ticket, u32 *err, const u32 *__restrict__ tile_offs,
    const u8 *__restrict__ text, RadixKeyGen kg) {
    constexpr int NW = RDX_NT / 64;
#ifdef TC_RADIX_DIAG
    const int shift_raw = shift;  // timing-only ablation bits ride in the high bits of `shift`
#else
    const int shift_raw = shift & 0x300000;
#endif
    shift &= 0xff;
    // staging area for the sorted tile; the per-wave histograms overlay its head (they
    // are dead once every item's local position sits in a register): 48 KB + 2 KB => 3
    // blocks per CU
    __shared__ __attribute__((aligned(16))) unsigned char s_raw[RDX_TILE * 12];
    u64 *s_keys = reinterpret_cast<u64 *>(s_raw);
    u32 *s_vals = reinterpret_cast<u32 *>(s_raw + RDX_TILE * 8);
    u32 *s_hist = reinterpret_cast<u32 *>(s_raw);
    u64 *s_mask = reinterpret_cast<u64 *>(s_raw + NW * RDX_BINS * 4);
    static_assert(NW * RDX_BINS * 12 <= RDX_TILE * 12, "histograms + match masks must fit under the staging area");
    __shared__ u32 s_dbase[RDX_BINS];
    __shared__ u32 s_gbase[RDX_BINS];
    __shared__ u16 s_klut[KEYGEN ? 256 : 1];
    __shared__ u32 s_scan[RDX_NT / 64 + 1];
    __shared__ u32 s_tile;

    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    if (KEYGEN)
        for (int i = tid; i < 256; i += RDX_NT) s_klut[i] = kg.lut[i];
    if (SPLIT) {
        if (tid == 0) s_tile = blockIdx.x;
    } else if (shift_raw & 0x200000) {
        // XCD-aware tile order (guide T1): workgroups are dealt round-robin over the 8 XCDs,
        // so blocks with equal blockIdx % 8 share an L2.  Each label x draws tickets from its
        // own counter and takes tiles in runs of 8: ticket a -> tile (a/8)*64 + 8x + a%8.
        // Consecutive tiles append to the same bucket tails, so the partially written lines
        // at the seams merge in ONE L2 instead of being evicted half-filled from two (a
        // read-for-merge plus a partial write each).  A tile may now wait on a tile whose
        // ticket is drawn up to 56 blocks later; the bounded look-back spin + the host's
        // retry with the plain single counter cover a dispatch order that breaks this.
        if (tid == 0) {
            const u32 x = blockIdx.x & 7u;
            const u32 a = atomicAdd(ticket + 32 * x, 1u);
            const u32 RL = RDX_XCD_RUN;                  // tiles per run
            const u32 T = (u32)(((u64)n + RDX_TILE - 1) / RDX_TILE), Gf = T / (8 * RL);
            u32 t;
            if (a < Gf * RL) t = (a / RL) * (8 * RL) + x * RL + (a % RL);
            else t = Gf * (8 * RL) + (a - Gf * RL) * 8 + x;   // tail (< 8*RL tiles): interleaved
            s_tile = t;
        }
    } else {
        if (tid == 0) s_tile = atomicAdd(ticket, 1u);  // one counter: always safe
    }
    __syncthreads();
    const u32 first = s_tile, G = PERSIST ? gridDim.x : 0x7fffffffu;
    const u32 ntiles = (u32)(((u64)n + RDX_TILE - 1) / RDX_TILE);
    const u32 wofs = w * 64 * RDX_ITEMS;
    u32 *wh = s_hist + w * RDX_BINS;

    u64 key[RDX_ITEMS], nkey[RDX_ITEMS];
    u32 val[RDX_ITEMS], nval[RDX_ITEMS];
    auto load_tile = [&](u32 tile, u64 *kk, u32 *vv) {
        const u64 base = (u64)tile * RDX_TILE;
        const u64 *kt = kin + base;  // uniform base + 32-bit lane offset
        const u32 *vt = vin + base;
        const u32 lim = tile < ntiles ? ((n - base) < (u64)RDX_TILE ? (u32)(n - base) : (u32)RDX_TILE) : 0u;
#pragma unroll
        for (int k = 0; k < RDX_ITEMS; k++) {
            const u32 o = wofs + k * 64 + l;
#ifdef TC_RADIX_DIAG
            if (shift_raw & 0x800) {  // DIAGNOSTIC bit3: no global loads
                kk[k] = ((base + o) * 0x9E3779B97F4A7C15ull) | 1ull;
                vv[k] = (u32)(base + o);
                continue;
            }
#endif
            if (o < lim) {
#ifdef RDX_NT_LOADS
                kk[k] = __builtin_nontemporal_load(kt + o);   // read-once stream: keep L2 for the scatter
                vv[k] = GEN_IDX ? (u32)base + o : __builtin_nontemporal_load(vt + o);
#else
                kk[k] = kt[o];
                vv[k] = GEN_IDX ? (u32)base + o : vt[o];
#endif
            } else {
                kk[k] = ~0ull;
                vv[k] = 0;
            }
        }
    };
    if (!KEYGEN) load_tile(first, key, val);

    for (u32 tile = first; tile < ntiles; tile += G) {
        const u64 base = (u64)tile * RDX_TILE;
        const u32 valid = (n - base) < (u64)RDX_TILE ? (u32)(n - base) : (u32)RDX_TILE;
        if (PERSIST) load_tile(tile + G, nkey, nval);  // prefetch; consumed next iteration
        if (KEYGEN) {
            // ---- keys of suffixes [base, base + TILE) straight from the text ---------------
            // LDS image (codes, G values, raw bytes) lives at +16 KB of the staging area,
            // clear of the histograms / match masks at its head
            constexpr int KG_PRE = 16, KG_SLOTS = RDX_TILE + KG_PRE + 80;
            u16 *k_c = reinterpret_cast<u16 *>(s_raw + 16384);
            u16 *k_g = k_c + KG_SLOTS;
            u8 *k_r = reinterpret_cast<u8 *>(k_g + KG_SLOTS);
            static_assert(16384 + KG_SLOTS * 5 <= RDX_TILE * 12, "key-generation image must fit in the staging area");
            const u32 nt = kg.n_text;
            const u32 span = RDX_TILE + kg.P * kg.s + kg.s;
            const u32 units = (KG_PRE + span + 15) / 16;
            const bool aligned = (((uintptr_t)text) & 15) == 0;
            for (u32 u = tid; u < units; u += RDX_NT) {
                const i64 p0 = (i64)base - KG_PRE + (i64)u * 16;
                u8 raw[16];
                if (aligned && p0 >= 0 && p0 + 16 <= (i64)nt) {
                    uint4 v = *reinterpret_cast<const uint4 *>(text + p0);
                    u32 xx[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int q = 0; q < 16; q++) raw[q] = (u8)(xx[q >> 2] >> (8 * (q & 3)));
                } else {
#pragma unroll
                    for (int q = 0; q < 16; q++) {
                        i64 pp = p0 + q;
                        raw[q] = (pp >= 0 && pp < (i64)nt) ? text[pp] : (u8)0;
                    }
                }
                u32 cw[8], rw[4];
                if (p0 >= 0 && p0 + 16 <= (i64)nt) {  // interior unit: no range checks per byte
#pragma unroll
                    for (int q = 0; q < 16; q += 2)
                        cw[q >> 1] = (u32)s_klut[raw[q]] | ((u32)s_klut[raw[q + 1]] << 16);
                } else {
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        i64 pa = p0 + q, pb = p0 + q + 1;
                        u32 ca = (pa >= 0 && pa < (i64)nt) ? (u32)s_klut[raw[q]] : 0u;
                        u32 cb = (pb >= 0 && pb < (i64)nt) ? (u32)s_klut[raw[q + 1]] : 0u;
                        cw[q >> 1] = ca | (cb << 16);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; q++)
                    rw[q] = raw[4 * q] | (raw[4 * q + 1] << 8) | (raw[4 * q + 2] << 16) | ((u32)raw[4 * q + 3] << 24);
                uint4 *dc = reinterpret_cast<uint4 *>(k_c + u * 16);
                dc[0] = make_uint4(cw[0], cw[1], cw[2], cw[3]);
                dc[1] = make_uint4(cw[4], cw[5], cw[6], cw[7]);
                *reinterpret_cast<uint4 *>(k_r + u * 16) = make_uint4(rw[0], rw[1], rw[2], rw[3]);
            }
            __syncthreads();
            const u32 gslots = KG_PRE + RDX_TILE + kg.P * kg.s;
            if (kg.s == 3 && kg.w == 8 && kg.P <= 7) {
                // the DNA-like configurations, unrolled: 3 symbols per 8-bit field, up to 7 fields
                const u32 B = kg.B;
                for (u32 q = tid; q < gslots; q += RDX_NT)
                    k_g[q] = (u16)((k_c[q] * B + k_c[q + 1]) * B + k_c[q + 2]);
                __syncthreads();
#pragma unroll
                for (int k = 0; k < RDX_ITEMS; k++) {
                    const u32 p = wofs + k * 64 + l;
                    if (base + p < n) {
                        const u32 q = KG_PRE + p;
                        u64 kk;
                        if (kg.P == 6) {  // (wave-uniform)
                            const u64 hi = ((u64)k_g[q] << 24) | ((u64)k_g[q + 3] << 16) | ((u64)k_g[q + 6] << 8) | (u64)k_g[q + 9];
                            const u64 lo = ((u64)k_g[q + 12] << 8) | (u64)k_g[q + 15];
                            kk = (hi << 32) | (lo << 16);
                        } else if (kg.P == 7) {
                            const u64 hi = ((u64)k_g[q] << 24) | ((u64)k_g[q + 3] << 16) | ((u64)k_g[q + 6] << 8) | (u64)k_g[q + 9];
                            const u64 lo = ((u64)k_g[q + 12] << 16) | ((u64)k_g[q + 15] << 8) | (u64)k_g[q + 18];
                            kk = (hi << 32) | (lo << 8);
                        } else {
                            kk = 0;
#pragma unroll
                            for (int f = 0; f < 7; f++)
                                if (f < (int)kg.P) kk |= (u64)k_g[q + 3 * f] << (56 - 8 * f);
                        }
                        key[k] = kk | (u64)k_r[q - 1];
                        val[k] = (u32)base + p;
                    } else {
                        key[k] = ~0ull;
                        val[k] = 0;
                    }
                }
            } else {
            for (u32 q = tid; q < gslots; q += RDX_NT) {
                u32 g = 0;
                for (u32 j = 0; j < kg.s; j++) g = g * kg.B + k_c[q + j];
                k_g[q] = (u16)g;
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < RDX_ITEMS; k++) {
                const u32 p = wofs + k * 64 + l;
                if (base + p < n) {
                    const u32 q = KG_PRE + p;
                    u64 kk = 0;
                    int sh = 64;
                    for (u32 f = 0; f < kg.P; f++) {
                        sh -= kg.w;
                        kk |= (u64)k_g[q + f * kg.s] << sh;
                    }
                    key[k] = kk | (u64)k_r[q - 1];
                    val[k] = (u32)base + p;
                } else {
                    key[k] = ~0ull;
                    val[k] = 0;
                }
            }
            }
            __syncthreads();
        }
        if (!(shift_raw & 0x4000))
            for (int i = tid; i < NW * RDX_BINS * 3; i += RDX_NT) s_hist[i] = 0;  // histograms + masks
        __syncthreads();
        // stable ranking inside the wave, item by item.  match-any through LDS: every lane
        // ORs its lane bit into a per-wave, per-digit 64-bit mask, then reads the mask of
        // its own digit back (2 LDS ops instead of 8 ballots + ~50 VALU per item); the
        // first lane of each digit group updates the wave histogram and clears the mask.
        u32 rnk[RDX_ITEMS];
        u32 dig[RDX_ITEMS];
        u64 *wm = s_mask + w * RDX_BINS;
        const u64 mybit = 1ull << l;
#pragma unroll
        for (int k = 0; k < RDX_ITEMS; k++) {
            u32 p = wofs + k * 64 + l;
            u32 d = p < valid ? (u32)((key[k] >> shift) & mask) : 255u;  // pads: last digit, last
            dig[k] = d;
            if (shift_raw & 0x1000) { rnk[k] = k; continue; }  // DIAGNOSTIC bit4
            atomicOr((unsigned long long *)&wm[d], (unsigned long long)mybit);
            __builtin_amdgcn_wave_barrier();
            u64 m = wm[d];
            u32 old = wh[d];
            u32 prior = __popcll(m & (mybit - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (prior == 0) {
                wh[d] = old + __popcll(m);
                wm[d] = 0ull;
            }
            __builtin_amdgcn_wave_barrier();
            rnk[k] = old + prior;
        }
        __syncthreads();
        // digit totals, exclusive over waves; one owner thread per digit
        u32 tot = 0;
        if (tid < RDX_BINS && !(shift_raw & 0x2000)) {  // DIAGNOSTIC bit5 skips
#pragma unroll
            for (int i = 0; i < NW; i++) {
                u32 c = s_hist[i * RDX_BINS + tid];
                s_hist[i * RDX_BINS + tid] = tot;
                tot += c;
            }
        }
        u32 tot_real = tot;
        if (tid == 255) tot_real = tot - (RDX_TILE - valid);
        u64 *st = status + (u64)tile * RDX_BINS + tid;
        if (!SPLIT && tid < RDX_BINS && !(shift_raw & 0x10000))  // DIAGNOSTIC bit8: no status stores
            lb_store(st, (tile == 0 ? LB_FLAG_INC : LB_FLAG_AGG) | (u64)tot_real);
        u32 dtot;
        u32 dbase = block_excl_sum<RDX_NT>(tid < RDX_BINS ? tot : 0u, s_scan, &dtot);
        if (tid < RDX_BINS) s_dbase[tid] = dbase;
        __syncthreads();
        // local sorted position of every item
        u32 pos[RDX_ITEMS];
#pragma unroll
        for (int k = 0; k < RDX_ITEMS; k++) pos[k] = s_dbase[dig[k]] + wh[dig[k]] + rnk[k];
        if (tid < RDX_BINS) {
            u32 excl = 0;
            if (SPLIT) {
                excl = tile_offs[(u64)tile * RDX_BINS + tid];  // already includes bucket base
                s_gbase[tid] = excl - dbase;
            } else {
                if (tile > 0 && !(shift_raw & 0x8000)) {  // DIAGNOSTIC bit7: no look-back
                    i64 t = (i64)tile - 1;
                    u32 spins = 0;
                    bool done = false;
                    while (!done) {
                        u64 sv[LBB];
#pragma unroll
                        for (int i = 0; i < LBB; i++) {
                            i64 ti = t - i;
                            sv[i] = ti >= 0 ? lb_load(status + (u64)ti * RDX_BINS + tid) : LB_FLAG_INC;
                        }
                        int consumed = LBB;
#pragma unroll
                        for (int i = 0; i < LBB; i++) {
                            if (done || consumed != LBB) continue;
                            u32 f = (u32)(sv[i] >> 62);
                            if (f == 0) {  // not published yet: resume the walk from this tile
                                consumed = i;
                                if (++spins > LB_SPIN_LIMIT) {
                                    atomicOr(err, 2u);
                                    done = true;
                                }
                                continue;
                            }
                            excl += (u32)LB_VALUE(sv[i]);
                            if (f == 2) done = true;
                        }
                        t -= consumed;
                        if (consumed == 0) __builtin_amdgcn_s_sleep(1);
                    }
                    if (!(shift_raw & 0x10000)) lb_store(st, LB_FLAG_INC | (u64)(excl + tot_real));
                }
                s_gbase[tid] = bucket_base[tid] + excl - dbase;
            }
        }
        __syncthreads();
        if (shift_raw & 0x200) {  // DIAGNOSTIC bit1: no LDS staging, plain coalesced copy-out
#pragma unroll
            for (int k = 0; k < RDX_ITEMS; k++) {
                u32 p = wofs + k * 64 + l;
                if (p < valid && (pos[k] != 0xffffffffu)) {
                    kout[base + p] = key[k];
                    vout[base + p] = val[k];
                }
            }
        } else {
#pragma unroll
        for (int k = 0; k < RDX_ITEMS; k++) {
            s_keys[pos[k]] = key[k];
            s_vals[pos[k]] = val[k];
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RDX_ITEMS; k++) {
            u32 p = tid + k * RDX_NT;
            if (p < valid) {
                u64 kk = s_keys[p];
                u32 d = (u32)((kk >> shift) & mask);
                u32 g = s_gbase[d] + p;
                if (!(shift_raw & 0x400) || kk == 0x123456789abcdefull) {  // DIAGNOSTIC bit2: no stores
                    kout[g] = kk;
                    vout[g] = s_vals[p];
                }
            }
        }
        }
        if (PERSIST) {
#pragma unroll
            for (int k = 0; k < RDX_ITEMS; k++) {
                key[k] = nkey[k];
                val[k] = nval[k];
            }
            __syncthreads();
        }
    }
}

#endif  // __HIPCC__

// Host driver.  Sorts n pairs by the plan's passes (stable, LSD).  Buffers ping-
// pong; on return keys/vals point at the sorted data and *_alt at scratch.
// `hist` (device, [npass][256] u32): if hist_ready the caller already filled it
// with digit COUNTS for every pass (fused into the key producer); else it is
// computed here.  `status` must hold tiles*256 u64 + 2 words.
struct RadixBuffers {
    u64 *keys, *keys_alt;
    u32 *vals, *vals_alt;
    u32 *hist;    // [RDX_MAX_PASSES][256]
    u64 *status;  // [tiles*256 + 2]
    size_t status_cap = 0;   // words behind `status` (0: unknown).  With room for every pass's status the sort zeroes
                             // them all at once instead of once per pass (a small sort is mostly its launches)
};
static inline size_t radix_status_words(u64 n) {
    size_t t = tc_cdiv(n, RDX_TILE);
    return t * RDX_BINS + t / 8 + 256;  // look-back granules, or (split variant) matrix + offsets + partials
}

void radix_sort_pairs(tc_ctx *ctx, RadixBuffers &b, u32 n, const RadixPlan &plan, bool gen_idx,
                      bool hist_ready, bool timed = false, const u8 *text = nullptr,
                      const RadixKeyGen *keygen = nullptr, bool xcd_group = false);
