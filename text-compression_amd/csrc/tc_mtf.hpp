// tc_mtf.hpp -- move-to-front transform and its inverse on the device.
//
// Replaces seqToMTF (reference MTF/Internal.hs:128-175; a strictly sequential
// findIndexL + deleteAt + cons per symbol) and seqFromMTF (:201-232).
//
// MTF is a scan over list states.  A chunk's effect on ANY incoming list L is
// "its distinct symbols, most recent first, followed by the rest of L in order"
// so a chunk is summarised by its recency list, and summaries compose
// associatively (apply the right chunk's recency list, least recent first, to the
// left one).  Three launches: per-chunk summaries -> scan of summaries -> replay of
// every chunk from its true incoming list.
//   sigma <= 16: the list is 16 nibbles in one 64-bit register, one chunk per
//                lane (ACGTN + '$' has sigma = 6).
//   sigma <= 257: the list lives across the lanes of a wave, one chunk per wave,
//                find by ballot, shift by lane rotate.
#pragma once
#include "tc_common.hpp"

// ---- symbol accessors (-1 = Nothing) -------------------------------------------
struct BwtAcc {  // (L, primary): the shape bytestringToBWT returns
    const u8 *L;
    i64 primary;
    int dup = 0;  // lane-chunk staging only: the primary slot repeats the symbol before it (see "sigma = 257")
    __device__ __forceinline__ int operator()(u64 j) const {
        return (i64)j == primary ? -1 : (int)L[j];
    }
};
struct SymAcc {  // Seq (Maybe Word8) as int16
    const i16 *s;
    __device__ __forceinline__ int operator()(u64 j) const { return (int)s[j]; }
};
struct U16Acc {  // plain integers (MTF index stream); never Nothing
    const u16 *v;
    __device__ __forceinline__ int operator()(u64 j) const { return (int)v[j]; }
};
struct U8Acc {   // MTF index stream of a small alphabet (fused encode: ranks < 16), one byte each
    const u8 *v;
    __device__ __forceinline__ int operator()(u64 j) const { return (int)v[j]; }
};
// value of a staged int16 slot as the accessor would have returned it
template <class Acc>
__device__ __forceinline__ int staged_value(i16 raw) { return (int)raw; }
template <>
__device__ __forceinline__ int staged_value<U16Acc>(i16 raw) { return (int)(u16)raw; }

#define MTF_NT 256
#ifndef MTF_CH
#define MTF_CH 128                      // symbols per lane chunk (nibble path); 1 GiB ACGTN: 64: 1.71 ms, 128: 1.43, 192: 1.47, 256: 1.69
#endif
#define MTF_TILE (MTF_NT * MTF_CH)      // 32768
#define MTF_STRIDE (MTF_CH + 4)         // LDS chunk stride in bytes: 17 dwords, conflict-free
#define MTFG_CH 4096                    // symbols per wave chunk (general path), multiple of 64

struct Lut8 { u8 v[260]; };     // index = sym + 1 -> code
struct Lut16 { u16 v[260]; };
struct SymTab { i16 v[260]; };  // code -> sym

#ifdef __HIPCC__

// Stage COUNT symbols starting at position `base` into LDS as int16 (-1 = Nothing, -2 =
// past the end) with the widest loads the source allows (narrow per-lane loads run at a
// fraction of the HBM rate on gfx950: 1 B/lane ~1.2 TB/s, 2 B ~2.2, 8-16 B ~5.5).
template <int COUNT, int NT>
__device__ __forceinline__ void stage_syms(const BwtAcc &acc, u64 base, u64 N, i16 *dst) {
    const u8 *src = acc.L + base;
    if ((((uintptr_t)src) & 15) == 0) {
        for (int c = threadIdx.x; c < COUNT / 16; c += NT) {
            u64 p = base + (u64)c * 16;
            if (p + 16 <= N) {
                uint4 v = *reinterpret_cast<const uint4 *>(src + (u64)c * 16);
                u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 16; q++) dst[c * 16 + q] = (i16)((x[q >> 2] >> (8 * (q & 3))) & 0xff);
            } else {
                for (int q = 0; q < 16; q++) dst[c * 16 + q] = (p + q < N) ? (i16)src[(u64)c * 16 + q] : (i16)-2;
            }
        }
    } else {
        for (int c = threadIdx.x; c < COUNT; c += NT) dst[c] = (base + c < N) ? (i16)src[c] : (i16)-2;
    }
    __syncthreads();
    if (acc.primary >= (i64)base && acc.primary < (i64)(base + COUNT) && threadIdx.x == 0)
        dst[acc.primary - (i64)base] = -1;
    __syncthreads();
}
template <int COUNT, int NT, class T16>
__device__ __forceinline__ void stage_syms16(const T16 *srcbase, u64 base, u64 N, i16 *dst) {
    const T16 *src = srcbase + base;
    if ((((uintptr_t)src) & 15) == 0) {
        for (int c = threadIdx.x; c < COUNT / 8; c += NT) {
            u64 p = base + (u64)c * 8;
            if (p + 8 <= N) {
                uint4 v = *reinterpret_cast<const uint4 *>(src + (u64)c * 8);
                *reinterpret_cast<uint4 *>(dst + c * 8) = v;
            } else {
                for (int q = 0; q < 8; q++) dst[c * 8 + q] = (p + q < N) ? (i16)src[(u64)c * 8 + q] : (i16)-2;
            }
        }
    } else {
        for (int c = threadIdx.x; c < COUNT; c += NT) dst[c] = (base + c < N) ? (i16)src[c] : (i16)-2;
    }
    __syncthreads();
}
template <int COUNT, int NT>
__device__ __forceinline__ void stage_syms(const SymAcc &acc, u64 base, u64 N, i16 *dst) {
    stage_syms16<COUNT, NT, i16>(acc.s, base, N, dst);
}
template <int COUNT, int NT>
__device__ __forceinline__ void stage_syms(const U16Acc &acc, u64 base, u64 N, i16 *dst) {
    // raw 16-bit image; read back through staged_value<U16Acc> (unsigned)
    stage_syms16<COUNT, NT, u16>(acc.v, base, N, dst);
}

// presence histogram over an accessor (standalone MTF / RLE / FM entry points)
template <class Acc>
__global__ __launch_bounds__(256) void sym_hist_kernel(Acc acc, u64 N, u32 *__restrict__ counts) {
    __shared__ u32 s_h[257];
    for (int i = threadIdx.x; i < 257; i += 256) s_h[i] = 0;
    __syncthreads();
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < N; j += (u64)gridDim.x * 256)
        atomicAdd(&s_h[acc(j) + 1], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < 257; i += 256)
        if (s_h[i]) atomicAdd(&counts[i], s_h[i]);
}

// ================================ nibble path ==================================
struct NibSumm {
    u64 perm;  // list, front = low nibble
    u32 mask;  // codes seen
};
#define NIB_IDENT 0xFEDCBA9876543210ull

__device__ __forceinline__ u32 nib_find(u64 list, u32 c) {
    u64 x = list ^ (0x1111111111111111ull * c);
    u64 t = (x - 0x1111111111111111ull) & ~x & 0x8888888888888888ull;
    return (u32)__builtin_ctzll(t) >> 2;
}
__device__ __forceinline__ u64 nib_front(u64 list, u32 pos, u32 c) {
    u64 lowmask = (1ull << (4 * pos)) - 1ull;      // nibbles [0, pos)
    u64 upto = (2ull << (4 * pos + 3)) - 1ull;     // nibbles [0, pos]
    return (list & ~upto) | ((list & lowmask) << 4) | (u64)c;
}
// The same two steps for an alphabet of at most 8 codes (a DNA record: 5 letters + the sentinel): the list is the low
// 8 nibbles, ONE 32-bit register -- half the instructions of the 64-bit forms (no 64-bit shifts, subtracts, ctz).
__device__ __forceinline__ u32 nib8_find(u32 list, u32 c) {
    const u32 x = list ^ (0x11111111u * c);
    const u32 t = (x - 0x11111111u) & ~x & 0x88888888u;
    return (u32)__builtin_ctz(t) >> 2;
}
__device__ __forceinline__ u32 nib8_front(u32 list, u32 pos, u32 c) {
    const u32 lowmask = (1u << (4 * pos)) - 1u;    // nibbles [0, pos)
    const u32 upto = (lowmask << 4) | 15u;         // nibbles [0, pos]
    return (list & ~upto) | ((list & lowmask) << 4) | c;
}
// a then b
__device__ __forceinline__ NibSumm nib_combine(NibSumm a, NibSumm b) {
    int d = __popc(b.mask);
    u64 perm = a.perm;
    for (int i = d - 1; i >= 0; i--) {
        u32 c = (u32)(b.perm >> (4 * i)) & 15u;
        perm = nib_front(perm, nib_find(perm, c), c);
    }
    return NibSumm{perm, a.mask | b.mask};
}
// a then b for an alphabet of at most 8 codes: both lists keep the codes 0..7 in their low 8 nibbles (codes that are not
// part of the alphabet stay behind the live ones), so the composition runs on 32 bits -- half the instructions
__device__ __forceinline__ NibSumm nib8_combine(NibSumm a, NibSumm b) {
    const int d = __popc(b.mask);
    u32 perm = (u32)a.perm;
    const u32 bp = (u32)b.perm;
    for (int i = d - 1; i >= 0; i--) {
        const u32 c = (bp >> (4 * i)) & 15u;
        perm = nib8_front(perm, nib8_find(perm, c), c);
    }
    return NibSumm{(a.perm & 0xFFFFFFFF00000000ull) | (u64)perm, a.mask | b.mask};
}
template <bool S8>
__device__ __forceinline__ NibSumm nib_combine_t(NibSumm a, NibSumm b) {
    return S8 ? nib8_combine(a, b) : nib_combine(a, b);
}
__device__ __forceinline__ NibSumm nib_shfl_up(NibSumm v, int d) {
    NibSumm r;
    r.perm = __shfl_up(v.perm, d, 64);
    r.mask = __shfl_up(v.mask, d, 64);
    return r;
}

// Stage one tile of codes into LDS (chunk-major, padded) and return this lane's
// chunk summary (from the identity list).  0xFF marks padding past N.
template <class Acc>
__device__ __forceinline__ void nib_stage(Acc acc, u64 N, u64 base, const u8 *s_lut, u8 *s_code) {
    for (u32 p = threadIdx.x; p < MTF_TILE; p += MTF_NT) {
        u64 j = base + p;
        u8 c = 0xFF;
        if (j < N) c = s_lut[acc(j) + 1];
        s_code[(p / MTF_CH) * MTF_STRIDE + (p % MTF_CH)] = c;
    }
}
// (L, primary): 16 bytes per lane per load, codes written as 4 dwords into the padded image
struct NibNoHook { __device__ __forceinline__ void operator()() const {} };
// `mid` runs between a tile's loads and their conversion (its own memory latency then overlaps theirs)
template <class Mid>
__device__ __forceinline__ void nib_stage_hook(BwtAcc acc, u64 N, u64 base, const u8 *s_lut, u8 *s_code, Mid mid) {
    const u8 *src = acc.L + base;
    if ((((uintptr_t)src) & 15) == 0 && base + MTF_TILE <= N) {
        // a tile inside the record: ALL of a thread's loads first, then the conversions (a load per loop turn, each
        // waited for before the next is issued, was 8 memory latencies in a row: 12 of a tile's ~34 us)
        constexpr int NL = MTF_TILE / 16 / MTF_NT;
        uint4 v[NL];
#pragma unroll
        for (int i = 0; i < NL; i++) v[i] = reinterpret_cast<const uint4 *>(src)[i * MTF_NT + threadIdx.x];
        mid();
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const u32 p0 = (u32)(i * MTF_NT + threadIdx.x) * 16;
            const u32 x[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
            u32 out[4];
#pragma unroll
            for (int q = 0; q < 4; q++)
                out[q] = (u32)s_lut[(x[q] & 255) + 1] | ((u32)s_lut[((x[q] >> 8) & 255) + 1] << 8) |
                         ((u32)s_lut[((x[q] >> 16) & 255) + 1] << 16) | ((u32)s_lut[(x[q] >> 24) + 1] << 24);
            u32 *dst = reinterpret_cast<u32 *>(s_code + (p0 / MTF_CH) * MTF_STRIDE + (p0 % MTF_CH));
            dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
        }
    } else if ((((uintptr_t)src) & 15) == 0) {
        mid();
        for (u32 c = threadIdx.x; c < MTF_TILE / 16; c += MTF_NT) {
            const u32 p0 = c * 16;
            const u64 j0 = base + p0;
            u32 out[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
            if (j0 + 16 <= N) {
                uint4 v = *reinterpret_cast<const uint4 *>(src + p0);
                u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; q++)
                    out[q] = (u32)s_lut[(x[q] & 255) + 1] | ((u32)s_lut[((x[q] >> 8) & 255) + 1] << 8) |
                             ((u32)s_lut[((x[q] >> 16) & 255) + 1] << 16) | ((u32)s_lut[(x[q] >> 24) + 1] << 24);
            } else {
                for (int q = 0; q < 16; q++)
                    if (j0 + q < N) out[q >> 2] = (out[q >> 2] & ~(0xFFu << (8 * (q & 3)))) | ((u32)s_lut[(u32)src[p0 + q] + 1] << (8 * (q & 3)));
            }
            u32 *dst = reinterpret_cast<u32 *>(s_code + (p0 / MTF_CH) * MTF_STRIDE + (p0 % MTF_CH));
            dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
        }
    } else {
        mid();
        for (u32 p = threadIdx.x; p < MTF_TILE; p += MTF_NT) {
            u64 j = base + p;
            s_code[(p / MTF_CH) * MTF_STRIDE + (p % MTF_CH)] = j < N ? s_lut[(u32)src[p] + 1] : (u8)0xFF;
        }
    }
    // the sentinel slot carries byte 0 in L: patch its code
    if (acc.primary >= (i64)base && acc.primary < (i64)(base + MTF_TILE)) {
        __syncthreads();
        if (threadIdx.x == 0) {
            u32 p = (u32)(acc.primary - (i64)base);
            s_code[(p / MTF_CH) * MTF_STRIDE + (p % MTF_CH)] = s_lut[0];
        }
    }
}

template <>
__device__ __forceinline__ void nib_stage<BwtAcc>(BwtAcc acc, u64 N, u64 base, const u8 *s_lut, u8 *s_code) {
    nib_stage_hook(acc, N, base, s_lut, s_code, NibNoHook{});
}

__device__ __forceinline__ NibSumm nib_chunk_summary(const u8 *s_code) {
    const u32 *cw = reinterpret_cast<const u32 *>(s_code + threadIdx.x * MTF_STRIDE);
    NibSumm s{NIB_IDENT, 0u};
#pragma unroll 4
    for (int q = 0; q < MTF_CH / 4; q++) {
        u32 wv = cw[q];
#pragma unroll
        for (int b = 0; b < 4; b++) {
            u32 c = (wv >> (8 * b)) & 0xff;
            if (c != 0xFF) {
                s.perm = nib_front(s.perm, nib_find(s.perm, c), c);
                s.mask |= 1u << c;
            }
        }
    }
    return s;
}

// inclusive block scan of summaries; returns this thread's EXCLUSIVE prefix and
// the block aggregate.  s_w needs MTF_NT/64 entries.
template <bool S8 = false>
__device__ __forceinline__ NibSumm nib_block_excl(NibSumm mine, NibSumm *s_w, NibSumm *agg) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    NibSumm inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        NibSumm t = nib_shfl_up(inc, d);
        if (l >= d) inc = nib_combine_t<S8>(t, inc);
    }
    NibSumm exc = nib_shfl_up(inc, 1);
    if (l == 0) exc = NibSumm{NIB_IDENT, 0u};
    if (l == 63) s_w[w] = inc;
    __syncthreads();
    NibSumm pre{NIB_IDENT, 0u}, tot{NIB_IDENT, 0u};
    for (int i = 0; i < MTF_NT / 64; i++) {
        if (i < w) pre = nib_combine_t<S8>(pre, s_w[i]);
        tot = nib_combine_t<S8>(tot, s_w[i]);
    }
    *agg = tot;
    return nib_combine_t<S8>(pre, exc);
}


// MTF list (codes, front = low nibble) just before position `pos`, recovered by scanning
// BACKWARDS: the list is "symbols by last occurrence, most recent first, then the
// never-seen ones in alphabet order".  One wave walks back 64 positions at a time until
// at most one of the sigma codes is still unseen (its place is then forced: last).  On
// high-entropy text that is one or two steps.  Returns false (every lane) if the window
// of MTF_BACK_MAX positions did not settle it -- the caller then flags the slow path.
#ifndef MTF_BACK_MAX
#define MTF_BACK_MAX 8192
#endif
#ifndef MTF_BACK_FAR
#define MTF_BACK_FAR (1u << 19)
#endif
template <class Acc>
__device__ __forceinline__ bool nib_list_before(Acc acc, u64 pos, u32 sigma, const u8 *s_lut, u64 *out, const u32 *giveup = nullptr) {
    u32 seen = 0;          // bit c set: code c already placed
    u64 list = 0;          // placed codes, most recent first
    u32 placed = 0;
    u64 p = pos;
    const u32 all = (sigma >= 32 ? 0xffffffffu : ((1u << sigma) - 1u));
    auto step64 = [&]() {
        const u64 lo = p >= 64 ? p - 64 : 0;                 // chunk [lo, p)
        const u64 j = lo + lane_id();
        const u32 c = j < p ? (u32)s_lut[acc(j) + 1] : 0xffu;
        // walk this chunk from its most recent position down: take the unseen code with the
        // highest position, repeat (at most sigma times)
        u32 todo = all & ~seen;
        while (todo) {
            u64 best = 0;   // ballot of lanes holding a not-yet-placed code
            best = __ballot(c < 16u && ((todo >> c) & 1u));
            if (!best) break;
            const int hl = 63 - __builtin_clzll(best);
            const u32 cc = (u32)__shfl((int)c, hl, 64);
            list |= (u64)cc << (4 * placed);
            placed++;
            seen |= 1u << cc;
            todo &= ~(1u << cc);
        }
        p = lo;
    };
    for (u32 step = 0; step < MTF_BACK_MAX / 64 && p > 0 && __popc(all & ~seen) > 1; step++) step64();
    // Still open after MTF_BACK_MAX positions: the column has long runs here (repeat-rich or periodic text: a poly-A tract
    // puts tens of thousands of equal symbols side by side).  Such stretches are SKIPPED 1024 positions at a time -- a
    // lane looks at 16 bytes and only says whether any of them is a code still missing -- and only a block that holds
    // one is walked 64 positions at a time.  (Round 4: repeat-rich DNA used to fail here and rerun MTF by the
    // three-kernel path: 5.4 instead of 1.9 ms per GiB.)  A last column with its byte array only; bounded by
    // MTF_BACK_FAR positions (a text of one letter never gets here: one missing code ends the walk).  The bound is
    // what a tile may spend: on a text of period 4096 (runs of n / 4096 equal symbols in the last column) every tile of
    // the 1 GiB record walked back over a million positions, 256 at a time -- 218 ms of MTF; now the walk gives up after
    // 2^19 positions, at once when another tile already has (`giveup`: the caller's flag -- the kernel's result is void
    // then), and the record takes the three-kernel path.
    if constexpr (std::is_same<Acc, BwtAcc>::value) {
        for (u32 far = 0; far < MTF_BACK_FAR / 1024 && p >= 1024 && (p & 15) == 0 && __popc(all & ~seen) > 1; far++) {
            if (giveup && (far & 15) == 0 && __atomic_load_n(giveup, __ATOMIC_RELAXED) != 0u) return false;
            const u64 lo = p - 1024;
            const bool has_primary = acc.primary >= (i64)lo && acc.primary < (i64)p;
            bool any = true;
            if (!has_primary && (((uintptr_t)acc.L) & 15) == 0) {
                const uint4 x4 = *reinterpret_cast<const uint4 *>(acc.L + lo + 16 * lane_id());
                const u32 xs[4] = {x4.x, x4.y, x4.z, x4.w};
                const u32 todo = all & ~seen;
                u32 hit = 0;
#pragma unroll
                for (int q = 0; q < 16; q++) {
                    const u32 c = (u32)s_lut[((xs[q >> 2] >> (8 * (q & 3))) & 255u) + 1u];
                    hit |= c < 16u ? (todo >> c) & 1u : 0u;
                }
                any = __ballot(hit != 0) != 0ull;
            }
            if (!any) {
                p = lo;
                continue;
            }
            for (int sub = 0; sub < 16 && __popc(all & ~seen) > 1; sub++) step64();
        }
    }
    u32 rest = all & ~seen;
    if (__popc(rest) > 1 && p > 0) return false;   // window exhausted, still ambiguous
    // never-seen codes follow in alphabet order (exactly right when p == 0 was reached)
    while (rest) {
        u32 cc = (u32)__builtin_ctz(rest);
        list |= (u64)cc << (4 * placed);
        placed++;
        rest &= rest - 1;
    }
    // codes >= sigma are not part of the alphabet: park them behind (identity order)
    for (u32 cc = sigma; cc < 16; cc++) {
        list |= (u64)cc << (4 * placed);
        placed++;
    }
    *out = list;
    return true;
}

template <class Acc>
__global__ __launch_bounds__(MTF_NT) void mtf_nib_summary_kernel(Acc acc, u64 N,
                                                                  Lut8 lut,
                                                                  u64 *__restrict__ t_perm,
                                                                  u32 *__restrict__ t_mask) {
    __shared__ __attribute__((aligned(16))) u8 s_code[MTF_NT * MTF_STRIDE];
    __shared__ u8 s_lut[260];
    __shared__ NibSumm s_w[MTF_NT / 64];
    for (int i = threadIdx.x; i < 257; i += MTF_NT) s_lut[i] = lut.v[i];
    __syncthreads();
    nib_stage(acc, N, (u64)blockIdx.x * MTF_TILE, s_lut, s_code);
    __syncthreads();
    NibSumm mine = nib_chunk_summary(s_code);
    NibSumm agg;
    (void)nib_block_excl(mine, s_w, &agg);
    if (threadIdx.x == 0) {
        t_perm[blockIdx.x] = agg.perm;
        t_mask[blockIdx.x] = agg.mask;
    }
}

// exclusive scan over the tile summaries, in place; one block.  Element `tiles`
// receives the grand total (the final list).
__global__ __launch_bounds__(MTF_NT) void mtf_nib_scan_kernel(u64 *t_perm, u32 *t_mask, u32 tiles) {
    __shared__ NibSumm s_w[MTF_NT / 64];
    const u32 per = (tiles + MTF_NT - 1) / MTF_NT;
    const u32 lo = threadIdx.x * per, hi = lo + per < tiles ? lo + per : tiles;
    NibSumm mine{NIB_IDENT, 0u};
    for (u32 t = lo; t < hi; t++) mine = nib_combine(mine, NibSumm{t_perm[t], t_mask[t]});
    NibSumm agg;
    NibSumm run = nib_block_excl(mine, s_w, &agg);
    for (u32 t = lo; t < hi; t++) {
        NibSumm cur{t_perm[t], t_mask[t]};
        t_perm[t] = run.perm;
        t_mask[t] = run.mask;
        run = nib_combine(run, cur);
    }
    if (threadIdx.x == 0) {
        t_perm[tiles] = agg.perm;
        t_mask[tiles] = agg.mask;
    }
}

// One pass over a lane's chunk from the identity list, sigma <= 8.  The image holds code * 0x11 per symbol (0xFF: past
// the end), so v_perm of the loaded word is the code in all eight nibbles; the list is one 32-bit register; a first
// occurrence is a rank not below the number of codes met so far (the unmet ones keep their order behind the met ones).
// The ranks go back IN PLACE as nibbles (symbol i of the chunk: nibble i & 7 of word i >> 3) -- half the image, and
// the form the run detection below wants.  evc / evp: code and chunk position of the first occurrences, k4 = 4 * count.
template <bool PADS>
__device__ __forceinline__ void nib8_chunk_ranks(u32 *cw, u32 &lst, u32 &seen, u32 &evc, u64 &evp, u32 &k4) {
#pragma unroll 2
    for (int q = 0; q < MTF_CH / 8; q++) {
        const u32 w0 = cw[2 * q], w1 = cw[2 * q + 1];
        u32 nw = 0;
#pragma unroll
        for (int b = 0; b < 8; b++) {
            const u32 wv = b < 4 ? w0 : w1;
            if (PADS && ((wv >> (8 * (b & 3))) & 0xffu) == 0xffu) continue;
            const u32 bc = __builtin_amdgcn_perm(wv, wv, 0x01010101u * (u32)(b & 3));
            const u32 x = bc ^ lst;
            const u32 t = (x - 0x11111111u) & ~x & 0x88888888u;
            const u32 f = (u32)__builtin_ctz(t);           // 4 * rank + 3
            const u32 sh = f & 28u;
            if (f > k4) {
                const u32 c = bc & 7u;
                seen |= 1u << c;
                evc |= c << k4;
                evp |= (u64)(8 * q + b) << (2 * k4);
                k4 += 4;
            }
            const u32 hm = 0xFFFFFFF0u << sh;              // the nibbles behind the rank stay
            const u32 al = __builtin_amdgcn_alignbit(lst, bc, 28);   // (list << 4) | code
            lst = (lst & hm) | (al & ~hm);
            nw |= b == 0 ? sh >> 2 : sh << (4 * b - 2);
        }
        cw[q] = nw;
    }
}

// FASTIN: the tile's incoming list is recovered in-kernel by nib_list_before (no summary /
// scan launches); `flag` is raised when that fails and the host reruns the 3-kernel path.
// SMALL: sigma <= 8 -- the pass over the chunk keeps its list in 32 bits (nib8_find / nib8_front)
template <class Acc, bool FASTIN, class OT = u16, bool SMALL = false>
__global__ __launch_bounds__(MTF_NT) void mtf_nib_apply_kernel(Acc acc, u64 N,
                                                                Lut8 lut,
                                                                const u64 *__restrict__ t_perm,
                                                                OT *__restrict__ idx, u32 sigma,
                                                                u32 *flag) {
    __shared__ __attribute__((aligned(16))) u8 s_code[MTF_NT * MTF_STRIDE];
    __shared__ u8 s_lut[260];
    __shared__ NibSumm s_w[MTF_NT / 64];
    __shared__ u8 s_lut11[SMALL ? 260 : 4];   // SMALL: the image holds code * 0x11 (nib8_chunk_ranks)
    for (int i = threadIdx.x; i < 257; i += MTF_NT) {
        s_lut[i] = lut.v[i];
        if (SMALL) s_lut11[i] = (u8)((lut.v[i] & 7u) * 0x11u);
    }
    __syncthreads();
    const u64 base = (u64)blockIdx.x * MTF_TILE;
    // true incoming list of this lane's chunk: tile's incoming list, then the block-local prefix applied to it.
    // Wave 0 recovers the tile's list (a global round trip or two) while the tile's own loads are in flight (a last
    // column: between its loads and their conversion); the barriers publish it.
    __shared__ u64 s_in;
    auto list_in = [&]() {
        if (FASTIN && threadIdx.x < 64) {
            u64 l0 = NIB_IDENT;
            bool ok = nib_list_before(acc, base, sigma, s_lut, &l0, flag);
            if (threadIdx.x == 0) {
                s_in = l0;
                if (!ok) atomicOr(flag, 1u);
            }
        }
    };
    if constexpr (std::is_same<Acc, BwtAcc>::value) {
        nib_stage_hook(acc, N, base, SMALL ? s_lut11 : s_lut, s_code, list_in);
        __syncthreads();
    } else {
        nib_stage(acc, N, base, SMALL ? s_lut11 : s_lut, s_code);
        __syncthreads();
        list_in();
    }
    // ONE full pass per chunk, from the identity list.  The rank of a symbol that already occurred
    // in the chunk does not depend on the incoming list (everything in front of it was used since),
    // so it is final; only the FIRST occurrence of each code (at most sigma per chunk) needs the
    // true list -- those are recorded (code, position) and replayed below, a handful of steps
    // instead of a second pass over all 64 symbols.
    u32 *cw = reinterpret_cast<u32 *>(s_code + threadIdx.x * MTF_STRIDE);
    NibSumm mine{NIB_IDENT, 0u};
    u64 ev_code = 0, ev_pos0 = 0, ev_pos1 = 0;
    u32 nev = 0;
    if (SMALL) {
        // (the ranks come back as NIBBLES, symbol i of the chunk in nibble i & 7 of word i >> 3)
        u32 lst = (u32)NIB_IDENT, seen = 0, evc = 0, k4 = 0;
        if (base + MTF_TILE >= N) nib8_chunk_ranks<true>(cw, lst, seen, evc, ev_pos0, k4);
        else nib8_chunk_ranks<false>(cw, lst, seen, evc, ev_pos0, k4);
        mine.perm = (NIB_IDENT & 0xFFFFFFFF00000000ull) | (u64)lst;
        mine.mask = seen;
        ev_code = evc;
        nev = k4 >> 2;
    } else {
#pragma unroll 4
    for (int q = 0; q < MTF_CH / 4; q++) {
        u32 wv = cw[q], ov = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const u32 c = (wv >> (8 * b)) & 0xff;
            if (c != 0xFF) {
                const u32 pos = nib_find(mine.perm, c);
                mine.perm = nib_front(mine.perm, pos, c);
                ov |= pos << (8 * b);
                if (!((mine.mask >> c) & 1u)) {
                    mine.mask |= 1u << c;
                    ev_code |= (u64)c << (4 * nev);
                    const u64 pp = (u64)(4 * q + b);
                    if (nev < 8) ev_pos0 |= pp << (8 * nev);
                    else ev_pos1 |= pp << (8 * (nev - 8));
                    nev++;
                }
            }
        }
        cw[q] = ov;  // ranks overwrite the codes in place
    }
    }
    NibSumm agg;
    NibSumm exc = nib_block_excl<SMALL>(mine, s_w, &agg);
    NibSumm in = nib_combine_t<SMALL>(NibSumm{FASTIN ? s_in : t_perm[blockIdx.x], 0u}, exc);
    {
        u64 list = in.perm;
        u8 *cb = s_code + threadIdx.x * MTF_STRIDE;
        for (u32 e = 0; e < nev; e++) {
            const u32 c = (u32)(ev_code >> (4 * e)) & 15u;
            const u32 pp = (u32)((e < 8 ? ev_pos0 >> (8 * e) : ev_pos1 >> (8 * (e - 8))) & 255u);
            const u32 pos = nib_find(list, c);
            list = nib_front(list, pos, c);
            if (SMALL) {
                const u32 shf = 4u * (pp & 7u);
                cw[pp >> 3] = (cw[pp >> 3] & ~(15u << shf)) | (pos << shf);
            } else {
                cb[pp] = (u8)pos;
            }
        }
    }
    __syncthreads();
    if constexpr (SMALL) {
        // the nibble image back to one byte per rank: 16 ranks (two words of the image) per store
        static_assert(sizeof(OT) == 1, "the 32-bit list is for the byte-wide index stream");
        auto spread = [](u32 h) -> u32 {   // four nibbles (bits 0..15) -> four bytes
            u32 x = (h | (h << 8)) & 0x00FF00FFu;
            return (x | (x << 4)) & 0x0F0F0F0Fu;
        };
        const bool al = (((uintptr_t)(idx + base)) & 15) == 0;
        for (u32 g = threadIdx.x; g < MTF_TILE / 16; g += MTF_NT) {
            const u32 p = 16 * g;
            const u32 *sc = reinterpret_cast<const u32 *>(s_code + (p / MTF_CH) * MTF_STRIDE + (p % MTF_CH) / 2);
            const u32 n0 = sc[0], n1 = sc[1];
            const uint4 o = make_uint4(spread(n0 & 0xffffu), spread(n0 >> 16), spread(n1 & 0xffffu), spread(n1 >> 16));
            if (al && base + p + 16 <= N) {
                reinterpret_cast<uint4 *>(idx + base)[g] = o;
            } else {
                const u32 ow[4] = {o.x, o.y, o.z, o.w};
                for (u32 q = 0; q < 16; q++)
                    if (base + p + q < N) idx[base + p + q] = (OT)((ow[q >> 2] >> (8 * (q & 3))) & 0xffu);
            }
        }
        return;
    }
    if constexpr (sizeof(OT) == 1) {
        // byte stream (fused encode): 16 ranks per store, straight from the LDS image
        if ((((uintptr_t)(idx + base)) & 15) == 0) {
            uint4 *o = reinterpret_cast<uint4 *>(idx + base);
            for (u32 g = threadIdx.x; g < MTF_TILE / 16; g += MTF_NT) {
                const u32 p = 16 * g;
                const u32 *sc = reinterpret_cast<const u32 *>(s_code + (p / MTF_CH) * MTF_STRIDE + (p % MTF_CH));
                if (base + p + 16 <= N) {
                    o[g] = make_uint4(sc[0], sc[1], sc[2], sc[3]);
                } else {
                    const u8 *sb = reinterpret_cast<const u8 *>(sc);
                    for (u32 q = 0; q < 16; q++)
                        if (base + p + q < N) idx[base + p + q] = (OT)sb[q];
                }
            }
        } else {
            for (u32 p = threadIdx.x; p < MTF_TILE; p += MTF_NT)
                if (base + p < N) idx[base + p] = (OT)s_code[(p / MTF_CH) * MTF_STRIDE + (p % MTF_CH)];
        }
        return;
    }
    if ((((uintptr_t)(idx + base)) & 3) == 0) {
        u32 *o32 = reinterpret_cast<u32 *>(idx + base);
        for (u32 c = threadIdx.x; c < MTF_TILE / 2; c += MTF_NT) {
            const u32 p = 2 * c;
            const u8 *sc = s_code + (p / MTF_CH) * MTF_STRIDE + (p % MTF_CH);
            if (base + p + 2 <= N) o32[c] = (u32)sc[0] | ((u32)sc[1] << 16);
            else if (base + p < N) idx[base + p] = (OT)sc[0];
        }
    } else {
        for (u32 p = threadIdx.x; p < MTF_TILE; p += MTF_NT)
            if (base + p < N) idx[base + p] = (OT)s_code[(p / MTF_CH) * MTF_STRIDE + (p % MTF_CH)];
    }
}

// list after the last symbol (the MTF result's second component), FASTIN path
template <class Acc>
__global__ __launch_bounds__(64) void mtf_nib_final_kernel(Acc acc, u64 N, Lut8 lut, u32 sigma,
                                                           u64 *out, u32 *flag) {
    __shared__ u8 s_lut[260];
    for (int i = threadIdx.x; i < 257; i += 64) s_lut[i] = lut.v[i];
    __syncthreads();
    u64 l0 = NIB_IDENT;
    bool ok = nib_list_before(acc, N, sigma, s_lut, &l0);
    if (threadIdx.x == 0) {
        *out = l0;
        if (!ok) atomicOr(flag, 1u);
    }
}

// ================================ general path ==================================
// List position q = r*64 + lane lives in row[r] of lane `lane`.  ROWS = ceil(sigma/64).
template <int ROWS>
struct WaveList {
    u32 row[ROWS];
    __device__ __forceinline__ void init_identity() {
#pragma unroll
        for (int r = 0; r < ROWS; r++) row[r] = r * 64 + lane_id();
    }
    // move code c (wave-uniform, present in the list) to the front; returns its position
    __device__ __forceinline__ u32 step(u32 c) {
        int pos = 0;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            u64 m = __ballot(row[r] == c);
            if (m) pos = r * 64 + __builtin_ctzll(m);
        }
        u32 carry = c;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            u32 last = __shfl(row[r], 63, 64);
            u32 sh = __shfl_up(row[r], 1, 64);
            if (lane_id() == 0) sh = carry;
            int q = r * 64 + (int)lane_id();
            row[r] = (q <= pos) ? sh : row[r];
            carry = last;
        }
        return (u32)pos;
    }
    __device__ __forceinline__ void load(const u16 *p) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) row[r] = p[r * 64 + lane_id()];
    }
    __device__ __forceinline__ void store(u16 *p) const {
#pragma unroll
        for (int r = 0; r < ROWS; r++) p[r * 64 + lane_id()] = (u16)row[r];
    }
};

// one wave per chunk: final list from identity + number of distinct codes seen.
// lists: [chunks][ROWS*64] u16; seen: [chunks] u32.  Seen codes end up in front.
template <class Acc, int ROWS>
__global__ __launch_bounds__(64) void mtf_gen_summary_kernel(Acc acc, u64 N,
                                                             Lut16 lut,
                                                             u16 *__restrict__ lists,
                                                             u32 *__restrict__ seen) {
    const u64 base = (u64)blockIdx.x * MTFG_CH;
    WaveList<ROWS> wl;
    wl.init_identity();
    u32 maxpos1 = 0;  // distinct seen = 1 + max position ever found... tracked via first-seen count
    u32 nseen = 0;
    for (u32 o = 0; o < MTFG_CH; o += 64) {
        u64 j = base + o + lane_id();
        u32 code = (j < N) ? (u32)lut.v[acc(j) + 1] : 0xFFFFu;
        u32 cnt = (base + o + 64 <= N) ? 64u : (u32)(N > base + o ? N - (base + o) : 0);
        for (u32 t = 0; t < cnt; t++) {
            u32 c = __shfl(code, t, 64);
            u32 pos = wl.step(c);
            // a first occurrence is found at position >= nseen (unseen codes sit behind
            // the seen ones)
            if (pos >= nseen) nseen++;
        }
    }
    (void)maxpos1;
    wl.store(lists + (u64)blockIdx.x * (ROWS * 64));
    if (lane_id() == 0) seen[blockIdx.x] = nseen;
}

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// S := "S then B" for list states held across the lanes of a wave.  B is a summary: its first
// Bd entries are the distinct codes it saw, most recent first.  The result is B's recency list
// followed by the entries of S that B did not see, in S's order (one ballot-compaction per
// row instead of Bd sequential move-to-front steps).  Sd counts the leading "seen" entries of
// S (S itself a summary from the identity list); it is maintained for callers that need it.
// inB: zeroed byte table indexed by code (left zeroed); newl: scratch list.
template <int ROWS, class LT>
__device__ __forceinline__ void wl_compose(WaveList<ROWS> &S, u32 &Sd, const WaveList<ROWS> &B, u32 Bd,
                                           u32 nvalid, u8 *inB, LT *newl) {
    if (Bd == 0) return;
    const u32 lane = lane_id();
#pragma unroll
    for (int r = 0; r < ROWS; r++)
        if (r * 64 + lane < Bd) inB[B.row[r]] = 1;
    wave_fence();
    u32 off = Bd, kept = 0;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const u32 q = r * 64 + lane, v = S.row[r];
        const bool keep = q < nvalid && inB[v] == 0;
        const u64 m = __ballot(keep);
        if (keep) newl[off + __popcll(m & lanemask_lt())] = (LT)v;
        kept += (u32)__popcll(__ballot(keep && q < Sd));
        off += (u32)__popcll(m);
        if (q < Bd) newl[q] = (LT)B.row[r];
    }
    wave_fence();
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        const u32 q = r * 64 + lane;
        if (q < nvalid) S.row[r] = newl[q];
        if (q < Bd) inB[B.row[r]] = 0;
    }
    wave_fence();
    Sd = Bd + kept;
}

// Exclusive scan of the chunk summaries, in place: lists[c] := list before chunk c; slot `chunks`
// receives the final list.  One block of MTFG_SCAN_WAVES waves: every wave folds a contiguous
// segment of chunks, the segment aggregates are folded, then every wave replays its segment
// from its true incoming list.
#define MTFG_SCAN_WAVES 16
template <int ROWS>
__global__ __launch_bounds__(64 * MTFG_SCAN_WAVES) void mtf_gen_scan_kernel(u16 *lists, const u32 *seen,
                                                                           u32 chunks) {
    constexpr int LW = ROWS * 64;
    __shared__ u8 s_in[MTFG_SCAN_WAVES][LW];
    __shared__ u16 s_new[MTFG_SCAN_WAVES][LW];
    __shared__ u16 s_agg[MTFG_SCAN_WAVES][LW];
    __shared__ u32 s_aggd[MTFG_SCAN_WAVES];
    const u32 w = threadIdx.x >> 6, lane = lane_id();
    const u32 per = (chunks + MTFG_SCAN_WAVES - 1) / MTFG_SCAN_WAVES;
    const u32 lo = w * per < chunks ? w * per : chunks;
    const u32 hi = lo + per < chunks ? lo + per : chunks;
    for (int i = lane; i < LW; i += 64) s_in[w][i] = 0;
    wave_fence();
    WaveList<ROWS> st, rec;
    u32 sd = 0;
    st.init_identity();
    for (u32 c = lo; c < hi; c++) {
        rec.load(lists + (u64)c * LW);
        wl_compose<ROWS, u16>(st, sd, rec, seen[c], LW, s_in[w], s_new[w]);
    }
    st.store(s_agg[w]);
    if (lane == 0) s_aggd[w] = sd;
    __syncthreads();
    st.init_identity();
    sd = 0;
    for (u32 v = 0; v < w; v++) {
        rec.load(s_agg[v]);
        wl_compose<ROWS, u16>(st, sd, rec, s_aggd[v], LW, s_in[w], s_new[w]);
    }
    for (u32 c = lo; c < hi; c++) {
        u16 *lp = lists + (u64)c * LW;
        rec.load(lp);
        st.store(lp);
        wl_compose<ROWS, u16>(st, sd, rec, seen[c], LW, s_in[w], s_new[w]);
    }
    if (w == MTFG_SCAN_WAVES - 1) st.store(lists + (u64)chunks * LW);
}

template <class Acc, int ROWS>
__global__ __launch_bounds__(64) void mtf_gen_apply_kernel(Acc acc, u64 N,
                                                           Lut16 lut,
                                                           const u16 *__restrict__ lists,
                                                           u16 *__restrict__ idx) {
    const u64 base = (u64)blockIdx.x * MTFG_CH;
    WaveList<ROWS> wl;
    wl.load(lists + (u64)blockIdx.x * (ROWS * 64));
    for (u32 o = 0; o < MTFG_CH; o += 64) {
        u64 j = base + o + lane_id();
        u32 code = (j < N) ? (u32)lut.v[acc(j) + 1] : 0xFFFFu;
        u32 cnt = (base + o + 64 <= N) ? 64u : (u32)(N > base + o ? N - (base + o) : 0);
        u32 out = 0;
        for (u32 t = 0; t < cnt; t++) {
            u32 c = __shfl(code, t, 64);
            u32 pos = wl.step(c);
            if (lane_id() == t) out = pos;
        }
        if (j < N) idx[j] = (u16)out;
    }
}

// ---- general path, timestamps (the default beyond 64 symbols) -------------------------------
// The MTF rank of a symbol is the number of distinct symbols met since its previous occurrence.  Give
// every code a timestamp -- 257 + j for the code at position j, 256 - q for a code not met yet that
// starts at list position q -- and the rank of code c at position i is the number of codes whose
// timestamp exceeds c's: a compare per list entry instead of a shift of the list.  A wave keeps the
// timestamps across its lanes (code q: row q / 64, lane q % 64): per symbol one v_readlane for c's own
// timestamp, ROWS compares + scalar popcounts, one select for the update -- no cross-lane shuffle and
// no dependence on the data (the list-shifting wave path spends ~60 wave instructions per symbol, the
// lane chunks pay the largest rank among 64 lanes: 1 GiB of uniform bytes 258 ms, ASCII text 81 ms).
// The state a chunk starts from is a prefix MAXIMUM per code over the chunks before it (last occurrence),
// so the "summary" pass is one LDS atomicMax per run of equal symbols and the scan is 257 independent
// max-scans -- no list composition.  N + 257 must fit 32 bits.
//   mtf_ts_last_kernel   per chunk, per code: the largest timestamp in the chunk (0: code absent)
//   mtf_ts_scan_kernel   phase 0: maxima per segment of TS_SEG chunks; phase 1: exclusive over segments
//                        (seeded with the identity list), the last row = state after the text;
//                        phase 2: exclusive inside every segment, in place
//   mtf_ts_final_kernel  the list after the last symbol: position of code q = number of larger timestamps
//   mtf_ts_apply_kernel  one wave per chunk, ranks out
#define TS_CH 8192
#define TS_STRIDE 320
#define TS_SEG 256
#ifndef TS_WPB
#define TS_WPB 4
#endif

template <class Acc>
__global__ __launch_bounds__(256) void mtf_ts_last_kernel(Acc acc, u64 N, Lut16 lut, u32 *__restrict__ ts) {
    __shared__ u32 s_last[TS_STRIDE];
    __shared__ u16 s_lut[260];
    const u32 tid = threadIdx.x;
    for (u32 i = tid; i < TS_STRIDE; i += 256) s_last[i] = 0;
    for (u32 i = tid; i < 257; i += 256) s_lut[i] = lut.v[i];
    __syncthreads();
    const u64 base = (u64)blockIdx.x * TS_CH;
    for (u32 o = 0; o < TS_CH; o += 256) {
        const u64 j = base + o + tid;
        const u32 code = j < N ? (u32)s_lut[acc(j) + 1] : 0xffffu;
        // only the last position of a run of equal codes inside the wave goes to LDS
        const u32 nxt = __shfl_down(code, 1, 64);
        if (j < N && ((tid & 63) == 63 || nxt != code)) atomicMax(&s_last[code], (u32)j + 257u);
    }
    __syncthreads();
    for (u32 i = tid; i < TS_STRIDE; i += 256) ts[(u64)blockIdx.x * TS_STRIDE + i] = s_last[i];
}

// seg: [nseg + 1][TS_STRIDE]
template <int PHASE>
__global__ __launch_bounds__(TS_STRIDE) void mtf_ts_scan_kernel(u32 *__restrict__ ts, u32 chunks, u32 *__restrict__ seg,
                                                               u32 nseg, u32 sigma) {
    const u32 q = threadIdx.x;
    if (PHASE == 1) {
        u32 run = q < sigma ? 256u - q : 0u;
        for (u32 g = 0; g < nseg; g++) {
            const u32 cur = seg[(u64)g * TS_STRIDE + q];
            seg[(u64)g * TS_STRIDE + q] = run;
            run = run > cur ? run : cur;
        }
        seg[(u64)nseg * TS_STRIDE + q] = run;
        return;
    }
    const u32 g = blockIdx.x;
    const u32 lo = g * TS_SEG, hi = lo + TS_SEG < chunks ? lo + TS_SEG : chunks;
    if (PHASE == 0) {
        u32 mx = 0;
        for (u32 c = lo; c < hi; c++) {
            const u32 v = ts[(u64)c * TS_STRIDE + q];
            mx = mx > v ? mx : v;
        }
        seg[(u64)g * TS_STRIDE + q] = mx;
    } else {
        u32 run = seg[(u64)g * TS_STRIDE + q];
#pragma unroll 4
        for (u32 c = lo; c < hi; c++) {
            const u32 cur = ts[(u64)c * TS_STRIDE + q];
            ts[(u64)c * TS_STRIDE + q] = run;
            run = run > cur ? run : cur;
        }
    }
}

__global__ __launch_bounds__(TS_STRIDE) void mtf_ts_final_kernel(const u32 *__restrict__ state, u32 sigma,
                                                                u16 *__restrict__ list) {
    __shared__ u32 s_ts[TS_STRIDE];
    const u32 q = threadIdx.x;
    s_ts[q] = state[q];
    __syncthreads();
    if (q >= sigma) return;
    const u32 t = s_ts[q];
    u32 pos = 0;
    for (u32 i = 0; i < sigma; i++) pos += s_ts[i] > t ? 1u : 0u;
    list[pos] = (u16)q;
}

// The table lives in LDS: c's own timestamp is a broadcast read at a wave-uniform address, the rows are
// re-read every step (ROWS wide reads), lane 0 stores the update and the rank; the 64 ranks of a batch are
// read back one per lane.  With the table in registers the row of c has to be selected by uniform branches
// or three scalar instructions per row, and keeping the rows current costs a compare + select per row:
// 1 GiB of uniform bytes (sigma 257) 65 ms and 54 ms that way, 37 ms this way (same results from stores by
// every lane instead of lane 0, and from 8 waves per workgroup instead of 4).
// A wave works on TS_ILP chunks at once (independent tables, steps interleaved): a step is a chain of LDS
// round trips (store -> broadcast read -> row reads -> compares -> popcounts -> store), and one chain per wave
// leaves the CU waiting on latency even with every wave slot taken.
#ifndef TS_ILP
#define TS_ILP 2
#endif
template <class Acc, int ROWS>
__global__ __launch_bounds__(64 * TS_WPB) void mtf_ts_apply_kernel(Acc acc, u64 N, Lut16 lut, const u32 *__restrict__ ts_in,
                                                                  u16 *__restrict__ idx, u32 chunks) {
    // TS_WPB waves per workgroup (nothing shared between them: only so that enough waves fit a CU)
    __shared__ u32 s_ts_all[TS_WPB][TS_ILP][TS_STRIDE];
    __shared__ u32 s_out_all[TS_WPB][TS_ILP][64];
    const u32 lane = lane_id(), wv = threadIdx.x >> 6;
    const u32 first = (blockIdx.x * TS_WPB + wv) * TS_ILP;
    if (first >= chunks) return;
    const u32 nk = chunks - first < (u32)TS_ILP ? chunks - first : (u32)TS_ILP;
    for (u32 k = 0; k < nk; k++)
#pragma unroll
        for (int r = 0; r < ROWS; r++)
            s_ts_all[wv][k][r * 64 + lane] = ts_in[(u64)(first + k) * TS_STRIDE + r * 64 + lane];
    wave_fence();
    if (TS_ILP > 1 && (u64)(first + TS_ILP) * TS_CH <= N) {   // TS_ILP whole chunks: interleaved
        for (u32 o = 0; o < TS_CH; o += 64) {
            u32 code[TS_ILP];
#pragma unroll
            for (int k = 0; k < TS_ILP; k++) code[k] = (u32)lut.v[acc((u64)(first + k) * TS_CH + o + lane) + 1];
#pragma unroll 4
            for (u32 t = 0; t < 64; t++) {
                u32 c[TS_ILP], rank[TS_ILP];
#pragma unroll
                for (int k = 0; k < TS_ILP; k++) {
                    c[k] = (u32)__builtin_amdgcn_readlane((int)code[k], (int)t);
                    const u32 tc = s_ts_all[wv][k][c[k]];
                    rank[k] = 0;
#pragma unroll
                    for (int r = 0; r < ROWS; r++)
                        rank[k] += (u32)__popcll(__ballot(s_ts_all[wv][k][r * 64 + lane] > tc));
                }
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < TS_ILP; k++) {
                        s_ts_all[wv][k][c[k]] = (u32)((first + k) * TS_CH + o + t) + 257u;
                        s_out_all[wv][k][t] = rank[k];
                    }
                }
                wave_fence();   // (lane 0's stores are read by the other lanes in the next step)
            }
#pragma unroll
            for (int k = 0; k < TS_ILP; k++) idx[(u64)(first + k) * TS_CH + o + lane] = (u16)s_out_all[wv][k][lane];
            wave_fence();
        }
        return;
    }
    for (u32 k = 0; k < nk; k++) {   // the text's last chunks: one after the other
        u32 *s_ts = s_ts_all[wv][k], *s_out = s_out_all[wv][k];
        const u64 base = (u64)(first + k) * TS_CH;
        for (u32 o = 0; o < TS_CH; o += 64) {
            if (base + o >= N) break;
            const u64 j = base + o + lane;
            const u32 code = j < N ? (u32)lut.v[acc(j) + 1] : 0u;
            const u32 cnt = base + o + 64 <= N ? 64u : (u32)(N - (base + o));
            const u32 stamp0 = (u32)(base + o) + 257u;
            for (u32 t = 0; t < cnt; t++) {
                const u32 c = (u32)__builtin_amdgcn_readlane((int)code, (int)t);
                const u32 tc = s_ts[c];
                u32 rank = 0;
#pragma unroll
                for (int r = 0; r < ROWS; r++) rank += (u32)__popcll(__ballot(s_ts[r * 64 + lane] > tc));
                if (lane == 0) {
                    s_ts[c] = stamp0 + t;
                    s_out[t] = rank;
                }
                wave_fence();
            }
            if (j < N) idx[j] = (u16)s_out[lane];
            wave_fence();
        }
    }
}

// ---- general path, lane chunks (sigma <= 256: byte codes) -------------------------------
// One chunk of GM_CH symbols per LANE (64 independent sequential chains per wave instead of
// one): the lane's list is a byte array in LDS, four codes per dword; a step scans dwords from
// the front (SWAR zero-byte test), shifting them up by one byte as it goes -- cost ~ rank / 4,
// and BWT output keeps ranks small.  Pass 1 runs every chunk from the identity list (summary);
// a wave then folds its 64 lane summaries in order (wl_compose), which yields the wave / tile
// aggregate (summary kernel) or, restarted from the true incoming list, every lane's incoming
// list in place (apply kernel); pass 2 replays the chunk from it and the ranks overwrite the
// codes.  Tile summaries are scanned by mtf_gen_scan_kernel in between.
#define GM_NT 256
#define GM_CH 128
#define GM_TILE (GM_NT * GM_CH)   // 32768
#define GM_STRIDE (GM_CH + 4)     // 33 dwords: conflict-free across lanes

struct GmArgs {
    u64 N;
    u32 sigma;
    u32 ls;      // lane list stride in dwords (odd, >= ceil(sigma / 4))
    Lut8 lut;
    u16 *lists;  // [tiles + 1][ROWS * 64]
    u32 *seen;   // [tiles]
    u16 *idx;
};
static inline size_t gm_lds_bytes(u32 ls) {
    return (size_t)GM_NT * GM_STRIDE + (size_t)GM_NT * ls * 4 + 272 + 3 * 4 * 256 + 64;
}

template <class Acc>
__device__ __forceinline__ void gm_stage(Acc acc, u64 N, u64 base, const u8 *s_lut, u8 *s_code) {
    for (u32 p = threadIdx.x; p < GM_TILE; p += GM_NT) {
        u64 j = base + p;
        s_code[(p / GM_CH) * GM_STRIDE + (p % GM_CH)] = j < N ? s_lut[acc(j) + 1] : (u8)0;
    }
}
template <>
__device__ __forceinline__ void gm_stage<BwtAcc>(BwtAcc acc, u64 N, u64 base, const u8 *s_lut, u8 *s_code) {
    const u8 *src = acc.L + base;
    if ((((uintptr_t)src) & 15) == 0) {
        for (u32 c = threadIdx.x; c < GM_TILE / 16; c += GM_NT) {
            const u32 p0 = c * 16;
            const u64 j0 = base + p0;
            u32 out[4] = {0, 0, 0, 0};
            if (j0 + 16 <= N) {
                uint4 v = *reinterpret_cast<const uint4 *>(src + p0);
                u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; q++)
                    out[q] = (u32)s_lut[(x[q] & 255) + 1] | ((u32)s_lut[((x[q] >> 8) & 255) + 1] << 8) |
                             ((u32)s_lut[((x[q] >> 16) & 255) + 1] << 16) | ((u32)s_lut[(x[q] >> 24) + 1] << 24);
            } else {
                for (int q = 0; q < 16; q++)
                    if (j0 + q < N) out[q >> 2] |= (u32)s_lut[(u32)src[p0 + q] + 1] << (8 * (q & 3));
            }
            u32 *dst = reinterpret_cast<u32 *>(s_code + (p0 / GM_CH) * GM_STRIDE + (p0 % GM_CH));
            dst[0] = out[0]; dst[1] = out[1]; dst[2] = out[2]; dst[3] = out[3];
        }
    } else {
        for (u32 p = threadIdx.x; p < GM_TILE; p += GM_NT) {
            u64 j = base + p;
            s_code[(p / GM_CH) * GM_STRIDE + (p % GM_CH)] = j < N ? s_lut[(u32)src[p] + 1] : (u8)0;
        }
    }
    if (acc.primary >= (i64)base && acc.primary < (i64)(base + GM_TILE)) {
        __syncthreads();
        if (threadIdx.x == 0) {
            u32 p = (u32)(acc.primary - (i64)base);
            s_code[(p / GM_CH) * GM_STRIDE + (p % GM_CH)] =
                acc.dup ? s_lut[(u32)acc.L[acc.primary - 1] + 1] : s_lut[0];
        }
    }
}

// move code c to the front of this lane's dword-packed list; returns its rank
__device__ __forceinline__ u32 gm_step(u32 *lst, u32 c, u32 ls) {
    u32 w = lst[0];
    if ((w & 0xffu) == c) return 0;
    const u32 cc = c * 0x01010101u;
    u32 carry = c, d = 0;
    while (true) {
        const u32 x = w ^ cc;
        const u32 z = (x - 0x01010101u) & ~x & 0x80808080u;
        if (z) {
            const u32 k = (u32)__builtin_ctz(z) >> 3;
            const u32 sh = (w << 8) | carry;
            const u32 keep = k == 3 ? 0u : (0xFFFFFFFFu << (8 * (k + 1)));
            lst[d] = (w & keep) | (sh & ~keep);
            return 4 * d + k;
        }
        lst[d] = (w << 8) | carry;
        carry = w >> 24;
        if (++d >= ls) return 0;  // unreachable for codes < sigma
        w = lst[d];
    }
}

// fold the 64 lane summaries of this wave into S in lane order; WRITE_IN: lane l's list is
// replaced by the state before it (its incoming list)
template <int ROWS, bool WRITE_IN>
__device__ __forceinline__ void gm_wave_pass(WaveList<ROWS> &S, u32 &Sd, u8 *wave_lists, u32 ls, u32 d_mine,
                                             u32 sigma, u8 *inB, u8 *newl) {
    const u32 lane = lane_id();
    for (u32 l = 0; l < 64; l++) {
        const u32 Bd = __shfl(d_mine, l, 64);
        u8 *ll = wave_lists + (size_t)l * ls * 4;
        WaveList<ROWS> B;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const u32 q = r * 64 + lane;
            B.row[r] = q < sigma ? (u32)ll[q] : 0xFFFFu;
        }
        if (WRITE_IN) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const u32 q = r * 64 + lane;
                if (q < sigma) ll[q] = (u8)S.row[r];
            }
        }
        wl_compose<ROWS, u8>(S, Sd, B, Bd, sigma, inB, newl);
    }
}

template <class Acc, int ROWS, bool APPLY>
__global__ __launch_bounds__(GM_NT) void mtf_gm_kernel(Acc acc, GmArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 gm_smem[];
    u8 *s_code = gm_smem;
    u32 *s_list = reinterpret_cast<u32 *>(gm_smem + GM_NT * GM_STRIDE);
    u8 *s_lut = reinterpret_cast<u8 *>(s_list + (size_t)GM_NT * a.ls);
    u8 *s_in = s_lut + 272;
    u8 *s_new = s_in + 4 * 256;
    u8 *s_wagg = s_new + 4 * 256;
    u32 *s_wd = reinterpret_cast<u32 *>(s_wagg + 4 * 256);
    const u32 tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const u32 sigma = a.sigma, ls = a.ls;
    constexpr int LW = ROWS * 64;
    const u64 base = (u64)blockIdx.x * GM_TILE;
    for (int i = tid; i < 257; i += GM_NT) s_lut[i] = a.lut.v[i];
    for (int i = tid; i < 4 * 256; i += GM_NT) s_in[i] = 0;
    __syncthreads();
    gm_stage(acc, a.N, base, s_lut, s_code);
    __syncthreads();

    // pass 1: this lane's chunk from the identity list
    u32 *lst = s_list + (size_t)tid * ls;
    const u64 cbase = base + (u64)tid * GM_CH;
    const u32 nvalid = cbase >= a.N ? 0u : (a.N - cbase >= GM_CH ? (u32)GM_CH : (u32)(a.N - cbase));
    u32 *cw = reinterpret_cast<u32 *>(s_code + (size_t)tid * GM_STRIDE);
    for (u32 d = 0; d < ls; d++) {
        u32 v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const u32 q = 4 * d + b;
            v |= (q < sigma ? q : 0xFFu) << (8 * b);
        }
        lst[d] = v;
    }
    // (APPLY) the rank of a symbol that already occurred in the chunk does not depend on the
    // incoming list, so pass 1 writes it at once; first occurrences keep their code and are
    // flagged in `firsts` (bit p of word p / 32) for the replay of pass 2
    u32 nseen = 0;
    u32 firsts[GM_CH / 32] = {};
    for (u32 q4 = 0; q4 < GM_CH / 4; q4++) {
        const u32 wv = cw[q4];
        u32 ov = wv;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            if (4 * q4 + b < nvalid) {
                const u32 r = gm_step(lst, (wv >> (8 * b)) & 0xffu, ls);
                if (r >= nseen) {
                    nseen++;
                    if (APPLY) firsts[q4 >> 3] |= 1u << (((q4 & 7u) << 2) + b);
                } else if (APPLY) {
                    ov = (ov & ~(0xffu << (8 * b))) | (r << (8 * b));
                }
            }
        }
        if (APPLY) cw[q4] = ov;
    }
    wave_fence();
    u8 *wave_lists = reinterpret_cast<u8 *>(s_list + (size_t)(w * 64) * ls);
    WaveList<ROWS> S;
    u32 Sd = 0;
    S.init_identity();
    gm_wave_pass<ROWS, false>(S, Sd, wave_lists, ls, nseen, sigma, s_in + w * 256, s_new + w * 256);
#pragma unroll
    for (int r = 0; r < ROWS; r++)
        if (r * 64 + lane < 256) s_wagg[w * 256 + r * 64 + lane] = (u8)S.row[r];
    if (lane == 0) s_wd[w] = Sd;
    __syncthreads();

    if (!APPLY) {
        if (w == 0) {
            WaveList<ROWS> T, B;
            u32 Td = 0;
            T.init_identity();
            for (u32 v = 0; v < GM_NT / 64; v++) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) B.row[r] = s_wagg[v * 256 + ((r * 64 + lane) & 255)];
                wl_compose<ROWS, u8>(T, Td, B, s_wd[v], sigma, s_in, s_new);
            }
            T.store(a.lists + (u64)blockIdx.x * LW);
            if (lane == 0) a.seen[blockIdx.x] = Td;
        }
        return;
    }
    // true incoming list of this wave: the tile's, then the waves before it
    WaveList<ROWS> B;
    S.load(a.lists + (u64)blockIdx.x * LW);
    Sd = 0;
    for (u32 v = 0; v < w; v++) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) B.row[r] = s_wagg[v * 256 + ((r * 64 + lane) & 255)];
        wl_compose<ROWS, u8>(S, Sd, B, s_wd[v], sigma, s_in + w * 256, s_new + w * 256);
    }
    gm_wave_pass<ROWS, true>(S, Sd, wave_lists, ls, nseen, sigma, s_in + w * 256, s_new + w * 256);
    wave_fence();
    // pass 2: only the first occurrences are replayed from the true incoming list
    {
        u8 *cb = s_code + (size_t)tid * GM_STRIDE;
#pragma unroll
        for (int wd = 0; wd < GM_CH / 32; wd++) {
            u32 m = firsts[wd];
            while (m) {
                const u32 p = 32u * wd + (u32)__builtin_ctz(m);
                m &= m - 1;
                cb[p] = (u8)gm_step(lst, (u32)cb[p], ls);
            }
        }
    }
    __syncthreads();
    if ((((uintptr_t)(a.idx + base)) & 15) == 0) {
        uint4 *o = reinterpret_cast<uint4 *>(a.idx + base);
        for (u32 g = tid; g < GM_TILE / 8; g += GM_NT) {
            const u32 p = 8 * g;
            const u32 *sc = reinterpret_cast<const u32 *>(s_code + (p / GM_CH) * GM_STRIDE + (p % GM_CH));
            if (base + p + 8 <= a.N) {
                const u32 lo = sc[0], hi = sc[1];
                o[g] = make_uint4((lo & 0xffu) | ((lo & 0xff00u) << 8), ((lo >> 16) & 0xffu) | ((lo >> 24) << 16),
                                  (hi & 0xffu) | ((hi & 0xff00u) << 8), ((hi >> 16) & 0xffu) | ((hi >> 24) << 16));
            } else {
                const u8 *sb = reinterpret_cast<const u8 *>(sc);
                for (u32 q = 0; q < 8; q++)
                    if (base + p + q < a.N) a.idx[base + p + q] = (u16)sb[q];
            }
        }
    } else {
        for (u32 p = tid; p < GM_TILE; p += GM_NT)
            if (base + p < a.N) a.idx[base + p] = (u16)s_code[(p / GM_CH) * GM_STRIDE + (p % GM_CH)];
    }
}

// ---- inverse MTF (seqFromMTF): same scan, but a chunk's effect is a general
// permutation of list POSITIONS (index -> move that position to the front) -------
// perm maps: new_list[q] = old_list[perm[q]].
template <int ROWS>
__device__ __forceinline__ u32 wl_step_pos(WaveList<ROWS> &wl, u32 pos) {
    // move the entry at `pos` (wave-uniform) to the front; returns the entry
    u32 c = __shfl(wl.row[0], 0, 64);
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        u32 v = __shfl(wl.row[r], pos & 63, 64);
        if ((int)(pos >> 6) == r) c = v;
    }
    u32 carry = c;
#pragma unroll
    for (int r = 0; r < ROWS; r++) {
        u32 last = __shfl(wl.row[r], 63, 64);
        u32 sh = __shfl_up(wl.row[r], 1, 64);
        if (lane_id() == 0) sh = carry;
        u32 q = r * 64 + lane_id();
        wl.row[r] = (q <= pos) ? sh : wl.row[r];
        carry = last;
    }
    return c;
}

// one wave per chunk: position permutation of the chunk (applied to identity)
template <int ROWS>
__global__ __launch_bounds__(64) void imtf_summary_kernel(const u16 *__restrict__ idx, u64 N,
                                                          u32 sigma, u16 *__restrict__ perms,
                                                          u32 *err) {
    const u64 base = (u64)blockIdx.x * MTFG_CH;
    WaveList<ROWS> wl;
    wl.init_identity();
    for (u32 o = 0; o < MTFG_CH; o += 64) {
        u64 j = base + o + lane_id();
        u32 p = (j < N) ? (u32)idx[j] : 0u;
        if (j < N && p >= sigma) atomicOr(err, 0x100u);  // DS.index out of range
        u32 cnt = (base + o + 64 <= N) ? 64u : (u32)(N > base + o ? N - (base + o) : 0);
        for (u32 t = 0; t < cnt; t++) {
            u32 pos = __shfl(p, t, 64);
            if (pos >= sigma) pos = 0;
            (void)wl_step_pos<ROWS>(wl, pos);
        }
    }
    wl.store(perms + (u64)blockIdx.x * (ROWS * 64));
}

// S[q] := S[P[q]] for lists held across the lanes of a wave (scratch: LW entries of LDS)
template <int ROWS, class LT>
__device__ __forceinline__ void wl_gather(WaveList<ROWS> &S, const WaveList<ROWS> &P, u32 nvalid, LT *scratch) {
    const u32 lane = lane_id();
#pragma unroll
    for (int r = 0; r < ROWS; r++)
        if (r * 64 + lane < nvalid) scratch[r * 64 + lane] = (LT)S.row[r];
    wave_fence();
#pragma unroll
    for (int r = 0; r < ROWS; r++)
        if (r * 64 + lane < nvalid) S.row[r] = scratch[P.row[r] < nvalid ? P.row[r] : 0];
    wave_fence();
}

// Exclusive scan over the chunk permutations, in place: perms[c] := list (of codes) before chunk
// c.  One block of MTFG_SCAN_WAVES waves: fold a contiguous segment each (perm composition),
// fold the segment aggregates, replay every segment from its true incoming list.
template <int ROWS>
__global__ __launch_bounds__(64 * MTFG_SCAN_WAVES) void imtf_scan_kernel(u16 *perms, u32 chunks) {
    constexpr int LW = ROWS * 64;
    __shared__ u16 s_tmp[MTFG_SCAN_WAVES][LW];
    __shared__ u16 s_agg[MTFG_SCAN_WAVES][LW];
    const u32 w = threadIdx.x >> 6;
    const u32 per = (chunks + MTFG_SCAN_WAVES - 1) / MTFG_SCAN_WAVES;
    const u32 lo = w * per < chunks ? w * per : chunks;
    const u32 hi = lo + per < chunks ? lo + per : chunks;
    WaveList<ROWS> st, pm;
    st.init_identity();
    for (u32 c = lo; c < hi; c++) {
        pm.load(perms + (u64)c * LW);
        wl_gather<ROWS, u16>(st, pm, LW, s_tmp[w]);
    }
    st.store(s_agg[w]);
    __syncthreads();
    st.init_identity();
    for (u32 v = 0; v < w; v++) {
        pm.load(s_agg[v]);
        wl_gather<ROWS, u16>(st, pm, LW, s_tmp[w]);
    }
    for (u32 c = lo; c < hi; c++) {
        u16 *pp = perms + (u64)c * LW;
        pm.load(pp);
        st.store(pp);
        wl_gather<ROWS, u16>(st, pm, LW, s_tmp[w]);
    }
}

template <int ROWS>
__global__ __launch_bounds__(64) void imtf_apply_kernel(const u16 *__restrict__ idx, u64 N,
                                                        u32 sigma,
                                                        const u16 *__restrict__ states,
                                                        SymTab sym_of_code,
                                                        i16 *__restrict__ out) {
    const u64 base = (u64)blockIdx.x * MTFG_CH;
    WaveList<ROWS> wl;
    wl.load(states + (u64)blockIdx.x * (ROWS * 64));
    for (u32 o = 0; o < MTFG_CH; o += 64) {
        u64 j = base + o + lane_id();
        u32 p = (j < N) ? (u32)idx[j] : 0u;
        u32 cnt = (base + o + 64 <= N) ? 64u : (u32)(N > base + o ? N - (base + o) : 0);
        u32 res = 0;
        for (u32 t = 0; t < cnt; t++) {
            u32 pos = __shfl(p, t, 64);
            if (pos >= sigma) pos = 0;
            u32 c = wl_step_pos<ROWS>(wl, pos);
            if (lane_id() == t) res = c;
        }
        if (j < N) out[j] = sym_of_code.v[res];
    }
}

// ---- inverse MTF, lane chunks (sigma <= 256) -------------------------------------------
// Same organisation as mtf_gm_kernel.  A chunk's effect on the list is a permutation of list
// POSITIONS (new[q] = old[P[q]]), obtained by running the chunk on the identity; permutations
// compose by a gather, so a wave folds its 64 lane permutations with one LDS round trip each.
struct GmiArgs {
    u64 N;
    u32 sigma;
    u32 ls;
    const u16 *idx;
    u16 *perms;  // [tiles + 1][ROWS * 64]
    SymTab tab;
    i16 *out;
    u32 *err;
};

// move the entry at position pos of this lane's dword-packed list to the front; returns it
__device__ __forceinline__ u32 gmi_step(u32 *lst, u32 pos) {
    const u32 D = pos >> 2, k = pos & 3u;
    const u32 wD = lst[D];
    const u32 c = (wD >> (8 * k)) & 0xffu;
    if (pos == 0) return c;
    u32 carry = c;
    for (u32 d = 0; d < D; d++) {
        const u32 w = lst[d];
        lst[d] = (w << 8) | carry;
        carry = w >> 24;
    }
    const u32 sh = (wD << 8) | carry;
    const u32 keep = k == 3 ? 0u : (0xFFFFFFFFu << (8 * (k + 1)));
    lst[D] = (wD & keep) | (sh & ~keep);
    return c;
}

template <int ROWS, bool WRITE_IN>
__device__ __forceinline__ void gmi_wave_pass(WaveList<ROWS> &S, u8 *wave_lists, u32 ls, u32 sigma, u8 *scratch) {
    const u32 lane = lane_id();
    for (u32 l = 0; l < 64; l++) {
        u8 *ll = wave_lists + (size_t)l * ls * 4;
        WaveList<ROWS> P;
#pragma unroll
        for (int r = 0; r < ROWS; r++) {
            const u32 q = r * 64 + lane;
            P.row[r] = q < sigma ? (u32)ll[q] : 0u;
        }
        if (WRITE_IN) {
#pragma unroll
            for (int r = 0; r < ROWS; r++) {
                const u32 q = r * 64 + lane;
                if (q < sigma) ll[q] = (u8)S.row[r];
            }
        }
        wl_gather<ROWS, u8>(S, P, sigma, scratch);
    }
}

template <int ROWS, bool APPLY>
__global__ __launch_bounds__(GM_NT) void imtf_gm_kernel(GmiArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 gm_smem[];
    u8 *s_code = gm_smem;
    u32 *s_list = reinterpret_cast<u32 *>(gm_smem + GM_NT * GM_STRIDE);
    i16 *s_tab = reinterpret_cast<i16 *>(s_list + (size_t)GM_NT * a.ls);   // 256 entries
    u8 *s_tmp = reinterpret_cast<u8 *>(s_tab + 256);                      // 4 x 256
    u8 *s_wagg = s_tmp + 4 * 256;                                         // 4 x 256
    const u32 tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    const u32 sigma = a.sigma, ls = a.ls;
    constexpr int LW = ROWS * 64;
    const u64 base = (u64)blockIdx.x * GM_TILE;
    if (APPLY) s_tab[tid] = a.tab.v[tid];
    // stage the indices as bytes (an index >= sigma is DS.index out of range: flagged, read as 0)
    {
        const u16 *src = a.idx + base;
        bool bad = false;
        if ((((uintptr_t)src) & 15) == 0) {
            for (u32 g = tid; g < GM_TILE / 8; g += GM_NT) {
                const u32 p = 8 * g;
                u32 v[8];
                if (base + p + 8 <= a.N) {
                    const uint4 t = *reinterpret_cast<const uint4 *>(src + p);
                    v[0] = t.x & 0xffffu; v[1] = t.x >> 16; v[2] = t.y & 0xffffu; v[3] = t.y >> 16;
                    v[4] = t.z & 0xffffu; v[5] = t.z >> 16; v[6] = t.w & 0xffffu; v[7] = t.w >> 16;
                } else {
#pragma unroll
                    for (int q = 0; q < 8; q++) v[q] = base + p + q < a.N ? (u32)src[p + q] : 0u;
                }
#pragma unroll
                for (int q = 0; q < 8; q++)
                    if (v[q] >= sigma) { bad = true; v[q] = 0; }
                u32 *dst = reinterpret_cast<u32 *>(s_code + (p / GM_CH) * GM_STRIDE + (p % GM_CH));
                dst[0] = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
                dst[1] = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
            }
        } else {
            for (u32 p = tid; p < GM_TILE; p += GM_NT) {
                u32 v = base + p < a.N ? (u32)src[p] : 0u;
                if (v >= sigma) { bad = true; v = 0; }
                s_code[(p / GM_CH) * GM_STRIDE + (p % GM_CH)] = (u8)v;
            }
        }
        if (bad) atomicOr(a.err, 0x100u);
    }
    __syncthreads();

    u32 *lst = s_list + (size_t)tid * ls;
    const u64 cbase = base + (u64)tid * GM_CH;
    const u32 nvalid = cbase >= a.N ? 0u : (a.N - cbase >= GM_CH ? (u32)GM_CH : (u32)(a.N - cbase));
    u32 *cw = reinterpret_cast<u32 *>(s_code + (size_t)tid * GM_STRIDE);
    for (u32 d = 0; d < ls; d++) {
        u32 v = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const u32 q = 4 * d + b;
            v |= (q < sigma ? q : 0xFFu) << (8 * b);
        }
        lst[d] = v;
    }
    // pass 1: the chunk's position permutation
    for (u32 q4 = 0; q4 < GM_CH / 4; q4++) {
        const u32 wv = cw[q4];
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (4 * q4 + b < nvalid) (void)gmi_step(lst, (wv >> (8 * b)) & 0xffu);
    }
    wave_fence();
    u8 *wave_lists = reinterpret_cast<u8 *>(s_list + (size_t)(w * 64) * ls);
    WaveList<ROWS> S, P;
    S.init_identity();
    gmi_wave_pass<ROWS, false>(S, wave_lists, ls, sigma, s_tmp + w * 256);
#pragma unroll
    for (int r = 0; r < ROWS; r++)
        if (r * 64 + lane < 256) s_wagg[w * 256 + r * 64 + lane] = (u8)S.row[r];
    __syncthreads();

    if (!APPLY) {
        if (w == 0) {
            WaveList<ROWS> T;
            T.init_identity();
            for (u32 v = 0; v < GM_NT / 64; v++) {
#pragma unroll
                for (int r = 0; r < ROWS; r++) P.row[r] = s_wagg[v * 256 + ((r * 64 + lane) & 255)];
                wl_gather<ROWS, u8>(T, P, sigma, s_tmp);
            }
            T.store(a.perms + (u64)blockIdx.x * LW);
        }
        return;
    }
    S.load(a.perms + (u64)blockIdx.x * LW);  // list of codes before this tile
    for (u32 v = 0; v < w; v++) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) P.row[r] = s_wagg[v * 256 + ((r * 64 + lane) & 255)];
        wl_gather<ROWS, u8>(S, P, sigma, s_tmp + w * 256);
    }
    gmi_wave_pass<ROWS, true>(S, wave_lists, ls, sigma, s_tmp + w * 256);
    wave_fence();
    // pass 2: replay from the true incoming list; codes overwrite the indices
    for (u32 q4 = 0; q4 < GM_CH / 4; q4++) {
        const u32 wv = cw[q4];
        u32 ov = 0;
#pragma unroll
        for (int b = 0; b < 4; b++)
            if (4 * q4 + b < nvalid) ov |= gmi_step(lst, (wv >> (8 * b)) & 0xffu) << (8 * b);
        cw[q4] = ov;
    }
    __syncthreads();
    if ((((uintptr_t)(a.out + base)) & 15) == 0) {
        uint4 *o = reinterpret_cast<uint4 *>(a.out + base);
        for (u32 g = tid; g < GM_TILE / 8; g += GM_NT) {
            const u32 p = 8 * g;
            const u8 *sb = s_code + (p / GM_CH) * GM_STRIDE + (p % GM_CH);
            if (base + p + 8 <= a.N) {
                u32 t[4];
#pragma unroll
                for (int q = 0; q < 4; q++)
                    t[q] = (u32)(u16)s_tab[sb[2 * q]] | ((u32)(u16)s_tab[sb[2 * q + 1]] << 16);
                o[g] = make_uint4(t[0], t[1], t[2], t[3]);
            } else {
                for (u32 q = 0; q < 8; q++)
                    if (base + p + q < a.N) a.out[base + p + q] = s_tab[sb[q]];
            }
        }
    } else {
        for (u32 p = tid; p < GM_TILE; p += GM_NT)
            if (base + p < a.N) a.out[base + p] = s_tab[s_code[(p / GM_CH) * GM_STRIDE + (p % GM_CH)]];
    }
}

// ---- inverse MTF, sigma <= 16: the nibble path mirrored ---------------------------------------
// A chunk's effect on the list is a permutation of list POSITIONS (run it on the identity): P with
// out[i] = in[P[i]], 16 nibbles in one register; "a then b" = a gathered by b.  A lane runs its
// MTF_CH-index chunk once, leaving in LDS, for every index, the position of the INCOMING list it
// reads (q) -- so after the scan over lanes and tiles the symbols are sixteen-way table lookups
// into the lane's incoming list, not a second sequential pass.
__device__ __forceinline__ u64 nibi_gather(u64 a, u64 b) {   // a then b
    u64 r = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) r |= ((a >> (4 * ((b >> (4 * i)) & 15ull))) & 15ull) << (4 * i);
    return r;
}
// stage MTF_TILE indices (u16 or u8) as bytes, chunk-major with MTF_STRIDE; an index >= sigma is
// DS.index out of range: flagged, read as 0
template <class IT>
__device__ __forceinline__ void nibi_stage(const IT *__restrict__ idx, u64 N, u64 base, u32 sigma, u8 *s_code,
                                           u32 *err) {
    bool bad = false;
    const IT *src = idx + base;
    if ((((uintptr_t)src) & 15) == 0) {
        constexpr int PER = 16 / sizeof(IT);   // indices per 16-byte load
        for (u32 g = threadIdx.x; g < MTF_TILE / PER; g += MTF_NT) {
            const u32 p = PER * g;
            u32 v[PER];
            if (base + p + PER <= N) {
                const uint4 t = *reinterpret_cast<const uint4 *>(src + p);
                const u32 x[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                for (int q = 0; q < PER; q++)
                    v[q] = sizeof(IT) == 2 ? (x[q >> 1] >> (16 * (q & 1))) & 0xffffu : (x[q >> 2] >> (8 * (q & 3))) & 0xffu;
            } else {
#pragma unroll
                for (int q = 0; q < PER; q++) v[q] = base + p + q < N ? (u32)src[p + q] : 0u;
            }
#pragma unroll
            for (int q = 0; q < PER; q++)
                if (v[q] >= sigma) { bad = true; v[q] = 0; }
            u32 *dst = reinterpret_cast<u32 *>(s_code + (p / MTF_CH) * MTF_STRIDE + (p % MTF_CH));
#pragma unroll
            for (int q = 0; q < PER / 4; q++)
                dst[q] = v[4 * q] | (v[4 * q + 1] << 8) | (v[4 * q + 2] << 16) | (v[4 * q + 3] << 24);
        }
    } else {
        for (u32 p = threadIdx.x; p < MTF_TILE; p += MTF_NT) {
            u32 v = base + p < N ? (u32)src[p] : 0u;
            if (v >= sigma) { bad = true; v = 0; }
            s_code[(p / MTF_CH) * MTF_STRIDE + (p % MTF_CH)] = (u8)v;
        }
    }
    if (bad) atomicOr(err, 0x100u);
}
// this lane's chunk on the identity: returns its permutation; RECORD: every index byte is replaced
// by the incoming-list position it reads
template <bool RECORD>
__device__ __forceinline__ u64 nibi_chunk(u8 *s_code, u32 nvalid) {
    u32 *cw = reinterpret_cast<u32 *>(s_code + threadIdx.x * MTF_STRIDE);
    u64 q = NIB_IDENT;
#pragma unroll 4
    for (u32 w = 0; w < MTF_CH / 4; w++) {
        const u32 wv = cw[w];
        u32 ov = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            if (4 * w + b < nvalid) {
                const u32 r = (wv >> (8 * b)) & 15u;
                const u32 c = (u32)(q >> (4 * r)) & 15u;
                q = nib_front(q, r, c);
                ov |= c << (8 * b);
            }
        }
        if (RECORD) cw[w] = ov;
    }
    return q;
}
// exclusive block scan of permutations (lane order); *agg = the block's permutation
__device__ __forceinline__ u64 nibi_block_excl(u64 mine, u64 *s_w, u64 *agg) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    u64 inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u64 t = __shfl_up(inc, d, 64);
        if (l >= d) inc = nibi_gather(t, inc);
    }
    u64 exc = __shfl_up(inc, 1, 64);
    if (l == 0) exc = NIB_IDENT;
    if (l == 63) s_w[w] = inc;
    __syncthreads();
    u64 pre = NIB_IDENT, tot = NIB_IDENT;
    for (int i = 0; i < MTF_NT / 64; i++) {
        if (i < w) pre = nibi_gather(pre, s_w[i]);
        tot = nibi_gather(tot, s_w[i]);
    }
    *agg = tot;
    return nibi_gather(pre, exc);
}
template <class IT>
__global__ __launch_bounds__(MTF_NT) void imtf_nib_summary_kernel(const IT *__restrict__ idx, u64 N, u32 sigma,
                                                                   u64 *__restrict__ t_perm, u32 *err) {
    __shared__ __attribute__((aligned(16))) u8 s_code[MTF_NT * MTF_STRIDE];
    __shared__ u64 s_w[MTF_NT / 64];
    const u64 base = (u64)blockIdx.x * MTF_TILE;
    nibi_stage<IT>(idx, N, base, sigma, s_code, err);
    __syncthreads();
    const u64 cbase = base + (u64)threadIdx.x * MTF_CH;
    const u32 nvalid = cbase >= N ? 0u : (N - cbase >= MTF_CH ? (u32)MTF_CH : (u32)(N - cbase));
    const u64 mine = nibi_chunk<false>(s_code, nvalid);
    u64 agg;
    (void)nibi_block_excl(mine, s_w, &agg);
    if (threadIdx.x == 0) t_perm[blockIdx.x] = agg;
}
// exclusive scan over the tile permutations, in place (one block)
__global__ __launch_bounds__(MTF_NT) void imtf_nib_scan_kernel(u64 *t_perm, u32 tiles) {
    __shared__ u64 s_w[MTF_NT / 64];
    const u32 per = (tiles + MTF_NT - 1) / MTF_NT;
    const u32 lo = threadIdx.x * per, hi = lo + per < tiles ? lo + per : tiles;
    u64 mine = NIB_IDENT;
    for (u32 t = lo; t < hi; t++) mine = nibi_gather(mine, t_perm[t]);
    u64 agg;
    u64 run = nibi_block_excl(mine, s_w, &agg);
    for (u32 t = lo; t < hi; t++) {
        const u64 cur = t_perm[t];
        t_perm[t] = run;
        run = nibi_gather(run, cur);
    }
}
// OT = i16: symbols through `tab`; OT = u8: the codes themselves
template <class IT, class OT>
__global__ __launch_bounds__(MTF_NT) void imtf_nib_apply_kernel(const IT *__restrict__ idx, u64 N, u32 sigma,
                                                                 const u64 *__restrict__ t_perm, SymTab tab,
                                                                 OT *__restrict__ out, u32 *err) {
    __shared__ __attribute__((aligned(16))) u8 s_code[MTF_NT * MTF_STRIDE];
    __shared__ u64 s_w[MTF_NT / 64];
    __shared__ i16 s_tab[16];
    if (threadIdx.x < 16) s_tab[threadIdx.x] = tab.v[threadIdx.x];
    const u64 base = (u64)blockIdx.x * MTF_TILE;
    nibi_stage<IT>(idx, N, base, sigma, s_code, err);
    __syncthreads();
    const u64 cbase = base + (u64)threadIdx.x * MTF_CH;
    const u32 nvalid = cbase >= N ? 0u : (N - cbase >= MTF_CH ? (u32)MTF_CH : (u32)(N - cbase));
    const u64 mine = nibi_chunk<true>(s_code, nvalid);
    u64 agg;
    const u64 exc = nibi_block_excl(mine, s_w, &agg);
    // the list this lane's chunk starts from (codes: the initial list is the identity)
    const u64 lin = nibi_gather(t_perm[blockIdx.x], exc);
    u32 *cw = reinterpret_cast<u32 *>(s_code + threadIdx.x * MTF_STRIDE);
#pragma unroll 4
    for (u32 w = 0; w < MTF_CH / 4; w++) {
        const u32 wv = cw[w];
        u32 ov = 0;
#pragma unroll
        for (int b = 0; b < 4; b++) ov |= ((u32)(lin >> (4 * ((wv >> (8 * b)) & 15u))) & 15u) << (8 * b);
        cw[w] = ov;
    }
    __syncthreads();
    if (sizeof(OT) == 2) {
        if ((((uintptr_t)(out + base)) & 15) == 0) {
            uint4 *o = reinterpret_cast<uint4 *>(out + base);
            for (u32 g = threadIdx.x; g < MTF_TILE / 8; g += MTF_NT) {
                const u32 p = 8 * g;
                const u8 *sb = s_code + (p / MTF_CH) * MTF_STRIDE + (p % MTF_CH);
                if (base + p + 8 <= N) {
                    u32 t[4];
#pragma unroll
                    for (int q = 0; q < 4; q++)
                        t[q] = (u32)(u16)s_tab[sb[2 * q]] | ((u32)(u16)s_tab[sb[2 * q + 1]] << 16);
                    o[g] = make_uint4(t[0], t[1], t[2], t[3]);
                } else {
                    for (u32 q = 0; q < 8; q++)
                        if (base + p + q < N) out[base + p + q] = (OT)s_tab[sb[q]];
                }
            }
        } else {
            for (u32 p = threadIdx.x; p < MTF_TILE; p += MTF_NT)
                if (base + p < N) out[base + p] = (OT)s_tab[s_code[(p / MTF_CH) * MTF_STRIDE + (p % MTF_CH)]];
        }
    } else {
        if ((((uintptr_t)(out + base)) & 15) == 0) {
            uint4 *o = reinterpret_cast<uint4 *>(out + base);
            for (u32 g = threadIdx.x; g < MTF_TILE / 16; g += MTF_NT) {
                const u32 p = 16 * g;
                const u32 *sb = reinterpret_cast<const u32 *>(s_code + (p / MTF_CH) * MTF_STRIDE + (p % MTF_CH));
                if (base + p + 16 <= N) {
                    o[g] = make_uint4(sb[0], sb[1], sb[2], sb[3]);
                } else {
                    const u8 *s8 = reinterpret_cast<const u8 *>(sb);
                    for (u32 q = 0; q < 16; q++)
                        if (base + p + q < N) out[base + p + q] = (OT)s8[q];
                }
            }
        } else {
            for (u32 p = threadIdx.x; p < MTF_TILE; p += MTF_NT)
                if (base + p < N) out[base + p] = (OT)s_code[(p / MTF_CH) * MTF_STRIDE + (p % MTF_CH)];
        }
    }
}

// ---- which general path?  The lane chunks cost ~ rank / 4 list words per symbol, the wave chunks a
// constant; on large alphabets with uniformly spread symbols (average rank ~ sigma / 2) the wave
// chunks win.  The rank of a symbol is the number of distinct symbols since its last occurrence, so
// the number of distinct symbols in short windows tells the two cases apart: MRS_BLOCKS * 256
// windows of MRS_WIN symbols spread over the stream, one per lane, a 256-bit set each.
#define MRS_WIN 256
#define MRS_BLOCKS 16
template <class Acc>
__global__ __launch_bounds__(256) void mtf_rank_sample_kernel(Acc acc, u64 N, u64 *__restrict__ out) {
    __shared__ u32 s_set[256 * 9];
    u32 *set = s_set + threadIdx.x * 9;
    for (int i = 0; i < 8; i++) set[i] = 0;
    const u64 w = (u64)blockIdx.x * 256 + threadIdx.x;
    const u64 start = (N - MRS_WIN) / (MRS_BLOCKS * 256 - 1) * w;
    for (u32 k = 0; k < MRS_WIN; k++) {
        const u32 c = (u32)(acc(start + k) & 255);
        set[c >> 5] |= 1u << (c & 31);
    }
    u32 sum = 0;
    for (int i = 0; i < 8; i++) sum += (u32)__popc(set[i]);
    for (int d = 32; d >= 1; d >>= 1) sum += (u32)__shfl_xor((int)sum, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd((unsigned long long *)out, (unsigned long long)sum);
}

// ---- sigma = 257: every byte value AND the sentinel -----------------------------------
// Nine-bit codes do not fit the byte lists of the lane-chunk kernels, but the sentinel of a BWT
// occurs exactly once, at the primary row p > 0, and the initial list has it in front.  Until p
// the list therefore reads [the d byte values met so far, by recency] Nothing [the others,
// sorted]; after p the same with "met since p".  So the ranks are those of a 256-symbol MTF over
// the bytes alone (row p repeating the byte before it: rank 0, list untouched), plus one for every
// FIRST occurrence of a byte value before p and for every first occurrence after p; row p itself
// gets d = the number of distinct values before it, and the final list is the byte list with
// Nothing inserted behind the values met after p.
#define M257_NONE 0xffffffffu
// first[b] = first row < p holding byte b, first[256 + b] = first row > p (M257_NONE: none)
__global__ __launch_bounds__(256) void mtf257_first_kernel(const u8 *__restrict__ L, u64 N, u64 p,
                                                           u32 *__restrict__ first) {
    __shared__ u32 s_first[512];
    for (int i = threadIdx.x; i < 512; i += 256) s_first[i] = M257_NONE;
    __syncthreads();
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < N; j += (u64)gridDim.x * 256) {
        if (j == p) continue;
        const u32 k = (u32)L[j] + (j > p ? 256u : 0u);
        if ((u32)j < s_first[k]) atomicMin(&s_first[k], (u32)j);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 256)
        if (s_first[i] != M257_NONE) atomicMin(&first[i], s_first[i]);
}
// res[0] = distinct values before p (= the rank of the sentinel), res[1] = distinct values after p
__global__ __launch_bounds__(512) void mtf257_fix_kernel(u16 *__restrict__ idx, const u32 *__restrict__ first,
                                                         u64 p, u64 *__restrict__ res) {
    __shared__ u32 s_n[2];
    if (threadIdx.x < 2) s_n[threadIdx.x] = 0;
    __syncthreads();
    const u32 f = first[threadIdx.x];
    if (f != M257_NONE) {
        idx[f] += 1;
        atomicAdd(&s_n[threadIdx.x >> 8], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        idx[p] = (u16)s_n[0];
        res[0] = s_n[0];
        res[1] = s_n[1];
    }
}

// The inverse.  d (the number of distinct values met) grows at every row whose rank exceeds it,
// at most 256 times before p and 256 times after; those rows are exactly the first occurrences.
#define M257_TILE 4096
__global__ __launch_bounds__(256) void imtf257_tmax_kernel(const u16 *__restrict__ idx, u64 N,
                                                           u16 *__restrict__ tmax) {
    __shared__ u32 s[4];
    const u64 base = (u64)blockIdx.x * M257_TILE;
    u32 m = 0;
    for (int k = 0; k < M257_TILE / 256; k++) {
        const u64 j = base + (u64)k * 256 + threadIdx.x;
        if (j < N) {
            const u32 r = idx[j];
            m = r > m ? r : m;
        }
    }
    for (int d = 32; d >= 1; d >>= 1) {
        const u32 o = (u32)__shfl_xor((int)m, d, 64);
        m = o > m ? o : m;
    }
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 a = s[0] > s[1] ? s[0] : s[1], b = s[2] > s[3] ? s[2] : s[3];
        tmax[blockIdx.x] = (u16)(a > b ? a : b);
    }
}
// block-wide minimum of v (M257_NONE = no candidate); every thread returns it
__device__ __forceinline__ u32 m257_block_min(u32 v, u32 *s_red) {
    for (int d = 32; d >= 1; d >>= 1) {
        const u32 o = (u32)__shfl_xor((int)v, d, 64);
        v = o < v ? o : v;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    u32 a = s_red[0] < s_red[1] ? s_red[0] : s_red[1], b = s_red[2] < s_red[3] ? s_red[2] : s_red[3];
    return a < b ? a : b;
}
// one block: fix[0..res[0]) = the first occurrences before p, fix[256..256+res[1]) = those after p
__global__ __launch_bounds__(256) void imtf257_chain_kernel(const u16 *__restrict__ idx, u64 N, u64 p,
                                                            const u16 *__restrict__ tmax,
                                                            u32 *__restrict__ fix, u64 *__restrict__ res) {
    __shared__ u16 s_r[M257_TILE];
    __shared__ u32 s_red[4];
    const u32 tid = threadIdx.x;
    for (int reg = 0; reg < 2; reg++) {
        const u64 lo = reg ? p + 1 : 0, hi = reg ? N : p;
        u32 d = 0;
        u64 cur = lo;
        while (cur < hi && d < 256) {
            // the next tile that can hold a rank above d (the tile `cur` is inside is always read)
            u64 t = cur / M257_TILE;
            if (cur % M257_TILE == 0) {
                const u64 tend = (hi + M257_TILE - 1) / M257_TILE;
                u32 found = M257_NONE;
                while (t < tend) {
                    u32 c = M257_NONE;
                    for (int q = 0; q < 8; q++) {
                        const u64 tt = t + (u64)tid * 8 + q;
                        if (c == M257_NONE && tt < tend && tmax[tt] > d) c = (u32)(tt - t);
                    }
                    found = m257_block_min(c, s_red);
                    if (found != M257_NONE) break;
                    t += 2048;
                }
                if (found == M257_NONE) break;
                t += found;
                cur = t * M257_TILE > cur ? t * M257_TILE : cur;
            }
            const u64 tb = t * M257_TILE;
            const u64 te = tb + M257_TILE < hi ? tb + M257_TILE : hi;
            __syncthreads();
            for (u32 q = tid; q < M257_TILE; q += 256) s_r[q] = tb + q < te ? idx[tb + q] : (u16)0;
            __syncthreads();
            for (;;) {
                u32 c = M257_NONE;
                const u32 q0 = tid * 16;
                for (u32 q = q0; q < q0 + 16; q++)
                    if (c == M257_NONE && tb + q >= cur && s_r[q] > d) c = q;
                const u32 f = m257_block_min(c, s_red);
                if (f == M257_NONE) break;
                if (tid == 0) fix[reg * 256 + d] = (u32)(tb + f);
                d++;
                cur = tb + f + 1;
                if (d >= 256) break;
            }
            if (d < 256) cur = tb + M257_TILE;
        }
        if (tid == 0) res[reg] = d;
        __syncthreads();
    }
}
// every row other than p must have a rank different from the d of its stretch (a rank equal to d
// would be another sentinel), and row p the rank d; res[2] != 0 otherwise (the caller then takes
// the nine-bit path, which reproduces the reference on any index stream)
__global__ __launch_bounds__(256) void imtf257_check_kernel(const u16 *__restrict__ idx, u64 N, u64 p,
                                                            const u32 *__restrict__ fix,
                                                            u64 *__restrict__ res) {
    __shared__ u32 s_fix[512];
    const u32 nA = (u32)res[0], nB = (u32)res[1];
    for (int i = threadIdx.x; i < 512; i += 256)
        s_fix[i] = (i < 256 ? (u32)i < nA : (u32)(i - 256) < nB) ? fix[i] : M257_NONE;
    __syncthreads();
    bool bad = false;
    for (u64 j = (u64)blockIdx.x * 256 + threadIdx.x; j < N; j += (u64)gridDim.x * 256) {
        const u32 r = idx[j];
        if (j == p) {
            bad |= r != nA;
            continue;
        }
        const u32 *f = s_fix + (j > p ? 256 : 0);
        u32 lo = 0, hi = 256;          // d = number of first occurrences before row j
        while (lo < hi) {
            const u32 mid = (lo + hi) >> 1;
            if (f[mid] < (u32)j) lo = mid + 1;
            else hi = mid;
        }
        bad |= r == lo && !(lo < 256 && f[lo] == (u32)j);
        bad |= r > 256;
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr((unsigned long long *)&res[2], 1ull);
}
__global__ __launch_bounds__(512) void imtf257_apply_kernel(u16 *__restrict__ idx, u64 p,
                                                            const u32 *__restrict__ fix,
                                                            const u64 *__restrict__ res) {
    const u32 t = threadIdx.x;
    if ((t < 256 ? t < (u32)res[0] : t - 256 < (u32)res[1])) idx[fix[t]] -= 1;
    if (t == 0) idx[p] = 0;
}
__global__ void imtf257_sentinel_kernel(i16 *out, u64 p) { out[p] = -1; }

#endif  // __HIPCC__
