// tc_fm_host.hpp -- FM-index: build, batched count, locate.
//
// Replaces (reference FMIndex/Internal.hs) seqToCc :275-316 (C[c], there derived
// from an O(n^2) rotation matrix, BWT/Internal.hs:209-241; here a byte histogram),
// seqToOccCK :195-259 (a full sigma x N table of inclusive counts; here one rank
// bit-vector per present byte: 64-byte lines = {u64 ones-before, 7 x u64 bits}, so
// every Occ(c,k) lookup touches exactly one line), countFMIndex :347-438 and
// locateFMIndex :448-542 mapped over the pattern list (FMIndex.hs:362-379,
// 411-432, 475-497: serial or parListChunk sparks; here one lane per pattern in one
// launch, result order = pattern order).
//
// Two symbols per step (round 3).  One backward-search step is one dependent random 64-byte line read, and the
// batch of BASELINE configs[3] runs at the rate the memory system serves such reads (50 G lines/s from 2^27-byte
// texts on, whatever the index size: scripts/fm_sweep.py) -- so the lever is the NUMBER of dependent reads.  For
// texts of at most FM_PAIR_SIGMA byte values a second set of rank bit-vectors is kept, one per PAIR (a, b) of
// byte values: bit j is set iff row j's suffix is preceded by "ab" (L[j] = b and T[SA[j] - 2] = a).  With
// C2[ab] = C[a] + Occ(a, C[b]) (the start of the "ab" interval) two pattern symbols are consumed by one lookup:
//   s' = C2[ab] + Occ2(ab, s - 1) + 1,  e' = C2[ab] + Occ2(ab, e)
// which is exactly what two steps of countFMIndex (:424-432) compute, because the rows with pair ab inside a
// range keep their relative order among the "ab"-prefixed suffixes.  The reference's stop rules stay as they
// are: the range is tested for emptiness before every (single or double) step (:387-389), a byte that is not
// in the text stops the loop where the reference stops it (:393,:421) -- a pair is only taken when both of its
// bytes occur -- and an empty range between the two symbols of a pair stays empty, so the result (0 = Nothing)
// is the same.  Cost: sigma^2 / 7 bytes per text byte (3.6 N for ACGTN).
#pragma once
#include "tc_decode_host.hpp"

#define FM_LINE_BITS 448  // 7 words of payload per 64-byte line
#define FM_PAIR_SIGMA 5   // pair vectors for texts of at most this many byte values (25 vectors)

struct tc_fm {
    int device = 0;
    u64 n = 0, N = 0, primary = 0;
    u32 sigma_bytes = 0;  // present byte values
    u8 *d_L = nullptr;
    u32 *d_sa = nullptr;
    u64 *d_bits = nullptr;  // [sigma_bytes][lines][8]
    u32 *d_tab = nullptr;   // [0..255] code of byte (0xFFFFFFFF absent), [256..511] C[code], [512..767] cnt[code]
    u64 *d_bits2 = nullptr; // [sigma_bytes^2][lines][8]: one rank bit-vector per pair of byte values (or null)
    u32 *d_tab2 = nullptr;  // [FM_PAIR_SIGMA^2] C2[a * sigma_bytes + b]
    void *bits2_chunks = nullptr;   // d_bits2 as mapped chunks (tc_chunked_alloc) instead of one hipMalloc block, or null
    u64 lines = 0;
    u32 counts[256];
    i16 sym_of_code[256];
};

#ifdef __HIPCC__

// one wave per line: ballots of (symbol == c) for the line's 7 words
__global__ __launch_bounds__(256) void fm_bits_kernel(const u8 *__restrict__ L, u64 N, u64 primary,
                                                      const u32 *__restrict__ tab, u32 sigma,
                                                      u64 lines, u64 *__restrict__ bits) {
    __shared__ u32 s_code[256];
    s_code[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    const u64 line = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= lines) return;
    const u32 l = threadIdx.x & 63;
    u32 sw[7];
#pragma unroll
    for (int w = 0; w < 7; w++) {
        u64 j = line * FM_LINE_BITS + (u64)w * 64 + l;
        sw[w] = (j < N && j != primary) ? s_code[L[j]] : 0xFFFFFFFFu;
    }
    for (u32 c = 0; c < sigma; c++) {
        u64 mine = 0;
        u32 pc = 0;
#pragma unroll
        for (int w = 0; w < 7; w++) {
            u64 m = __ballot(sw[w] == c);
            pc += (u32)__popcll(m);
            if ((int)l == w + 1) mine = m;
        }
        if (l == 0) mine = pc;  // ones in this line; made cumulative by fm_scan_kernel
        if (l < 8) bits[((u64)c * lines + line) * 8 + l] = mine;
    }
}

// per symbol: exclusive scan of the line counts (word 0 of each line); block c
__global__ __launch_bounds__(1024) void fm_scan_kernel(u64 *bits, u64 lines) {
    __shared__ u64 s_part[1024];
    u64 *b = bits + (u64)blockIdx.x * lines * 8;
    u64 per = (lines + 1023) / 1024;
    u64 lo = threadIdx.x * per, hi = lo + per < lines ? lo + per : lines;
    u64 v = 0;
    for (u64 t = lo; t < hi; t++) v += b[t * 8];
    s_part[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        u64 run = 0;
        for (int i = 0; i < 1024; i++) {
            u64 c = s_part[i];
            s_part[i] = run;
            run += c;
        }
    }
    __syncthreads();
    u64 run = s_part[threadIdx.x];
    for (u64 t = lo; t < hi; t++) {
        u64 c = b[t * 8];
        b[t * 8] = run;
        run += c;
    }
}

// the pair vectors: row j carries pair (a, b) = (T[SA[j] - 2], L[j]) when SA[j] >= 2 (a row whose suffix starts at
// text position 0 or 1 has no pair: nothing can be matched two symbols to its left)
__global__ __launch_bounds__(256) void fm_bits2_kernel(const u8 *__restrict__ L, const u32 *__restrict__ sa,
                                                       const u8 *__restrict__ text, u64 N,
                                                       const u32 *__restrict__ tab, u32 sigma, u64 lines,
                                                       u64 *__restrict__ bits2) {
    __shared__ u32 s_code[256];
    s_code[threadIdx.x] = tab[threadIdx.x];
    __syncthreads();
    const u64 line = (u64)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (line >= lines) return;
    const u32 l = threadIdx.x & 63;
    u32 sw[7];
#pragma unroll
    for (int w = 0; w < 7; w++) {
        const u64 j = line * FM_LINE_BITS + (u64)w * 64 + l;
        u32 pc = 0xFFFFFFFFu;
        if (j < N) {
            const u32 p = sa[j];
            if (p >= 2) pc = s_code[text[p - 2]] * sigma + s_code[L[j]];
        }
        sw[w] = pc;
    }
    for (u32 c = 0; c < sigma * sigma; c++) {
        u64 mine = 0;
        u32 pc = 0;
#pragma unroll
        for (int w = 0; w < 7; w++) {
            const u64 m = __ballot(sw[w] == c);
            pc += (u32)__popcll(m);
            if ((int)l == w + 1) mine = m;
        }
        if (l == 0) mine = pc;
        if (l < 8) bits2[((u64)c * lines + line) * 8 + l] = mine;
    }
}

// Occ(c, k): occurrences of code c in L[0 .. k)
__device__ __forceinline__ u64 fm_occ(const u64 *__restrict__ bits, u64 lines, u32 c, u64 k) {
    u64 line = k / FM_LINE_BITS;
    u32 off = (u32)(k - line * FM_LINE_BITS);
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(bits + ((u64)c * lines + line) * 8);
    ulonglong2 a = p[0], b = p[1], cc = p[2], d = p[3];
    u64 w[7] = {a.y, b.x, b.y, cc.x, cc.y, d.x, d.y};
    u64 r = a.x;
    u32 full = off >> 6, rem = off & 63;
#pragma unroll
    for (int i = 0; i < 7; i++) {
        u64 m = ((u32)i < full) ? ~0ull : (((u32)i == full) ? ((1ull << rem) - 1ull) : 0ull);
        r += (u64)__popcll(w[i] & m);
    }
    return r;
}

// Occ(c, k1) and Occ(c, k2), k1 <= k2, of one backward-search step.  Once the range [s, e] has
// narrowed (after ~13 steps on a 2^28 DNA text it is a single row) both positions lie in the same
// 64-byte line: it is fetched once.  Lanes whose positions straddle two lines fetch the second one.
__device__ __forceinline__ void fm_occ2(const u64 *__restrict__ bits, u64 lines, u32 c, u64 k1, u64 k2,
                                        u64 *r1, u64 *r2) {
    const u64 line1 = k1 / FM_LINE_BITS, line2 = k2 / FM_LINE_BITS;
    const u32 off1 = (u32)(k1 - line1 * FM_LINE_BITS), off2 = (u32)(k2 - line2 * FM_LINE_BITS);
    const ulonglong2 *p = reinterpret_cast<const ulonglong2 *>(bits + ((u64)c * lines + line1) * 8);
    ulonglong2 a = p[0], b = p[1], cc = p[2], d = p[3];
    {
        const u64 w[7] = {a.y, b.x, b.y, cc.x, cc.y, d.x, d.y};
        u64 r = a.x;
        const u32 full = off1 >> 6, rem = off1 & 63;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const u64 m = ((u32)i < full) ? ~0ull : (((u32)i == full) ? ((1ull << rem) - 1ull) : 0ull);
            r += (u64)__popcll(w[i] & m);
        }
        *r1 = r;
    }
    if (line2 != line1) {
        const ulonglong2 *q = reinterpret_cast<const ulonglong2 *>(bits + ((u64)c * lines + line2) * 8);
        a = q[0]; b = q[1]; cc = q[2]; d = q[3];
    }
    {
        const u64 w[7] = {a.y, b.x, b.y, cc.x, cc.y, d.x, d.y};
        u64 r = a.x;
        const u32 full = off2 >> 6, rem = off2 & 63;
#pragma unroll
        for (int i = 0; i < 7; i++) {
            const u64 m = ((u32)i < full) ? ~0ull : (((u32)i == full) ? ((1ull << rem) - 1ull) : 0ull);
            r += (u64)__popcll(w[i] & m);
        }
        *r2 = r;
    }
}

// C2[a * sigma + b] = C[a] + Occ(a, C[b]): the 0-based start of the interval of suffixes that begin with "ab"
__global__ void fm_c2_kernel(const u64 *__restrict__ bits, u64 lines, const u32 *__restrict__ tab, u32 sigma,
                             u32 *__restrict__ tab2) {
    const u32 t = threadIdx.x;
    if (t >= sigma * sigma) return;
    const u32 a = t / sigma, b = t % sigma;
    tab2[t] = tab[256 + a] + (u32)fm_occ(bits, lines, a, (u64)tab[256 + b]);
}

// countFMIndex (FMIndex/Internal.hs:347-438), one pattern per lane.
// ranges (optional): [2p] = s, [2p+1] = e (1-based inclusive) for non-empty results.
template <bool PAIRS>
__global__ __launch_bounds__(256) void fm_count_kernel(const u64 *__restrict__ bits, const u64 *__restrict__ bits2,
                                                       u64 lines, const u32 *__restrict__ tab,
                                                       const u32 *__restrict__ tab2, u32 sigma,
                                                       const u8 *__restrict__ pats,
                                                       const u64 *__restrict__ offs, u64 npat,
                                                       i64 *__restrict__ out,
                                                       u64 *__restrict__ ranges) {
    __shared__ u32 s_tab[768];
    __shared__ u32 s_tab2[FM_PAIR_SIGMA * FM_PAIR_SIGMA];
    for (int i = threadIdx.x; i < 768; i += 256) s_tab[i] = tab[i];
    if (PAIRS && threadIdx.x < FM_PAIR_SIGMA * FM_PAIR_SIGMA) s_tab2[threadIdx.x] = tab2[threadIdx.x];
    __syncthreads();
    u64 p = (u64)blockIdx.x * 256 + threadIdx.x;
    if (p >= npat) return;
    const u64 beg = offs[p], end = offs[p + 1];
    i64 s = -1, e = -1;
    bool first = true, flag = false;
    // the pattern is read right to left through an aligned 8-byte window: one load per 8 steps instead of
    // one uncoalesced byte load per step (64 lanes = 64 different lines every time).  The aligned word
    // that holds a valid byte lies in that byte's page, so reading it whole is always safe.
    uintptr_t wbase = ~(uintptr_t)0;
    u64 word = 0;
    auto byte_at = [&](u64 q) -> u32 {   // pats[q]
        const uintptr_t ad = (uintptr_t)(pats + q);
        if ((ad & ~(uintptr_t)7) != wbase) {
            wbase = ad & ~(uintptr_t)7;
            word = *reinterpret_cast<const u64 *>(wbase);
        }
        return (u32)(word >> (8 * (ad & 7))) & 255u;
    };
    u64 q = end;
    while (q > beg) {                   // right to left (:375)
        if (s > e) {                    // :387-389
            flag = true;
            break;
        }
        const u32 c = s_tab[byte_at(q - 1)];
        if (c == 0xFFFFFFFFu) break;    // findIndexL = Nothing: the loop just stops (:393,:421)
        const i64 C = (i64)s_tab[256 + c];
        if (first) {                    // :391-418
            s = C + 1;
            e = C + (i64)s_tab[512 + c];
            first = false;
            q--;
            continue;
        }
        if (PAIRS && q - 1 > beg) {     // two symbols by one lookup, when the one to the left occurs in the text too
            const u32 a = s_tab[byte_at(q - 2)];
            if (a != 0xFFFFFFFFu) {
                const u32 pr = a * sigma + c;
                u64 o1, o2;
                fm_occ2(bits2, lines, pr, (u64)(s - 1), (u64)e, &o1, &o2);
                const i64 C2 = (i64)s_tab2[pr];
                s = C2 + (i64)o1 + 1;
                e = C2 + (i64)o2;
                q -= 2;
                continue;
            }
        }
        u64 o1, o2;                     // :424-432 (s <= e here: s - 1 < e)
        fm_occ2(bits, lines, c, (u64)(s - 1), (u64)e, &o1, &o2);
        s = C + (i64)o1 + 1;
        e = C + (i64)o2;
        q--;
    }
    i64 cnt = (first || (e - s + 1) == 0 || flag) ? 0 : (e - s + 1);  // :366-371
    out[p] = cnt;
    if (ranges) {
        ranges[2 * p] = cnt ? (u64)s : 0;
        ranges[2 * p + 1] = cnt ? (u64)e : 0;
    }
}

__global__ __launch_bounds__(256) void fm_cnt_to_u64_kernel(const i64 *__restrict__ cnt, u64 npat,
                                                            u64 *__restrict__ len) {
    u64 p = (u64)blockIdx.x * 256 + threadIdx.x;
    if (p < npat) len[p] = (u64)cnt[p];
}

// locate: hits[off[p] + t] = sa[s - 1 + t] + 1 (FMIndex.hs:496: suffixstartpos, 1-based)
__global__ __launch_bounds__(256) void fm_locate_fill_kernel(const u64 *__restrict__ ranges,
                                                             const u64 *__restrict__ hoffs,
                                                             const u32 *__restrict__ sa, u64 npat,
                                                             u64 cap, u64 *__restrict__ hits) {
    u64 p = (u64)blockIdx.x * 256 + threadIdx.x;
    bool in = p < npat;
    u64 s = in ? ranges[2 * p] : 0, e = in ? ranges[2 * p + 1] : 0;
    u64 o = in ? hoffs[p] : 0;
    u64 len = (in && s) ? e - s + 1 : 0;
    const u64 LONG = 32;
    if (len && len < LONG)
        for (u64 t = 0; t < len; t++)
            if (o + t < cap) hits[o + t] = (u64)sa[s - 1 + t] + 1;
    u64 longmask = __ballot(len >= LONG);
    while (longmask) {
        int src = __builtin_ctzll(longmask);
        longmask &= longmask - 1;
        u64 ls = __shfl(s, src, 64), ll = __shfl(len, src, 64), lo = __shfl(o, src, 64);
        for (u64 t = lane_id(); t < ll; t += 64)
            if (lo + t < cap) hits[lo + t] = (u64)sa[ls - 1 + t] + 1;
    }
}

#endif  // __HIPCC__

// code of byte / C[code] / count[code] from the byte histogram; returns the number of present byte values
static u32 fm_make_tab(const u32 *counts, u32 *tab, i16 *sym_of_code) {
    u32 sig = 0, acc = 1;
    for (int b = 0; b < 256; b++) {
        tab[b] = 0xFFFFFFFFu;
        if (counts[b]) {
            tab[b] = sig;
            tab[256 + sig] = acc;
            tab[512 + sig] = counts[b];
            if (sym_of_code) sym_of_code[sig] = (i16)b;
            acc += counts[b];
            sig++;
        }
    }
    for (u32 c = sig; c < 256; c++) tab[256 + c] = tab[512 + c] = 0;
    return sig;
}

static void fm_release(tc_fm *fm) {
    if (!fm) return;
    (void)hipSetDevice(fm->device);
    if (fm->d_L) (void)hipFree(fm->d_L);
    if (fm->d_sa) (void)hipFree(fm->d_sa);
    if (fm->d_bits) (void)hipFree(fm->d_bits);
    if (fm->bits2_chunks) tc_chunked_free(fm->bits2_chunks);
    else if (fm->d_bits2) (void)hipFree(fm->d_bits2);
    if (fm->d_tab2) (void)hipFree(fm->d_tab2);
    if (fm->d_tab) (void)hipFree(fm->d_tab);
    delete fm;
}

// the pair vectors of a long text are gigabytes (3.6 bytes per text byte): with TC_FM_VMM = 21 .. 34 and from TC_FM_VMM_MIN_LOG2 (default 2^31 bytes) on
// they are mapped from 2^TC_FM_VMM-byte chunks like the workspace of a long record (default 0: one hipMalloc block -- the experiment
// of round 4: the count rate at 2^30 bytes of text does not depend on how the vectors are mapped)
static void fm_alloc_bits2(tc_ctx *ctx, tc_fm *fm, size_t bytes) {
    const int lg = env_int("TC_FM_VMM", 0);   // (measured, round 4: no gain -- profiles/r04_fm_sweep.txt -- so off unless asked for)
    if (lg >= 21 && lg <= 34 && bytes >= ((size_t)1 << env_int("TC_FM_VMM_MIN_LOG2", 31))) {
        void *p = tc_chunked_alloc(ctx, bytes, lg, &fm->bits2_chunks);
        if (p) {
            fm->d_bits2 = static_cast<u64 *>(p);
            return;
        }
    }
    TC_HIP(ctx, hipMalloc((void **)&fm->d_bits2, bytes));
}

// text_host or text_dev (a text already in HBM is used where it lies: no copy at all)
static tc_fm *fm_build_device(tc_ctx *ctx, const u8 *text_host, u64 n, const u8 *text_dev = nullptr) {
    tc_fm *fm = new tc_fm();
    fm->device = ctx->device;
    fm->n = n;
    fm->N = n + 1;
    try {
        const u64 N = n + 1;
        hipStream_t s = ctx->stream;
        TC_HIP(ctx, hipMalloc((void **)&fm->d_L, N + 16));
        TC_HIP(ctx, hipMalloc((void **)&fm->d_sa, N * sizeof(u32)));
        TC_HIP(ctx, hipMalloc((void **)&fm->d_tab, 768 * sizeof(u32)));
        u8 *d_text = nullptr;
        auto plan = [&](Arena &A, bool dry) {
            if (text_dev) {
                d_text = const_cast<u8 *>(text_dev);
            } else {
                d_text = A.get<u8>(n + 16);
                if (!dry) tc_h2d(ctx, d_text, text_host, n);
            }
            sa_build(ctx, A, d_text, n, fm->d_sa, fm->d_L, &fm->primary, fm->counts, dry);
        };
        ctx->stats = tc_stats{};
        ctx->stats.n = n; ctx->stats.N = N;
        Arena dry(nullptr);
        plan(dry, true);
        tc_ws_reserve(ctx, dry.off);
        Arena A(ctx->ws);
        plan(A, false);
        // C[c] = #symbols of text.'$' smaller than c ('$' = Nothing counts once)
        u32 tab[768];
        const u32 sig = fm_make_tab(fm->counts, tab, fm->sym_of_code);
        fm->sigma_bytes = sig;
        fm->lines = N / FM_LINE_BITS + 1;
        TC_HIP(ctx, hipMalloc((void **)&fm->d_bits, (size_t)sig * fm->lines * 64));
        tc_h2d(ctx, fm->d_tab, tab, sizeof tab);
        TC_HIP(ctx, hipStreamSynchronize(s));  // tab is a stack buffer
        fm_bits_kernel<<<tc_cdiv(fm->lines, 4), 256, 0, s>>>(fm->d_L, N, fm->primary, fm->d_tab, sig,
                                                            fm->lines, fm->d_bits);
        TC_LAUNCH_CHECK(ctx);
        fm_scan_kernel<<<sig, 1024, 0, s>>>(fm->d_bits, fm->lines);
        TC_LAUNCH_CHECK(ctx);
        if (sig <= FM_PAIR_SIGMA && n >= 2 && env_int("TC_FM_PAIRS", 1) != 0) {
            TC_HIP(ctx, hipMalloc((void **)&fm->d_tab2, FM_PAIR_SIGMA * FM_PAIR_SIGMA * sizeof(u32)));
            fm_alloc_bits2(ctx, fm, (size_t)sig * sig * fm->lines * 64);
            TC_HIP(ctx, hipMemsetAsync(fm->d_tab2, 0, FM_PAIR_SIGMA * FM_PAIR_SIGMA * sizeof(u32), s));
            fm_bits2_kernel<<<tc_cdiv(fm->lines, 4), 256, 0, s>>>(fm->d_L, fm->d_sa, d_text, N, fm->d_tab, sig,
                                                                 fm->lines, fm->d_bits2);
            TC_LAUNCH_CHECK(ctx);
            fm_scan_kernel<<<sig * sig, 1024, 0, s>>>(fm->d_bits2, fm->lines);
            TC_LAUNCH_CHECK(ctx);
            fm_c2_kernel<<<1, 64, 0, s>>>(fm->d_bits, fm->lines, fm->d_tab, sig, fm->d_tab2);
            TC_LAUNCH_CHECK(ctx);
        }
        tc_sync_check(ctx);
    } catch (...) {
        fm_release(fm);
        throw;
    }
    return fm;
}

// ranges (device, 2*npat u64) may be null
static void fm_count_device(tc_ctx *ctx, const tc_fm *fm, const u8 *d_pats, const u64 *d_offs,
                            u64 npat, i64 *d_out, u64 *d_ranges) {
    if (fm->d_bits2)
        fm_count_kernel<true><<<tc_cdiv(npat, 256), 256, 0, ctx->stream>>>(fm->d_bits, fm->d_bits2, fm->lines, fm->d_tab,
                                                                           fm->d_tab2, fm->sigma_bytes, d_pats, d_offs,
                                                                           npat, d_out, d_ranges);
    else
        fm_count_kernel<false><<<tc_cdiv(npat, 256), 256, 0, ctx->stream>>>(fm->d_bits, nullptr, fm->lines, fm->d_tab,
                                                                            nullptr, fm->sigma_bytes, d_pats, d_offs,
                                                                            npat, d_out, d_ranges);
    TC_LAUNCH_CHECK(ctx);
}
