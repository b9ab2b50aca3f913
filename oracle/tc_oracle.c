/*
 * tc_oracle.c -- CPU restatement of the reference (Matthew-Mosior/text-compression
 * v0.1.0.25) for the BWT -> MTF -> RLE / FM-index path.  TEST INFRASTRUCTURE ONLY:
 * see tc_oracle.h for who may call this and for the parity-pinning statement.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference root).  Plain C, no dependencies.
 */
#include "tc_oracle.h"

#include <stdlib.h>
#include <string.h>

/* ======================================================================= */
/* Data.BWT.Internal                                                        */
/* ======================================================================= */

/* Ord (Seq Word8): lexicographic, unsigned bytes, proper prefix first
 * (what `DS.unstableSortOn snd` compares, BWT/Internal.hs:130). */
static const uint8_t *g_t;
static int64_t g_n;
static int cmp_suffix(const void *pa, const void *pb) {
    int32_t a = *(const int32_t *)pa, b = *(const int32_t *)pb;
    if (a == b) return 0;
    int64_t la = g_n - a, lb = g_n - b;
    int64_t l = la < lb ? la : lb;
    int c = memcmp(g_t + a, g_t + b, (size_t)l);
    if (c) return c;
    return la < lb ? -1 : 1; /* shorter (prefix) first */
}

/* createSuffixArray, BWT/Internal.hs:110-134: DS.tails gives n+1 suffixes incl.
 * the empty one (:127); they are sorted by content (:130). */
void orc_suffix_array_naive(const uint8_t *t, int64_t n, int32_t *sa) {
    for (int64_t i = 0; i <= n; i++) sa[i] = (int32_t)i;
    g_t = t;
    g_n = n;
    qsort(sa, (size_t)(n + 1), sizeof(int32_t), cmp_suffix);
}

/* LSD radix sort of (key, val) by 32-bit key, 16 bits per pass. */
static void radix32(uint32_t *key, int32_t *val, uint32_t *tk, int32_t *tv, int64_t m,
                    uint32_t maxkey) {
    int passes = maxkey >> 16 ? 2 : 1;
    int64_t *cnt = (int64_t *)malloc(65537 * sizeof(int64_t));
    for (int p = 0; p < passes; p++) {
        int sh = 16 * p;
        memset(cnt, 0, 65537 * sizeof(int64_t));
        for (int64_t i = 0; i < m; i++) cnt[((key[i] >> sh) & 0xffff) + 1]++;
        for (int i = 0; i < 65536; i++) cnt[i + 1] += cnt[i];
        for (int64_t i = 0; i < m; i++) {
            int64_t d = cnt[(key[i] >> sh) & 0xffff]++;
            tk[d] = key[i];
            tv[d] = val[i];
        }
        uint32_t *sk = key; key = tk; tk = sk;
        int32_t *sv = val; val = tv; tv = sv;
    }
    if (passes & 1) { /* result sits in the caller's tmp arrays: copy back */
        memcpy(tk, key, (size_t)m * sizeof(uint32_t));
        memcpy(tv, val, (size_t)m * sizeof(int32_t));
    }
    free(cnt);
}

/* Same order as orc_suffix_array_naive (the total order alone fixes the result:
 * all n+1 suffixes are distinct), by Manber-Myers prefix doubling.  rank of a
 * position past the end never occurs: suffix i with h matched symbols has
 * i+h <= n, and position n (the empty suffix) has the unique smallest rank. */
void orc_suffix_array(const uint8_t *t, int64_t n, int32_t *sa) {
    int64_t N = n + 1;
    uint32_t *rank = (uint32_t *)malloc((size_t)N * sizeof(uint32_t));
    uint32_t *k1 = (uint32_t *)malloc((size_t)N * sizeof(uint32_t));
    uint32_t *tk = (uint32_t *)malloc((size_t)N * sizeof(uint32_t));
    int32_t *tv = (int32_t *)malloc((size_t)N * sizeof(int32_t));
    /* round 0: key = first 3 symbols, 0 = end marker ($ < every byte), byte+1 else */
    for (int64_t i = 0; i < N; i++) {
        uint32_t k = 0;
        for (int j = 0; j < 3; j++) k = k * 257u + (i + j < n ? (uint32_t)t[i + j] + 1u : 0u);
        k1[i] = k;
        sa[i] = (int32_t)i;
    }
    radix32(k1, sa, tk, tv, N, 257u * 257u * 257u);
    /* rank = index of first member of the equal-key group */
    int64_t groups = 0;
    for (int64_t j = 0; j < N; j++) {
        if (j == 0 || k1[j] != k1[j - 1]) groups++;
        rank[sa[j]] = (uint32_t)((j == 0 || k1[j] != k1[j - 1]) ? j : rank[sa[j - 1]]);
    }
    uint32_t *k2 = (uint32_t *)malloc((size_t)N * sizeof(uint32_t));
    for (int64_t h = 3; groups < N; h *= 2) {
        /* sort by (rank[i], rank[i+h]): LSD -- second key first, then first key */
        for (int64_t j = 0; j < N; j++) {
            int64_t i = sa[j];
            k2[j] = (i + h < N) ? rank[i + h] : 0u; /* unreachable for tied i, see above */
        }
        radix32(k2, sa, tk, tv, N, (uint32_t)N);
        for (int64_t j = 0; j < N; j++) k1[j] = rank[sa[j]];
        radix32(k1, sa, tk, tv, N, (uint32_t)N);
        /* re-rank; k1 = old rank (sorted), need second key again for equality */
        groups = 0;
        uint32_t prev1 = 0, prev2 = 0, cur = 0;
        for (int64_t j = 0; j < N; j++) {
            int64_t i = sa[j];
            uint32_t a = k1[j], b = (i + h < N) ? rank[i + h] : 0u;
            if (j == 0 || a != prev1 || b != prev2) { groups++; cur = (uint32_t)j; }
            tk[j] = cur;
            prev1 = a; prev2 = b;
        }
        for (int64_t j = 0; j < N; j++) rank[sa[j]] = tk[j];
    }
    free(k2); free(tv); free(tk); free(k1); free(rank);
}

/* saToBWT, BWT/Internal.hs:98-106: startpos /= 1 => Just t[startpos-2] (1-based),
 * else Nothing. */
void orc_sa_to_bwt(const uint8_t *t, int64_t n, const int32_t *sa, int16_t *L) {
    for (int64_t j = 0; j <= n; j++) L[j] = sa[j] == 0 ? (int16_t)-1 : (int16_t)t[sa[j] - 1];
}

/* toBWT, BWT.hs:55-64. */
int64_t orc_bwt_encode(const uint8_t *t, int64_t n, int16_t *L) {
    if (n == 0) return 0; /* BWT.hs:58 */
    int32_t *sa = (int32_t *)malloc((size_t)(n + 1) * sizeof(int32_t));
    orc_suffix_array(t, n, sa);
    orc_sa_to_bwt(t, n, sa, L);
    free(sa);
    return n + 1;
}

/* fromBWT, BWT.hs:93-104: zip with 0-based positions, sort by (symbol, position)
 * (sortTB, BWT/Internal.hs:144-149) == stable counting sort by symbol, Nothing
 * first; then magicInverseBWT, BWT/Internal.hs:163-200. */
int64_t orc_bwt_decode(const int16_t *L, int64_t N, uint8_t *out) {
    if (N == 0) return 0; /* :164-167 */
    int64_t cnt[258];
    memset(cnt, 0, sizeof cnt);
    for (int64_t j = 0; j < N; j++) cnt[L[j] + 1 + 1]++;
    for (int s = 0; s < 257; s++) cnt[s + 1] += cnt[s];
    int16_t *ssym = (int16_t *)malloc((size_t)N * sizeof(int16_t));
    int64_t *spos = (int64_t *)malloc((size_t)N * sizeof(int64_t));
    for (int64_t j = 0; j < N; j++) {
        int64_t d = cnt[L[j] + 1]++;
        ssym[d] = L[j];
        spos[d] = j;
    }
    int64_t len = 0, rc = 0;
    /* :172 findIndexL isNothing: sorted order puts every Nothing first */
    if (ssym[0] != -1) { rc = 0; goto done; } /* :173-174: no Nothing => empty */
    {
        int64_t e = 0;       /* :177 nothingindex */
        int64_t f = spos[0]; /* :179 snd nothingfirst */
        while (f != e) {     /* iBWT :189-200 */
            if (ssym[f] == -1) { rc = ORC_ERR_MALFORMED; goto done; } /* fromJust :195 */
            out[len++] = (uint8_t)ssym[f];
            f = spos[f];
        }
        rc = len;
    }
done:
    free(spos);
    free(ssym);
    return rc;
}

/* ======================================================================= */
/* Data.MTF.Internal                                                        */
/* ======================================================================= */

/* nubSeq', MTF/Internal.hs:79-99: unique elements, then unstableSort
 * (Nothing < Just 0 < ... < Just 255). */
static int32_t nub_sorted(const int16_t *x, int64_t N, int16_t *list) {
    uint8_t seen[257];
    memset(seen, 0, sizeof seen);
    for (int64_t j = 0; j < N; j++) seen[x[j] + 1] = 1;
    int32_t s = 0;
    for (int v = 0; v < 257; v++)
        if (seen[v]) list[s++] = (int16_t)(v - 1);
    return s;
}

/* seqToMTF, MTF/Internal.hs:128-175; per symbol findIndexL (:152,:165) then
 * updateSTMTFLSSeq (:117-125): emit i, delete at i, cons at front. */
int32_t orc_mtf_encode(const int16_t *x, int64_t N, int32_t *idx, int16_t *final_list) {
    if (N == 0) return 0; /* :129-132 */
    int16_t list[257];
    int32_t sigma = nub_sorted(x, N, list); /* :137 */
    for (int64_t j = 0; j < N; j++) {
        int32_t p = 0;
        while (list[p] != x[j]) p++;
        idx[j] = p;
        int16_t h = list[p];
        memmove(list + 1, list, (size_t)p * sizeof(int16_t));
        list[0] = h;
    }
    memcpy(final_list, list, (size_t)sigma * sizeof(int16_t));
    return sigma;
}

/* seqFromMTF, MTF/Internal.hs:201-232: il = nubSeq' (snd xss) (:214); per index
 * DS.index (throws when out of range) then move to front (:192-199). */
int64_t orc_mtf_decode(const int32_t *idx, int64_t N, const int16_t *list_in, int32_t nlist,
                       int16_t *out) {
    if (N == 0 || nlist == 0) return 0; /* :202-209 */
    int16_t list[257];
    int32_t sigma = nub_sorted(list_in, nlist, list);
    for (int64_t j = 0; j < N; j++) {
        int32_t p = idx[j];
        if (p < 0 || p >= sigma) return ORC_ERR_MALFORMED;
        int16_t h = list[p];
        out[j] = h;
        memmove(list + 1, list, (size_t)p * sizeof(int16_t));
        list[0] = h;
    }
    return N;
}

/* ======================================================================= */
/* Data.RLE.Internal                                                        */
/* ======================================================================= */

/* seqToRLE, RLE/Internal.hs:104-153.  A literal walk of iRLE's four branches. */
int64_t orc_rle_encode(const int16_t *x, int64_t N, int64_t *counts, int16_t *syms) {
    if (N == 0) return 0; /* :105-108 */
    int64_t k = 0;
    int64_t count = 1; /* :113 */
    int16_t item = x[0]; /* :114 */
    for (int64_t j = 1; j < N; j++) {
        int16_t y = x[j];
        if (y == -1) { /* :134-140 */
            counts[k] = count; syms[k] = item; k++;
            counts[k] = 1; syms[k] = -1; k++;
            item = -1; /* count is NOT reset: Q6 stale count */
        } else if (item == -1) { /* :141-144 */
            count = 1; item = y;
        } else if (item == y) { /* :145-147 */
            count++;
        } else { /* :148-153 */
            counts[k] = count; syms[k] = item; k++;
            count = 1; item = y;
        }
    }
    counts[k] = count; syms[k] = item; k++; /* :125-130 end-of-input flush */
    return k;
}

/* seqFromRLE, RLE/Internal.hs:155-189. */
int64_t orc_rle_decode(const int64_t *counts, const int16_t *syms, int64_t npairs, int16_t *out) {
    int64_t len = 0;
    for (int64_t k = 0; k < npairs; k++) {
        if (syms[k] == -1) { /* isJust y1 && isNothing y2 (:168-170,:177-179) */
            if (out) out[len] = -1;
            len++;
        } else { /* replicateM_ count (:174,:184): count <= 0 => nothing */
            for (int64_t r = 0; r < counts[k]; r++) {
                if (out) out[len] = syms[k];
                len++;
            }
        }
    }
    return len;
}

/* Q4b glue: plain run-length pairs over integers (no sentinel can occur). */
int64_t orc_rle_encode_u32(const int32_t *x, int64_t N, int64_t *counts, int32_t *vals) {
    int64_t k = 0;
    for (int64_t j = 0; j < N; j++) {
        if (k > 0 && vals[k - 1] == x[j]) counts[k - 1]++;
        else { vals[k] = x[j]; counts[k] = 1; k++; }
    }
    return k;
}

/* ======================================================================= */
/* Data.FMIndex.Internal                                                    */
/* ======================================================================= */

/* seqToCc, FMIndex/Internal.hs:275-316, applied to the F column (FMIndex.hs:
 * 176-181 = first symbol of each sorted rotation = sorted multiset of L): for each
 * present symbol (nubSeq', sorted) the 0-based index of its first occurrence in F. */
int32_t orc_fm_cc(const int16_t *L, int64_t N, int16_t *c_sym, int64_t *c_val) {
    int64_t cnt[257];
    memset(cnt, 0, sizeof cnt);
    for (int64_t j = 0; j < N; j++) cnt[L[j] + 1]++;
    int32_t s = 0;
    int64_t acc = 0;
    for (int v = 0; v < 257; v++) {
        if (cnt[v]) {
            c_sym[s] = (int16_t)(v - 1);
            c_val[s] = acc;
            s++;
        }
        acc += cnt[v];
    }
    return s;
}

/* seqToOccCK, FMIndex/Internal.hs:195-259: per present symbol a row of
 * (k, Occ(c,k), L[k]) for k = 1..N with Occ INCLUSIVE of position k (:233-248). */
void orc_fm_occ(const int16_t *L, int64_t N, int32_t sigma, const int16_t *c_sym, int32_t *occ) {
    for (int32_t r = 0; r < sigma; r++) {
        int32_t c = 0;
        for (int64_t k = 0; k < N; k++) {
            if (L[k] == c_sym[r]) c++;
            occ[(int64_t)r * N + k] = c;
        }
    }
}

struct orc_fm {
    int64_t n, N;
    int16_t *L;
    int32_t *sa;
    int32_t sigma;
    int16_t c_sym[257];
    int64_t c_val[257];
    int32_t row_of[257]; /* symbol+1 -> row or -1 */
    int32_t *ckpt;       /* [sigma][N/64+1] counts before block */
};

#define ORC_CK 64

orc_fm *orc_fm_build(const uint8_t *t, int64_t n) {
    orc_fm *f = (orc_fm *)calloc(1, sizeof(orc_fm));
    f->n = n;
    if (n == 0) return f; /* FMIndex.hs:366: empty input => empty results */
    f->N = n + 1;
    f->L = (int16_t *)malloc((size_t)f->N * sizeof(int16_t));
    f->sa = (int32_t *)malloc((size_t)f->N * sizeof(int32_t));
    orc_suffix_array(t, n, f->sa);
    orc_sa_to_bwt(t, n, f->sa, f->L);
    f->sigma = orc_fm_cc(f->L, f->N, f->c_sym, f->c_val);
    for (int v = 0; v < 257; v++) f->row_of[v] = -1;
    for (int32_t r = 0; r < f->sigma; r++) f->row_of[f->c_sym[r] + 1] = r;
    int64_t nb = f->N / ORC_CK + 1;
    f->ckpt = (int32_t *)calloc((size_t)(f->sigma * nb), sizeof(int32_t));
    int32_t run[257];
    memset(run, 0, sizeof run);
    for (int64_t k = 0; k < f->N; k++) {
        if (k % ORC_CK == 0)
            for (int32_t r = 0; r < f->sigma; r++) f->ckpt[r * nb + k / ORC_CK] = run[r];
        run[f->row_of[f->L[k] + 1]]++;
    }
    /* Occ(c, N) is asked for (step 0 of a pattern ending in the largest symbol has e = N): when N is
     * a multiple of the checkpoint distance the block of k = N starts AT N and needs its checkpoint too */
    if (f->N % ORC_CK == 0)
        for (int32_t r = 0; r < f->sigma; r++) f->ckpt[r * nb + f->N / ORC_CK] = run[r];
    return f;
}

void orc_fm_free(orc_fm *f) {
    if (!f) return;
    free(f->L); free(f->sa); free(f->ckpt); free(f);
}

/* Occ(c,k): occurrences of row r's symbol in L[1..k] (1-based, inclusive). */
static int64_t occ_at(const orc_fm *f, int32_t r, int64_t k) {
    int64_t nb = f->N / ORC_CK + 1;
    int64_t b = k / ORC_CK;
    int64_t c = f->ckpt[r * nb + b];
    for (int64_t j = b * ORC_CK; j < k; j++) c += (f->L[j] == f->c_sym[r]);
    return c;
}

/* The shared backward-search loop of countFMIndex (:372-438) and locateFMIndex
 * (:473-542).  Returns 1 and [s,e] (1-based inclusive) when the result is a
 * non-empty range, 0 for Nothing/Empty. */
static int fm_range(const orc_fm *f, const uint8_t *pat, int64_t m, int64_t *ps, int64_t *pe) {
    if (m == 0 || f->n == 0) return 0; /* :348-351 */
    int64_t s = -1, e = -1;
    int counter = 0, flag = 0;
    for (int64_t q = m - 1; q >= 0; q--) { /* (as :|> a): right to left (:375) */
        if (s > e) { flag = 1; break; }    /* :387-389 */
        int32_t r = f->row_of[(int)pat[q] + 1];
        if (r < 0) break; /* findIndexL = Nothing => pure (): loop stops (:393,:421) */
        if (counter == 0) { /* :391-418 */
            s = f->c_val[r] + 1;
            e = (r == f->sigma - 1) ? f->N : f->c_val[r + 1]; /* :394-399 / :407-408 */
            counter = 1;
        } else { /* :424-432 */
            int64_t ns = f->c_val[r] + occ_at(f, r, s - 1) + 1;
            int64_t ne = f->c_val[r] + occ_at(f, r, e);
            s = ns; e = ne;
        }
    }
    if ((s == -1 && e == -1) || (e - s + 1) == 0 || flag) return 0; /* :366-369 */
    *ps = s; *pe = e;
    return 1;
}

int64_t orc_fm_count(const orc_fm *f, const uint8_t *pat, int64_t m) {
    int64_t s, e;
    if (!fm_range(f, pat, m, &s, &e)) return 0;
    return e - s + 1; /* :370-371 */
}

/* bytestringFMIndexCountS / ...CountP (FMIndex.hs:362-379,411-432): countFMIndex mapped over the
 * pattern list, result order = pattern order; 0 stands for Nothing.  Reads the index only, so
 * callers may run slices of one batch on several threads (what parListChunk does, :417-423). */
void orc_fm_count_batch(const orc_fm *f, const uint8_t *pats, const int64_t *offs, int64_t npat,
                        int64_t *out) {
    for (int64_t j = 0; j < npat; j++) out[j] = orc_fm_count(f, pats + offs[j], offs[j + 1] - offs[j]);
}

int64_t orc_fm_locate(const orc_fm *f, const uint8_t *pat, int64_t m, int64_t *out, int64_t cap) {
    int64_t s, e;
    if (!fm_range(f, pat, m, &s, &e)) return 0;
    int64_t k = 0;
    for (int64_t x = s; x <= e && k < cap; x++) /* FMIndex.hs:496: suffixstartpos (sa[x-1]) */
        out[k++] = (int64_t)f->sa[x - 1] + 1;
    return k;
}

/* ======================================================================= */
/* synthetic inputs, SURVEY.md 8(d)                                        */
/* ======================================================================= */
static inline uint64_t splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

void orc_gen_acgtn(uint64_t seed, int64_t n, uint8_t *out) {
    static const char A[5] = {'A', 'C', 'G', 'T', 'N'};
    for (int64_t i = 0; i < n; i++) {
        uint64_t x = splitmix64_at(seed, (uint64_t)i);
        out[i] = (uint8_t)A[(uint32_t)(((x >> 32) * 5) >> 32)];
    }
}

void orc_gen_ascii(uint64_t seed, int64_t n, uint8_t *out) {
    for (int64_t i = 0; i < n; i++) {
        uint64_t x = splitmix64_at(seed, (uint64_t)i);
        out[i] = (uint8_t)(0x20 + (uint32_t)(((x >> 32) * 95) >> 32));
    }
}
