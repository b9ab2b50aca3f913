/*
 * tc_oracle.h -- CPU restatement ("oracle") of the reference's BWT / MTF / RLE /
 * FM-index semantics.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * or call anything in oracle/.  The product library (libtextcomp.so) never links,
 * loads or calls it.
 *
 * Parity status: PINNED by the reference's own known-answer tests
 *   - RLE.hs:279-319  (s1<->rle1, s2<->rle2, four HUnit cases)
 *   - MTF.hs:287-299  ("aaabbbccc" <-> MTF indices + final list, two HUnit cases)
 *   - FMIndex/Internal.hs:49-113 (abracadabra doc tables: L, C[c], Occ(c,k))
 * transcribed as data into tests/golden/ and checked by tests/test_oracle_golden.py.
 * NOT pinned by any reference test (reference ships none): count/locate results
 * and the MTF->RLE composition.  The Haskell reference itself
 * cannot be built here (no GHC in the image), so there is no oracle/_ref.
 *
 * Symbol model: `Maybe Word8` <-> int16_t, -1 = Nothing, 0..255 = Just byte.
 * Order -1 < 0 < ... < 255 equals `Ord (Maybe Word8)`.
 */
#ifndef TC_ORACLE_H
#define TC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_OK 0
#define ORC_ERR_MALFORMED (-3) /* the reference would throw (fromJust / index / read) */

/* ---- Data.BWT.Internal / Data.BWT --------------------------------------- */

/* createSuffixArray (BWT/Internal.hs:110-134), literal restatement: comparison
 * sort of all n+1 suffixes (the empty one included), unsigned bytes, a proper
 * prefix sorts first.  sa[j] = 0-based start (reference: 1-based).  O(n^2 log n)
 * worst case -- small inputs only. */
void orc_suffix_array_naive(const uint8_t *t, int64_t n, int32_t *sa);

/* Same result (all keys distinct => order-determined), by prefix
 * doubling with radix sorts; O(n log n).  Used for large parity cases and as the
 * "port" CPU baseline. */
void orc_suffix_array(const uint8_t *t, int64_t n, int32_t *sa);

/* saToBWT (BWT/Internal.hs:98-106): L[j] = t[sa[j]-1], Nothing where sa[j]==0. */
void orc_sa_to_bwt(const uint8_t *t, int64_t n, const int32_t *sa, int16_t *L);

/* toBWT / bytestringToBWT (BWT.hs:55-70). Returns N = n+1, or 0 for n == 0
 * (BWT.hs:58: empty input => empty BWT, no lone sentinel). */
int64_t orc_bwt_encode(const uint8_t *t, int64_t n, int16_t *L);

/* fromBWT + sortTB + magicInverseBWT (BWT.hs:93-104, BWT/Internal.hs:144-200).
 * Generic over any Seq (Maybe Word8): zero, one or several Nothings.
 * Returns output length, or ORC_ERR_MALFORMED where fromJust would throw. */
int64_t orc_bwt_decode(const int16_t *L, int64_t N, uint8_t *out);

/* ---- Data.MTF.Internal ------------------------------------------------- */

/* seqToMTF (MTF/Internal.hs:128-175) with nubSeq' (:79-99).  idx[N] 0-based
 * positions; final_list[sigma] = list AFTER the last move (Q3).  Returns sigma. */
int32_t orc_mtf_encode(const int16_t *x, int64_t N, int32_t *idx, int16_t *final_list);

/* seqFromMTF (MTF/Internal.hs:201-232): initial list = sort(unique(list)).
 * Returns N, 0 if either part is empty, ORC_ERR_MALFORMED on index out of range. */
int64_t orc_mtf_decode(const int32_t *idx, int64_t N, const int16_t *list, int32_t nlist,
                       int16_t *out);

/* ---- Data.RLE.Internal ------------------------------------------------- */

/* seqToRLE (RLE/Internal.hs:104-153) incl. sentinel quirks Q5-Q7.  Output as
 * pairs: counts[k] (rendered `show count` by the caller), syms[k].  Capacity
 * needed: at most 2N+1 pairs... (N+1 is enough).  Returns #pairs. */
int64_t orc_rle_encode(const int16_t *x, int64_t N, int64_t *counts, int16_t *syms);

/* seqFromRLE (RLE/Internal.hs:155-189), pairwise; (Just _, Nothing) => one
 * Nothing whatever the count; count <= 0 replicates nothing.  Returns length
 * (call with out == NULL to size). */
int64_t orc_rle_decode(const int64_t *counts, const int16_t *syms, int64_t npairs, int16_t *out);

/* Q4b glue (no reference function exists): RLE of the MTF index stream as plain
 * integers -- no sentinel in that stream.  Returns #runs. */
int64_t orc_rle_encode_u32(const int32_t *x, int64_t N, int64_t *counts, int32_t *vals);

/* ---- Data.FMIndex.Internal --------------------------------------------- */

/* seqToCc (FMIndex/Internal.hs:275-316) over the F column (sorted L): one row
 * per present symbol (Nothing first).  c_sym[sigma], c_val[sigma]. Returns sigma. */
int32_t orc_fm_cc(const int16_t *L, int64_t N, int16_t *c_sym, int64_t *c_val);

/* seqToOccCK (FMIndex/Internal.hs:195-259): full table, row r = r-th present
 * symbol (sorted, Nothing first), occ[r*N + (k-1)] = inclusive count, k=1..N. */
void orc_fm_occ(const int16_t *L, int64_t N, int32_t sigma, const int16_t *c_sym, int32_t *occ);

/* countFMIndex (FMIndex/Internal.hs:347-438) incl. Q10.  Returns count, 0 for
 * Nothing.  Works from L directly (Occ by scanning with checkpoints). */
typedef struct orc_fm orc_fm;
orc_fm *orc_fm_build(const uint8_t *t, int64_t n);
void orc_fm_free(orc_fm *f);
int64_t orc_fm_count(const orc_fm *f, const uint8_t *pat, int64_t m);
/* the same over a batch (offs[npat + 1] into pats); thread-safe on a built index */
void orc_fm_count_batch(const orc_fm *f, const uint8_t *pats, const int64_t *offs, int64_t npat,
                        int64_t *out);
/* locateFMIndex (:448-542) + FMIndex.hs:496: 1-based text positions in SA order.
 * Returns number of hits written (<= cap). */
int64_t orc_fm_locate(const orc_fm *f, const uint8_t *pat, int64_t m, int64_t *out, int64_t cap);

/* ---- synthetic inputs -------------------------------------- */
void orc_gen_acgtn(uint64_t seed, int64_t n, uint8_t *out);
void orc_gen_ascii(uint64_t seed, int64_t n, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
