/*
 * textcomp.h -- C ABI of libtextcomp.so: the MI355X (gfx950) BWT -> MTF -> RLE /
 * FM-index hot path behind the Data.BWT / Data.MTF / Data.RLE / Data.FMIndex
 * module surface of Matthew-Mosior/text-compression (v0.1.0.25).
 *
 * The reference has no FFI of its own (not one `foreign import`); these entry
 * points are what a Haskell shim's `foreign import ccall safe` would bind in
 * place of the L2 `Seq (Maybe a)` kernels (INTEGRATION.md shows the stubs).
 * Each function cites the reference function(s) it replaces, paths relative to
 * the reference's src/Data/.
 *
 * Conventions
 *   - plain C: pointers and sizes only; no C++/torch types; no exceptions.
 *   - `Maybe Word8` <-> int16_t, -1 = Nothing (order -1 < 0 < .. < 255 equals
 *     `Ord (Maybe Word8)`).  A BWT is `uint8_t L[N]` + `primary` = the slot that
 *     holds Nothing (its byte in L is 0).  N = n + 1; n == 0 <=> N == 0
 *     (BWT.hs:58: empty input gives an empty BWT, no lone sentinel).
 *   - caller allocates and frees every input/output buffer; the library owns only
 *     tc_ctx / tc_fm handles.  Variable-size outputs use in/out capacity words.
 *   - return 0 = TC_OK, negative = error; tc_last_error(ctx) has the text.
 *     TC_ERR_MALFORMED marks inputs on which the reference itself throws
 *     (fromJust / DS.index / read).
 *   - one tc_ctx = one device + one HIP stream + one workspace; calls on one ctx
 *     are serialised by the caller, distinct ctxs are independent.
 *   - `*_dev` entry points take DEVICE pointers for the bulk arrays (the
 *     benchmark path: inputs and outputs resident in HBM); scalar outputs are
 *     host words.  All calls return after the ctx stream has drained.
 *   - limits: n <= TC_MAX_N.  There is NO CPU fallback: without a usable HIP
 *     device every compute call fails with TC_ERR_HIP.
 */
#ifndef TEXTCOMP_H
#define TEXTCOMP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TC_OK 0
#define TC_ERR_ARG (-1)
#define TC_ERR_CAPACITY (-2)
#define TC_ERR_MALFORMED (-3)
#define TC_ERR_HIP (-4)
#define TC_ERR_OOM (-5)
#define TC_ERR_INTERNAL (-6)
#define TC_ERR_NCCL (-7)     /* RCCL missing or failing (tc_comm_* only) */

#define TC_MAX_N ((uint64_t)0x7ffffff0u) /* indices are 31-bit on the device */
#define TC_MAX_SIGMA 257                 /* 256 byte values + Nothing */
#define TC_MAX_ROUNDS 40

typedef struct tc_ctx tc_ctx;
typedef struct tc_fm tc_fm;

/* Work counters of the last encode on this ctx: the inputs of the algorithmic-
 * byte formula of SURVEY.md 8(d) / DESIGN.md, plus per-stage device time. */
typedef struct tc_stats {
    uint64_t n;                      /* input bytes */
    uint64_t N;                      /* n + 1 */
    uint32_t sigma;                  /* present symbols incl. Nothing */
    uint32_t rounds;                 /* suffix-sort rounds executed (round 0 = k-mer sort) */
    uint64_t m[TC_MAX_ROUNDS];       /* suffixes sorted in round r (m[0] = N) */
    uint32_t key_bytes[TC_MAX_ROUNDS];   /* k_r: key bytes moved per element */
    uint32_t passes[TC_MAX_ROUNDS];      /* P_r: radix passes */
    uint32_t h[TC_MAX_ROUNDS];           /* symbols resolved entering round r */
    uint64_t runs;                   /* RLE runs produced */
    float ms_sa, ms_bwt, ms_mtf, ms_rle, ms_total; /* device time, HIP events */
    /* dominant kernel (one radix-sort pass over all N suffixes), timed with HIP
     * events on the ctx stream when tc_ctx_set_profile(ctx, 1) is on */
    uint32_t radix_launches;         /* round-0 pass launches timed */
    float ms_radix;                  /* their summed duration */
    uint32_t keygen_fused;           /* 1: the first pass builds its keys from the text (reads 1 B,
                                        writes 12 B per suffix instead of 12 + 12) */
    uint32_t finish_pass;            /* 1: round 0 = partial sort + finish kernel (12 B read, 5 B written) */
    uint32_t sample_dups;            /* of 8192 sampled suffixes, how many repeated another sample's
                                        globally sorted prefix (> 10 %: full path without trying the
                                        finish pass) */
    uint32_t msd_path;               /* 1: round 0 ran as the MSD partition levels + bucket finish (tc_msd.hpp:
                                        long texts over a small alphabet); radix_launches / ms_radix then time
                                        msd_partition_kernel (first launch reads the text: 1 + 12 B per suffix,
                                        the others 12 + 12 B) */
    uint32_t msd_keyonly;            /* 1: the MSD levels moved keys only (no suffix array was asked for: encode, BWT):
                                        8 + 8 B per suffix per level (the first: 1 + 8) instead of 12 + 12; the suffix starts
                                        of the tied set were found again by one pass over the text */
    uint32_t ticket_fallbacks;       /* suffix sorts of this ctx that had to be redone with the single tile-ticket
                                        counter because a look-back of the XCD-grouped ticket order ran into its
                                        spin limit (LSD passes only; 0 in a healthy run, cumulative per ctx) */
    /* Layout note: fields are only ever APPENDED from round 4 on (round 3 put msd_keyonly in front of
     * ticket_fallbacks: callers built against the round-2 header must be rebuilt). */
    uint32_t ws_chunks;              /* physical chunks the context's workspace is mapped from (0: one hipMalloc block) */
    uint32_t ws_grown;               /* how often that workspace grew in place (more chunks mapped; cumulative per ctx) */
    uint32_t seg_rounds;             /* doubling rounds whose sort was the segmented one (tc_seg.hpp), last suffix sort */
    uint32_t chain_rounds;           /* doubling rounds run as chain rounds (tc_chain.hpp: periodic text), last suffix sort; each is ONE
                                        entry of m[] / h[] with passes[] = 2 (was reserved0: same layout) */
} tc_stats;

/* The encoded block of the fused BWT -> MTF -> RLE pipeline.  The reference has
 * no such container (it never feeds MTF output into RLE); this is
 * the documented glue: RLE runs over the MTF index stream as plain integers, with
 * the BWT primary index and the MTF final list (MTF/Internal.hs:125,140-141)
 * carried in the header. */
typedef struct tc_block {
    uint64_t n;                          /* out: input length */
    uint64_t primary;                    /* out: BWT slot of Nothing */
    uint32_t sigma;                      /* out: MTF alphabet size */
    int16_t final_list[TC_MAX_SIGMA];    /* out: MTF list after the last move */
    uint64_t nruns;                      /* in: capacity of the run arrays; out: runs */
    uint32_t *run_count;                 /* [capacity] run lengths */
    uint16_t *run_value;                 /* [capacity] MTF index of each run */
} tc_block;

/* ---- context ------------------------------------------------------------ */
int tc_ctx_create(int device, tc_ctx **out);
void tc_ctx_destroy(tc_ctx *ctx);
const char *tc_last_error(const tc_ctx *ctx);
const char *tc_version(void);
int tc_get_stats(const tc_ctx *ctx, tc_stats *out);
/* Stream the ctx launches on (a hipStream_t), for callers that time or order
 * work against it. */
void *tc_ctx_stream(const tc_ctx *ctx);
/* on != 0: bracket every round-0 radix pass with HIP events (tc_stats.ms_radix). */
int tc_ctx_set_profile(tc_ctx *ctx, int on);
/* Workspace placement.  Where the context's workspace lands in device memory decides which of two speeds the
 * partition levels of a long record run at (1 GiB ACGTN on MI355X: 28.9 or 31.0 ms per encode, fixed for the
 * life of the workspace).  This call encodes the caller's representative record
 * (arguments as tc_encode_dev; `out` is overwritten) on up to `tries` differently placed workspaces -- a
 * rejected block stays allocated until the call ends, so that the next one lands elsewhere; blocks are only
 * added while device memory has room for them -- and keeps the fastest.  ms (host, [tries], optional)
 * receives the encode time per placement (0 = not tried), *chosen its index.  Not part of the reference's
 * surface: a set-up step for long-lived contexts, before any timed work. */
int tc_ctx_place_workspace(tc_ctx *ctx, const uint8_t *d_text, uint64_t n, tc_block *out, int tries, double *ms,
                           int *chosen);

/* ---- Data.BWT ------------------------------------------------------------ */
/* bytestringToBWT (BWT.hs:68-70) = toBWT (:55-64) = createSuffixArray
 * (BWT/Internal.hs:110-134) + saToBWT (:98-106).  L has n+1 bytes. */
int tc_bwt_encode(tc_ctx *ctx, const uint8_t *text, uint64_t n, uint8_t *L, uint64_t *primary);
int tc_bwt_encode_dev(tc_ctx *ctx, const uint8_t *d_text, uint64_t n, uint8_t *d_L,
                      uint64_t *primary);
/* createSuffixArray alone: sa[j] = 0-based start of the j-th smallest suffix of
 * text.'$' (reference: 1-based suffixstartpos), n+1 entries. */
int tc_suffix_array(tc_ctx *ctx, const uint8_t *text, uint64_t n, uint32_t *sa);

/* bytestringFromWord8BWT (BWT.hs:108-110) = fromBWT (:93-104) + sortTB
 * (BWT/Internal.hs:144-149) + magicInverseBWT (:163-200), for a well-formed BWT
 * (exactly one Nothing at `primary`).  text receives N-1 bytes. */
int tc_bwt_decode(tc_ctx *ctx, const uint8_t *L, uint64_t N, uint64_t primary, uint8_t *text);
/* The same for ANY Seq (Maybe Word8) (zero or several Nothings):
 * n_out = bytes produced (<= N); TC_ERR_MALFORMED where fromJust would throw. */
int tc_bwt_decode_sym(tc_ctx *ctx, const int16_t *sym, uint64_t N, uint8_t *text,
                      uint64_t *n_out);

/* ---- Data.MTF ------------------------------------------------------------ */
/* bytestringBWTToMTFB (MTF.hs:117-122) = seqToMTF (MTF/Internal.hs:128-175):
 * idx[N] 0-based list positions; final_list[sigma] = list AFTER the last move;
 * alphabet = sorted present symbols, Nothing first.  primary < 0: no Nothing. */
int tc_mtf_encode(tc_ctx *ctx, const uint8_t *L, uint64_t N, int64_t primary, uint16_t *idx,
                  int16_t *final_list, uint32_t *sigma);
/* bytestringToMTFB-shaped input (MTF.hs:157-161): any Seq (Maybe Word8). */
int tc_mtf_encode_sym(tc_ctx *ctx, const int16_t *sym, uint64_t N, uint16_t *idx,
                      int16_t *final_list, uint32_t *sigma);
/* bytestringBWTFromMTFB (MTF.hs:240-245) = seqFromMTF (MTF/Internal.hs:201-232):
 * initial list = sort(unique(list)) (:214); out-of-range index => TC_ERR_MALFORMED
 * (DS.index).  N == 0 or nlist == 0 => empty output (:202-209). */
int tc_mtf_decode(tc_ctx *ctx, const uint16_t *idx, uint64_t N, const int16_t *list,
                  uint32_t nlist, int16_t *sym);

/* ---- Data.RLE ------------------------------------------------------------ */
/* bytestringBWTToRLEB (RLE.hs:117-123) = seqToRLE (RLE/Internal.hs:104-153) incl.
 * the sentinel quirks.  Pair k = (counts[k], syms[k]); the Haskell
 * side renders counts with `show` (RLE/Internal.hs:128).  nruns: in capacity
 * (2N is always enough), out pairs written. */
int tc_rle_encode(tc_ctx *ctx, const uint8_t *L, uint64_t N, int64_t primary, uint32_t *counts,
                  int16_t *syms, uint64_t *nruns);
/* bytestringToRLEB-shaped input (RLE.hs:155-159): any Seq (Maybe Word8). */
int tc_rle_encode_sym(tc_ctx *ctx, const int16_t *sym, uint64_t N, uint32_t *counts,
                      int16_t *syms, uint64_t *nruns);
/* Q4b glue: runs of a plain integer stream (the MTF indices; no sentinel). */
int tc_rle_encode_u16(tc_ctx *ctx, const uint16_t *vals, uint64_t N, uint32_t *counts,
                      uint16_t *run_vals, uint64_t *nruns);
/* bytestringBWTFromRLEB (RLE.hs:237-241) = seqFromRLE (RLE/Internal.hs:155-189):
 * (count, Nothing) => one Nothing whatever the count.  N: in capacity, out length. */
int tc_rle_decode(tc_ctx *ctx, const uint32_t *counts, const int16_t *syms, uint64_t nruns,
                  int16_t *sym_out, uint64_t *N);
int tc_rle_decode_u16(tc_ctx *ctx, const uint32_t *counts, const uint16_t *run_vals,
                      uint64_t nruns, uint16_t *vals_out, uint64_t *N);

/* ---- fused pipeline (benchmark path) ------------------------------------- */
/* bytestringToBWT -> bytestringBWTToMTFB -> RLE of the index stream, one call. */
int tc_encode(tc_ctx *ctx, const uint8_t *text, uint64_t n, tc_block *out);
/* d_text and out->run_count / out->run_value are device pointers. */
int tc_encode_dev(tc_ctx *ctx, const uint8_t *d_text, uint64_t n, tc_block *out);
/* Inverse chain: RLE -> MTF -> BWT decode; text receives blk->n bytes. */
int tc_decode(tc_ctx *ctx, const tc_block *blk, uint8_t *text);
int tc_decode_dev(tc_ctx *ctx, const tc_block *blk, uint8_t *d_text);

/* ---- encoded-block wire format (used by the multi-GPU gather) ----- */
/* The reference has no on-disk / wire format.  Packed form of a tc_block's runs, chosen by sigma
 * (values must be < sigma):
 *   sigma <= 6  (an ACGTN record: 5 letters + sentinel) -- nibble stream: code 0..11 starts a run
 *       (value = code % 6, count = 1 + code / 6); a following 12 / 13 raises a count of 2 to 3 / 4;
 *       a following 14 means "count = next uint32 of the escape list" (counts 0 and >= 5, in run
 *       order); 15 is padding (the stream is dense; its end is padded to a 16-byte boundary).  The
 *       escape list (4 bytes each) follows the nibble body.  `packed` must be 16-byte aligned.
 *   sigma <= 16 -- byte k (k < nruns) = value | (min(count, 15) << 4)
 *   sigma > 16  -- two bytes per run: value low byte, then count (escape 127) | ninth value bit << 7
 *       in both byte forms count >= 15 (127) additionally appends the pair (run index, count) as two
 *       uint32 words to the escape list that follows the bytes at the next 8-byte boundary.
 * *packed_bytes: in = capacity of `packed`, out = bytes used (TC_ERR_CAPACITY: bytes needed);
 * tc_block_packed_bound(nruns, sigma) is always enough.  *nesc returns the number of escapes.
 * All pointers are DEVICE pointers. */
uint64_t tc_block_packed_bound(uint64_t nruns, uint32_t sigma);
int tc_block_pack_dev(tc_ctx *ctx, const tc_block *blk, uint8_t *d_packed, uint64_t *packed_bytes,
                      uint64_t *nesc);
/* Inverse: fills blk->run_count / blk->run_value (device, capacity blk->nruns >= nruns) from
 * packed_bytes bytes; a body that does not hold exactly nruns runs and nesc escapes is
 * TC_ERR_MALFORMED. */
int tc_block_unpack_dev(tc_ctx *ctx, const uint8_t *d_packed, uint64_t packed_bytes, uint64_t nruns,
                        uint32_t sigma, uint64_t nesc, tc_block *blk);

/* ---- encoded-block container ------------------------------------------- */
/* One self-describing byte string per record: a TC_CONTAINER_HEADER-byte little-endian header
 * (magic "TCBLK01", n, primary, nruns, escapes, payload bytes, 64-bit payload checksum, sigma,
 * run format id, final MTF list) followed by the packed runs of tc_block_pack_dev.  The reference
 * has no on-disk / wire format; this is what a caller stores or ships.  Buffers must be 16-byte
 * aligned.  *bytes: in = capacity, out = bytes used (TC_ERR_CAPACITY: bytes needed, as far as known;
 * tc_container_bound is always enough).  Reading verifies magic, sizes and the checksum
 * (TC_ERR_MALFORMED). */
#define TC_CONTAINER_HEADER 640
uint64_t tc_container_bound(uint64_t nruns, uint32_t sigma);
int tc_block_to_container_dev(tc_ctx *ctx, const tc_block *blk /* device runs */, uint8_t *d_out, uint64_t *bytes);
/* Text -> container in one call, everything on the device: the bytes of tc_encode_dev followed by
 * tc_block_to_container_dev, without the run arrays in between -- for sigma <= 6 (an ACGTN record) the RLE
 * stage writes the container's nibble stream itself and the container is sealed (escape list, checksum,
 * header) by device kernels.  This is what one step of the multi-GPU path produces and ships
 * (tc_comm_gather).  d_text, d_out: device pointers; d_out 16-byte aligned; *bytes as above
 * (tc_container_bound(n + 2, TC_MAX_SIGMA) is always enough; an ACGTN record needs ~0.45 n). */
int tc_encode_container_dev(tc_ctx *ctx, const uint8_t *d_text, uint64_t n, uint8_t *d_out, uint64_t *bytes);
/* blk->run_count / run_value: device arrays of capacity blk->nruns (TC_ERR_CAPACITY: blk->nruns = needed). */
int tc_container_to_block_dev(tc_ctx *ctx, const uint8_t *d_in, uint64_t bytes, tc_block *blk);
/* Host side: text -> container and back in one call each; only the compact form crosses PCIe.
 * tc_container_info reads n and nruns from a container in HOST memory. */
int tc_encode_container(tc_ctx *ctx, const uint8_t *text, uint64_t n, uint8_t *out, uint64_t *bytes);
int tc_container_info(tc_ctx *ctx, const uint8_t *container, uint64_t bytes, uint64_t *n, uint64_t *nruns);
int tc_decode_container(tc_ctx *ctx, const uint8_t *container, uint64_t bytes, uint8_t *text, uint64_t *n_out);

/* ---- chunked stream (texts longer than one record / than HBM) ------------- */
/* A text of any length is cut into records of block_bytes (0 = TC_STREAM_BLOCK_DEFAULT; at most
 * TC_MAX_N; the last record is the remainder; an empty text is one empty record); every record is
 * encoded on its own -- the BWT is global within a record only, as with bzip2's blocks -- and the
 * stream is the records' containers back to back.  Host buffers; record k+1 is copied in and
 * container k-1 copied out while the device encodes record k.  *bytes: in = capacity of `out`,
 * out = bytes used (TC_ERR_CAPACITY: *bytes = tc_stream_bound, always enough).  A stream of one
 * record is byte-identical to tc_encode_container's output. */
#define TC_STREAM_BLOCK_DEFAULT ((uint64_t)1 << 30)
uint64_t tc_stream_bound(uint64_t n, uint64_t block_bytes);
int tc_encode_stream(tc_ctx *ctx, const uint8_t *text, uint64_t n, uint64_t block_bytes, uint8_t *out,
                     uint64_t *bytes);
/* total text length and number of records of a stream in HOST memory (headers only) */
int tc_stream_info(tc_ctx *ctx, const uint8_t *stream, uint64_t bytes, uint64_t *n_total, uint64_t *nblocks);
/* *n_out: in = capacity of `text`, out = bytes written (TC_ERR_CAPACITY: bytes needed); every
 * container's checksum is verified (TC_ERR_MALFORMED). */
int tc_decode_stream(tc_ctx *ctx, const uint8_t *stream, uint64_t bytes, uint8_t *text, uint64_t *n_out);

/* ---- Data.FMIndex -------------------------------------------------------- */
/* bytestringToBWTToFMIndexB (FMIndex.hs:108-111,162-183): C[c] (seqToCc,
 * FMIndex/Internal.hs:275-316), Occ (seqToOccCK :195-259, kept as rank
 * bit-vectors instead of the full sigma x N table) and the suffix array. */
int tc_fm_build(tc_ctx *ctx, const uint8_t *text, uint64_t n, tc_fm **out);
/* The same index from a text that already lies in HBM (device pointer; read where it is, never copied or modified;
 * it may be released once the call has returned). */
int tc_fm_build_dev(tc_ctx *ctx, const uint8_t *d_text, uint64_t n, tc_fm **out);
void tc_fm_free(tc_fm *fm);
/* bytestringFMIndexCountS / ...CountP (FMIndex.hs:362-379,411-432) =
 * countFMIndex (FMIndex/Internal.hs:347-438) mapped over the patterns in ONE
 * batched launch; pattern j = pats[offs[j] .. offs[j+1]).  out[j] = count, 0 for
 * Nothing (Q10); result order = pattern order. */
int tc_fm_count(tc_ctx *ctx, const tc_fm *fm, const uint8_t *pats, const uint64_t *offs,
                uint64_t npat, int64_t *out);
int tc_fm_count_dev(tc_ctx *ctx, const tc_fm *fm, const uint8_t *d_pats, const uint64_t *d_offs,
                    uint64_t npat, int64_t *d_out);
/* bytestringFMIndexLocateS / ...LocateP (FMIndex.hs:475-497,538-563) =
 * locateFMIndex (FMIndex/Internal.hs:448-542): 1-based text positions in SA
 * order.  hit_offs[npat+1] (out) delimits each pattern's hits inside hits[];
 * *nhits: in capacity, out total hits (TC_ERR_CAPACITY sets the needed total). */
int tc_fm_locate(tc_ctx *ctx, const tc_fm *fm, const uint8_t *pats, const uint64_t *offs,
                 uint64_t npat, uint64_t *hit_offs, uint64_t *hits, uint64_t *nhits);
/* seqToCc / seqFromFMIndex views for the Haskell shim: present symbols (sorted,
 * Nothing first) with C[c]; and L / primary. */
int tc_fm_info(const tc_fm *fm, uint64_t *N, uint32_t *sigma, int16_t *c_sym, uint64_t *c_val,
               uint64_t *primary);

/* Replication of a built index over the GPUs of a node (FM-count shards by pattern batch,
 * the index is broadcast once): the index as ONE device byte string (header, C / code tables, rank
 * bit-vectors; with_locate != 0 adds the last column and the suffix array that tc_fm_locate needs) and
 * back.  The caller moves the bytes (RCCL broadcast); tc_fm_import_dev checks the header
 * (TC_ERR_MALFORMED) and copies out of d_in, which may be released afterwards.  An index imported
 * without the locate part answers tc_fm_count only (tc_fm_locate: TC_ERR_ARG).  Buffers 16-byte
 * aligned.  *bytes: in = capacity, out = bytes used (TC_ERR_CAPACITY: bytes needed).  The byte string is BUILD-SPECIFIC
 * (it carries a format version: "TCFMI02" since round 3; an export of another version is refused with a message that
 * says so): it travels between the ranks of one job, it is not an archive format. */
uint64_t tc_fm_export_bound(const tc_fm *fm, int with_locate);
int tc_fm_export_dev(tc_ctx *ctx, const tc_fm *fm, int with_locate, uint8_t *d_out, uint64_t *bytes);
int tc_fm_import_dev(tc_ctx *ctx, const uint8_t *d_in, uint64_t bytes, tc_fm **out);

/* ---- the exchange of the multi-GPU path ------------------------------------------ */
/* One process per GPU, one tc_ctx per process, one record per GPU: nothing is exchanged during the
 * encode.  A tc_comm is an RCCL communicator over the ranks of the job, bound at run time (dlopen: a
 * process that never calls tc_comm_* never loads RCCL; failures are TC_ERR_NCCL).  Rank 0 obtains an id
 * with tc_comm_unique_id and hands its TC_COMM_ID_BYTES bytes to the other ranks by whatever channel
 * started them (environment, file, MPI ...); then every rank calls tc_comm_create (collective).
 *
 * tc_comm_gather (collective): the variable-size gather of one container (tc_block_to_container_dev)
 * per rank on `root`.  Sizes travel by an all-gather of one word per rank (sizes[world], host, out on
 * every rank); then the root receives rank r's bytes at d_recv + r * slot_bytes (its own container is
 * copied there too) in ONE group of point-to-point transfers -- it ingests on all of its xGMI links at
 * once.  A size above slot_bytes is TC_ERR_CAPACITY on EVERY rank (before anything is sent).  The call
 * returns once the transfers are POSTED on the communicator's own stream: the gather of record k overlaps
 * the encode of record k + 1; d_container and d_recv belong to the exchange until tc_comm_wait returns.
 * tc_comm_broadcast (collective, complete on return): `bytes` of d_buf from root to all (an exported
 * FM-index, tc_fm_export_dev -> tc_fm_import_dev).  All buffers are device pointers. */
#define TC_COMM_ID_BYTES 128
typedef struct tc_comm tc_comm;
int tc_comm_unique_id(tc_ctx *ctx, uint8_t *id /* [TC_COMM_ID_BYTES] */);
int tc_comm_create(tc_ctx *ctx, const uint8_t *id, int rank, int world, tc_comm **out);
void tc_comm_destroy(tc_comm *comm);
int tc_comm_gather(tc_comm *comm, int root, const uint8_t *d_container, uint64_t bytes, uint8_t *d_recv,
                   uint64_t slot_bytes, uint64_t *sizes);
int tc_comm_wait(tc_comm *comm);
/* The exchange overlaps the next record's encode; to keep RCCL's workgroups from holding back the encode's
 * partition levels (which want whole CUs), the communicator's stream is restricted to TC_COMM_CUS compute units
 * (environment; default 8 when world > 1, one per XCD; 0: unrestricted) and the context's partition levels
 * split their work over the others.  Returns how many CUs this communicator is restricted to (0: none). */
int tc_comm_reserved_cus(const tc_comm *comm);
int tc_comm_broadcast(tc_comm *comm, int root, uint8_t *d_buf, uint64_t bytes);

/* ---- synthetic inputs, generated on the device ------------- */
/* kind 0: iid ACGTN, kind 1: printable ASCII; the classes away from iid text that bench.py reports beside the headline
 * (round 4; every byte a function of (kind, seed, position) alone): 2 genome-like (iid ACGT with a 300-bp repeat family in
 * ~10 % of the sequence, poly-A tracts, (CA)n), 3 Zipf-distributed words of 2 .. 9 letters from a 20 000-word vocabulary,
 * 4 runs (a letter repeats with probability 0.9), 5 a 4096-byte block repeated, 6 an assembly with gaps (iid ACGT, one run of
 * n / 64 'N's and sixteen of n / 4096).  d_out is a device pointer. */
int tc_generate_dev(tc_ctx *ctx, int kind, uint64_t seed, uint64_t n, uint8_t *d_out);

#ifdef __cplusplus
}
#endif
#endif /* TEXTCOMP_H */
