/*
 * textcomp_debug.h -- measurement hooks of libtextcomp.so (not part of the drop-in
 * boundary): calibration micro-kernels used to put the pipeline's kernels next to what
 * the memory system delivers on this device for the same access width.
 */
#ifndef TEXTCOMP_DEBUG_H
#define TEXTCOMP_DEBUG_H
#include "textcomp.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The container's position-dependent 64-bit checksum (checksum64_kernel) of any device range
 * (4-byte aligned, length a multiple of 4): lets a test compare full-size device outputs with
 * a digest computed once by the CPU oracle (tests/golden/c3_digest.json). */
int tc_dbg_checksum64_dev(tc_ctx *ctx, const void *d_p, uint64_t bytes, uint64_t *out);
/* Streams `bytes` from one workspace buffer to another `iters` times and returns the
 * mean copy rate in GB/s (read + write bytes / time).  width: bytes per lane per access
 * (1, 2, 4, 8, 16).  mode 0: copy, 1: read-only (sum), 2: write-only (fill). */
int tc_dbg_stream_bench(tc_ctx *ctx, uint64_t bytes, int width, int mode, int iters, double *gbps);
/* Sorts n random (u64 key, u32 value) pairs by the top `key_bits` key bits with the
 * device radix sort and returns the mean duration of one pass in ms (HIP events).
 * check != 0: verify the result is sorted and stable (returns TC_ERR_INTERNAL if not). */
int tc_dbg_sort_bench(tc_ctx *ctx, uint64_t n, int key_bits, int iters, int check, double *ms_per_pass);
/* The memory pattern of one radix pass alone: 4096-pair tiles read coalesced, written as `bins`
 * segments per tile, each behind the same segment of the previous tile; xrun > 0: blocks on the
 * same XCD take tiles in runs of xrun (the pass's XCD-aware order); mean ms per pass. */
int tc_dbg_scatter_bench(tc_ctx *ctx, uint64_t n, uint32_t bins, uint32_t xrun, int iters, double *ms_per_pass);
/* Where the hardware puts the workgroups of a grid launched on this context's stream: `grid` workgroups of 1024
 * threads with `lds_bytes` of LDS each (147456: one per CU) spin for `spin_cycles`; out6[6 * grid] (host)
 * receives per workgroup: XCC id, HW_ID register, start (2 words, 100 MHz wall clock), duration, scratch. */
int tc_dbg_dispatch_probe(tc_ctx *ctx, uint32_t grid, uint32_t lds_bytes, uint32_t spin_cycles, uint32_t *out6);
#ifdef __cplusplus
}
#endif
#endif
