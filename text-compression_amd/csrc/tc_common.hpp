// tc_common.hpp -- shared host/device plumbing of libtextcomp (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <string>
#include <vector>

#include "textcomp.h"

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int16_t i16;
typedef int32_t i32;
typedef int64_t i64;

#define TC_WAVE 64

// ------------------------------------------------------------------ context
struct tc_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    char *ws = nullptr;  // device workspace arena
    size_t ws_cap = 0;
    size_t ws_off = 0;
    // workspace built from separately created physical chunks mapped into one reserved address range
    // (TC_WS_VMM = log2 of the chunk size; 0 = one hipMalloc block): empty unless that way was taken
    std::vector<hipMemGenericAllocationHandle_t> ws_chunks;
    size_t ws_chunk_bytes = 0, ws_mapped = 0;
    size_t ws_reserved = 0;   // address range reserved for such a workspace (it grows by mapping more chunks: nothing is released)
    u32 *d_err = nullptr;     // device error word (look-back spin overflow etc.)
    u64 *d_scalars = nullptr; // small device scratch for scalar results (64 words)
    u64 *h_scalars = nullptr; // pinned mirror
    u8 *h_hdr = nullptr;      // pinned staging for a container header (1 KB)
    hipEvent_t ev[8] = {};
    hipEvent_t pev[2 * 16] = {};  // per-pass event pairs (profile mode)
    int profile = 0;
    int num_cus = 0;
    int reserved_cus = 0;  // CUs left to a tc_comm's stream: the partition levels split their work over the others
    void *hostpipe = nullptr;   // page-locked staging ring + persistent device buffers of the host entry points (textcomp.hip)
    u32 stats_ws_grown = 0;  // how often a chunked workspace grew in place
    int mtf_fastin_failed = 0;  // the one-kernel MTF + RLE of this encode could not recover a tile's list by its backward scan:
                                // the two-stage path that follows starts with the summaries (the same scan would fail again)
    int live_comms = 0;    // communicators created on this context and not yet destroyed
    int safe_tickets = 0;  // set after a look-back spin overflow: single ticket counter
    u32 ticket_fallbacks = 0;  // how often that happened (reported in tc_stats)
    int pev_used = 0;
    std::string err;
    tc_stats stats = {};
};

struct TcFail {
    int code;
};

#define TC_HIP(ctx, expr)                                                                  \
    do {                                                                                   \
        hipError_t e__ = (expr);                                                           \
        if (e__ != hipSuccess) {                                                           \
            char b__[512];                                                                 \
            snprintf(b__, sizeof b__, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,        \
                     hipGetErrorString(e__));                                              \
            (ctx)->err = b__;                                                              \
            throw TcFail{e__ == hipErrorOutOfMemory ? TC_ERR_OOM : TC_ERR_HIP};            \
        }                                                                                  \
    } while (0)

#define TC_FAIL(ctx, code_, ...)                         \
    do {                                                 \
        char b__[512];                                   \
        snprintf(b__, sizeof b__, __VA_ARGS__);          \
        (ctx)->err = b__;                                \
        throw TcFail{code_};                             \
    } while (0)

#define TC_LAUNCH_CHECK(ctx) TC_HIP(ctx, hipGetLastError())

// Bump allocator over the ctx workspace.  A top-level call first sizes its need
// with Arena(nullptr) (dry run), grows the workspace once, then carves for real.
struct Arena {
    char *base;
    size_t off = 0;
    explicit Arena(char *b) : base(b) {}
    template <class T>
    T *get(size_t count) {
        size_t bytes = (count * sizeof(T) + 255) & ~size_t(255);
        T *p = base ? reinterpret_cast<T *>(base + off) : nullptr;
        off += bytes;
        return p;
    }
};

void tc_ws_reserve(tc_ctx *ctx, size_t bytes);
// co-resident grid for persistent kernels: CUs x blocks_per_cu (TC_GRID_SCALE_PCT env scales it)
u32 tc_persistent_grid(tc_ctx *ctx, int blocks_per_cu);
// same, capped by what the occupancy query admits for this kernel/block size
template <class K>
static inline u32 tc_persistent_grid_for(tc_ctx *ctx, K kernel, int threads, int want_per_cu) {
    int occ = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, 0) != hipSuccess || occ < 1) {
        (void)hipGetLastError();
        occ = 1;
    }
    return tc_persistent_grid(ctx, occ < want_per_cu ? occ : want_per_cu);
}
// device memory as separately created physical chunks mapped into one address range (what the workspace of a long record
// is made of: textcomp.hip); *handle releases it.  Returns null when the mapping is not available.
void *tc_chunked_alloc(tc_ctx *ctx, size_t bytes, int chunk_log2, void **handle);
void tc_chunked_free(void *handle);
void tc_sync_check(tc_ctx *ctx);  // stream sync + device error word check

static inline u32 tc_cdiv(u64 a, u64 b) { return (u32)((a + b - 1) / b); }

// ------------------------------------------------------------ device helpers
#ifdef __HIPCC__

__device__ __forceinline__ u32 lane_id() { return __lane_id(); }

__device__ __forceinline__ u64 lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

// inclusive wave scan (sum) over 64 lanes
__device__ __forceinline__ u32 wave_incl_sum(u32 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += t;
    }
    return v;
}
__device__ __forceinline__ u32 wave_incl_max(u32 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u32 t = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v = v > t ? v : t;
    }
    return v;
}
__device__ __forceinline__ u64 wave_incl_max64(u64 v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        u64 t = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v = v > t ? v : t;
    }
    return v;
}
__device__ __forceinline__ u32 wave_sum(u32 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}
__device__ __forceinline__ u64 wave_max64(u64 v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        u64 t = __shfl_xor(v, d, 64);
        v = v > t ? v : t;
    }
    return v;
}

// Block-wide exclusive sum scan for NT threads (NT multiple of 64, <= 1024).
// `smem` needs NT/64 + 1 words.  Returns exclusive prefix; *total = block sum.
template <int NT>
__device__ __forceinline__ u32 block_excl_sum(u32 v, u32 *smem, u32 *total) {
    const int w = threadIdx.x >> 6;
    u32 inc = wave_incl_sum(v);
    __syncthreads();
    if (lane_id() == 63) smem[w] = inc;
    __syncthreads();
    u32 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        u32 s = smem[i];
        if (i < w) base += s;
        tot += s;
    }
    *total = tot;
    return base + inc - v;
}

// Block-wide inclusive max scan (u64).  `smem` needs NT/64 u64 words.
template <int NT>
__device__ __forceinline__ u64 block_incl_max64(u64 v, u64 *smem, u64 *total) {
    const int w = threadIdx.x >> 6;
    u64 inc = wave_incl_max64(v);
    __syncthreads();
    if (lane_id() == 63) smem[w] = inc;
    __syncthreads();
    u64 base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        u64 s = smem[i];
        if (i < w) base = base > s ? base : s;
        tot = tot > s ? tot : s;
    }
    *total = tot;
    return inc > base ? inc : base;
}

// ---- decoupled look-back over tiles (single-pass chained scan) -------------
// One 8-byte granule per tile: [63:62] flag, [61:0] value, written by ONE relaxed
// agent-scope atomic store and polled by relaxed agent-scope atomic loads (the
// "data-tagged granule" hand-off of the MI355X guide: no separate payload, so no
// release/acquire fence is needed).  Tile ids come from an atomic ticket, so a
// tile only ever waits on tiles whose blocks are already resident.
#define LB_FLAG_AGG (1ull << 62)
#define LB_FLAG_INC (2ull << 62)
#define LB_VALUE(x) ((x) & ((1ull << 62) - 1))
#define LB_SPIN_LIMIT (1u << 24)

__device__ __forceinline__ void lb_store(u64 *p, u64 v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 lb_load(const u64 *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct OpSum {
    __device__ static u64 apply(u64 a, u64 b) { return a + b; }
    __device__ static u64 identity() { return 0; }
};
struct OpMax {
    __device__ static u64 apply(u64 a, u64 b) { return a > b ? a : b; }
    __device__ static u64 identity() { return 0; }
};
// two 31-bit fields, componentwise max (both fields monotone in the tile order)
struct OpMaxPair {
    __device__ static u64 apply(u64 a, u64 b) {
        u64 ah = a >> 31, bh = b >> 31, al = a & 0x7fffffffull, bl = b & 0x7fffffffull;
        return ((ah > bh ? ah : bh) << 31) | (al > bl ? al : bl);
    }
    __device__ static u64 identity() { return 0; }
};

// Called by ALL threads of wave 0 of the block (other waves wait at the caller's
// barrier).  Publishes this tile's aggregate, walks predecessors 64 at a time and
// returns the exclusive prefix (valid in every lane of the calling wave).
template <class Op>
__device__ __forceinline__ u64 lb_exclusive(u64 *status, u32 tile, u64 aggregate, u32 *err) {
    if (tile == 0) {
        if (lane_id() == 0) lb_store(&status[0], LB_FLAG_INC | aggregate);
        return Op::identity();
    }
    if (lane_id() == 0) lb_store(&status[tile], LB_FLAG_AGG | aggregate);
    u64 excl = Op::identity();
    i64 look = (i64)tile - 1;  // lane 0 looks at `look`, lane l at look - l
    u32 spins = 0;
    while (true) {
        i64 idx = look - (i64)lane_id();
        u64 s = LB_FLAG_INC;  // virtual tiles before 0: inclusive identity
        if (idx >= 0) s = lb_load(&status[idx]);
        u64 invalid = __ballot((s >> 62) == 0);
        u64 incmask = __ballot((s >> 62) == 2);
        // lanes strictly beyond the first inclusive one do not contribute -- and are not waited for either (a window
        // holds tiles of 64 workgroups: waiting for all of them made every tile wait for the slowest of its neighbours)
        int first_inc = incmask ? __builtin_ctzll(incmask) : 64;
        if (invalid & (first_inc >= 63 ? ~0ull : ((2ull << first_inc) - 1ull))) {
            if (++spins > LB_SPIN_LIMIT) {
                if (lane_id() == 0) atomicOr(err, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        u64 v = ((int)lane_id() <= first_inc) ? LB_VALUE(s) : Op::identity();
        // reduce across the wave (order-insensitive ops only: sum / max)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) v = Op::apply(v, __shfl_xor(v, d, 64));
        excl = Op::apply(excl, v);
        if (incmask) break;
        look -= 64;
    }
    if (lane_id() == 0) lb_store(&status[tile], LB_FLAG_INC | Op::apply(excl, aggregate));
    return excl;
}

// The same for "the last non-zero value before this tile" where the values grow with the tile (positions of run ends):
// the answer is the NEAREST predecessor's value that is not zero (or an inclusive one), so the wait is for that tile
// and the ones between only -- normally tile - 1 alone -- not for all 64 tiles of a look-back window.  A tile with a
// non-zero aggregate publishes it as inclusive at once.
__device__ __forceinline__ u64 lb_exclusive_last(u64 *status, u32 tile, u64 aggregate, u32 *err) {
    if (lane_id() == 0) lb_store(&status[tile], ((tile == 0 || aggregate != 0) ? LB_FLAG_INC : LB_FLAG_AGG) | aggregate);
    if (tile == 0) return 0;
    i64 look = (i64)tile - 1;
    u32 spins = 0;
    while (true) {
        const i64 idx = look - (i64)lane_id();
        u64 s = LB_FLAG_INC;   // virtual tiles before 0: nothing before them
        if (idx >= 0) s = lb_load(&status[idx]);
        const u64 ends = __ballot((s >> 62) == 2 || ((s >> 62) == 1 && LB_VALUE(s) != 0));   // lanes that settle the answer
        const u64 invalid = __ballot((s >> 62) == 0);
        const int first_end = ends ? __builtin_ctzll(ends) : 64;
        const u64 nearer = first_end >= 64 ? ~0ull : ((1ull << first_end) - 1ull);
        if (invalid & nearer) {   // a nearer tile has not published yet
            if (++spins > LB_SPIN_LIMIT) {
                if (lane_id() == 0) atomicOr(err, 1u);
                return 0;
            }
            __builtin_amdgcn_s_sleep(1);
            continue;
        }
        if (ends) return LB_VALUE(__shfl(s, first_end, 64));
        look -= 64;   // 64 valid tiles, all without a run end
    }
}

// LDS-only barrier: orders the workgroup's LDS traffic and leaves global loads in flight -- a __syncthreads() waits
// for every memory operation the compiler knows of, so a prefetch issued before it is waited for right there.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#endif  // __HIPCC__
