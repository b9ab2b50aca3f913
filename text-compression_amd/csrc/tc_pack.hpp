// tc_pack.hpp -- wire format of a tc_block's runs (payload of the multi-GPU gather).
// The reference has no wire format; this one is ours and is an exact inverse pair.
//
// sigma <= 6 ("nibble stream", what an ACGTN record produces: 5 letters + sentinel):
//   code 0..11   start of a run: value = code % 6, count = 1 + code / 6
//   code 12, 13  follows a start with count 2: count becomes 3, 4
//   code 14      follows a start: count is the next word of the escape list (run order)
//   code 15      padding (every 16384-run tile of the packer ends on a 16-byte boundary)
// so a run costs 4 bits (count 1, 2) or 8 bits; iid ACGTN: 1.04 nibbles per run.
// 6 < sigma <= 16: one byte per run (value | min(count,15) << 4); sigma > 16: two bytes per run;
// both with an escape list of (run index, count) pairs.
#pragma once
#include "tc_common.hpp"

#define PK_NT 256
#define PK_RPT 16                       // runs per thread per sub-tile
#define PK_SUBRUNS (PK_NT * PK_RPT)     // 4096
#define PK_SUB 4
#define PK_TILE (PK_SUBRUNS * PK_SUB)   // 16384 runs per tile
#define PK_WORDS (PK_TILE * 2 / 16)     // u64 words of nibbles, worst case (2 per run)
#define PK_NIB_SIGMA 6
static_assert(PK_TILE * 2 < 65536, "a tile's nibble count shares a 32-bit word with its escape count (16 bits each)");

static inline int pack_format(u32 sigma) { return sigma <= PK_NIB_SIGMA ? 0 : sigma <= 16 ? 1 : 2; }

struct PackNibArgs {
    const u32 *cnt;
    const u16 *val;
    u64 nruns;
    u8 *out;        // 16-byte aligned
    u64 cap_units;  // 16-byte units available in `out`
    u32 *esc;
    u64 esc_cap;
    u64 *status;
    u32 *ticket;
    u32 *err;
    u32 ntiles;
};

// append the 1 or 2 nibbles of one run to the 128-bit string (L, H) of `len` nibbles
__device__ __forceinline__ void pk_append(u64 &L, u64 &H, u32 &len, u32 &nesc, u32 c, u32 v) {
    const u32 c1 = c - 1u;  // 0..3 for counts 1..4
    u32 pair, nl;
    if (c1 < 2u) {
        pair = v + 6u * c1;
        nl = 1;
    } else if (c1 < 4u) {
        pair = (v + 6u) | ((10u + c1) << 4);
        nl = 2;
    } else {
        pair = v | (14u << 4);
        nl = 2;
        nesc++;
    }
    const u32 sh = (len & 15u) * 4u;
    if (len < 16u) {
        L |= (u64)pair << sh;
        if (len == 15u && nl == 2u) H |= (u64)(pair >> 4);
    } else {
        H |= (u64)pair << sh;
    }
    len += nl;
}

__global__ __launch_bounds__(PK_NT) void pack_nib_kernel(PackNibArgs a) {
    __shared__ u64 nib[PK_WORDS + 2];
    __shared__ u32 wsum[PK_SUB][PK_NT / 64];
    __shared__ u32 s_tile;
    __shared__ u64 s_excl;
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    while (true) {
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= a.ntiles) break;
        const u64 base = (u64)tile * PK_TILE;
        for (int i = tid; i < PK_WORDS + 2; i += PK_NT) nib[i] = 0;

        u64 L[PK_SUB], H[PK_SUB];
        u32 x[PK_SUB];  // nibbles | escapes << 16
#pragma unroll
        for (int s = 0; s < PK_SUB; s++) {
            const u64 r0 = base + (u64)s * PK_SUBRUNS + (u64)tid * PK_RPT;
            u32 c[PK_RPT], v[PK_RPT];
            if (r0 + PK_RPT <= a.nruns) {
                const uint4 *pc = reinterpret_cast<const uint4 *>(a.cnt + r0);
                const uint4 *pv = reinterpret_cast<const uint4 *>(a.val + r0);
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    uint4 t = pc[q];
                    c[4 * q] = t.x; c[4 * q + 1] = t.y; c[4 * q + 2] = t.z; c[4 * q + 3] = t.w;
                }
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    uint4 t = pv[q];
                    v[8 * q] = t.x & 0xffffu; v[8 * q + 1] = t.x >> 16;
                    v[8 * q + 2] = t.y & 0xffffu; v[8 * q + 3] = t.y >> 16;
                    v[8 * q + 4] = t.z & 0xffffu; v[8 * q + 5] = t.z >> 16;
                    v[8 * q + 6] = t.w & 0xffffu; v[8 * q + 7] = t.w >> 16;
                }
            } else {
#pragma unroll
                for (int j = 0; j < PK_RPT; j++) {
                    const bool ok = r0 + j < a.nruns;
                    c[j] = ok ? a.cnt[r0 + j] : 0u;
                    v[j] = ok ? a.val[r0 + j] : 0xffffu;  // 0xffff: no run
                }
            }
            u64 l = 0, h = 0;
            u32 len = 0, ne = 0;
#pragma unroll
            for (int j = 0; j < PK_RPT; j++)
                if (v[j] != 0xffffu) pk_append(l, h, len, ne, c[j], v[j] < 5u ? v[j] : 5u);
            L[s] = l; H[s] = h;
            x[s] = len | (ne << 16);
        }
        // offsets: sub-tile major, then thread order
        u32 inc[PK_SUB];
#pragma unroll
        for (int s = 0; s < PK_SUB; s++) inc[s] = wave_incl_sum(x[s]);
        if (lane == 63) {
#pragma unroll
            for (int s = 0; s < PK_SUB; s++) wsum[s][w] = inc[s];
        }
        __syncthreads();  // also: nib[] zeroed
        u32 run = 0, excl[PK_SUB];
#pragma unroll
        for (int s = 0; s < PK_SUB; s++) {
            u32 b = run;
#pragma unroll
            for (int i = 0; i < PK_NT / 64; i++) {
                const u32 t = wsum[s][i];
                if (i < w) b += t;
                run += t;
            }
            excl[s] = b + inc[s] - x[s];
        }
        const u32 tile_len = run & 0xffffu, tile_esc = run >> 16;
        const u32 units = (tile_len + 31u) >> 5;
#pragma unroll
        for (int s = 0; s < PK_SUB; s++) {
            const u32 off = excl[s] & 0xffffu;
            const u32 wd = off >> 4, sh = (off & 15u) * 4u;
            const u64 l = L[s], h = H[s];
            const u64 w0 = l << sh;
            const u64 w1 = (sh ? l >> (64 - sh) : 0ull) | (h << sh);
            const u64 w2 = sh ? h >> (64 - sh) : 0ull;
            if (w0) atomicOr((unsigned long long *)&nib[wd], (unsigned long long)w0);
            if (w1) atomicOr((unsigned long long *)&nib[wd + 1], (unsigned long long)w1);
            if (w2) atomicOr((unsigned long long *)&nib[wd + 2], (unsigned long long)w2);
        }
        if (w == 0) {
            const u64 e = lb_exclusive<OpSum>(a.status, tile, ((u64)units << 32) | tile_esc, a.err);
            if (lane == 0) s_excl = e;
        }
        __syncthreads();
        const u64 excl_units = s_excl >> 32, excl_esc = s_excl & 0xffffffffull;
        for (u32 u = tid; u < units; u += PK_NT) {
            u64 w0 = nib[2 * u], w1 = nib[2 * u + 1];
            const u32 n0 = 32u * u;  // first nibble of this unit
            if (tile_len < n0 + 16u) w0 |= ~0ull << ((tile_len - n0) * 4u);
            if (tile_len < n0 + 32u) w1 = tile_len <= n0 + 16u ? ~0ull : (w1 | (~0ull << ((tile_len - n0 - 16u) * 4u)));
            const u64 g = excl_units + u;
            if (g < a.cap_units) reinterpret_cast<ulonglong2 *>(a.out)[g] = make_ulonglong2(w0, w1);
        }
        // escapes are rare: their owners re-read the counts and append them in run order
#pragma unroll
        for (int s = 0; s < PK_SUB; s++) {
            if (x[s] >> 16) {
                const u64 r0 = base + (u64)s * PK_SUBRUNS + (u64)tid * PK_RPT;
                u64 e = excl_esc + (excl[s] >> 16);
                for (int j = 0; j < PK_RPT; j++) {
                    if (r0 + j >= a.nruns) break;
                    const u32 c = a.cnt[r0 + j];
                    if (c - 1u >= 4u) {
                        if (e < a.esc_cap) a.esc[e] = c;
                        e++;
                    }
                }
            }
        }
    }
}

#define UP_NT 256
#define UP_NPT 32                   // nibbles per thread = one 16-byte unit
#define UP_TILE_UNITS UP_NT         // 4 KB of body per tile

struct UnpackNibArgs {
    const u8 *body;
    u64 units;
    const u32 *esc;
    u64 nesc;
    u64 nruns;
    u32 *cnt;
    u16 *val;
    u64 *status;
    u32 *ticket;
    u32 *err;
    u32 ntiles;
};

__global__ __launch_bounds__(UP_NT) void unpack_nib_kernel(UnpackNibArgs a) {
    __shared__ u32 s_first[UP_NT + 1];
    __shared__ u32 sm[UP_NT / 64 + 1];
    __shared__ u32 s_tile;
    __shared__ u64 s_excl;
    __shared__ u32 s_cnt[UP_NT * UP_NPT];   // a tile holds at most one run per nibble
    __shared__ u8 s_val[UP_NT * UP_NPT];
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    while (true) {
        if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
        __syncthreads();
        const u32 tile = s_tile;
        if (tile >= a.ntiles) break;
        const u64 unit = (u64)tile * UP_TILE_UNITS + tid;
        u64 w0 = ~0ull, w1 = ~0ull;
        if (unit < a.units) {
            const ulonglong2 t = reinterpret_cast<const ulonglong2 *>(a.body)[unit];
            w0 = t.x; w1 = t.y;
        }
        s_first[tid] = (u32)(w0 & 15u);
        if (tid == UP_NT - 1) s_first[UP_NT] = unit + 1 < a.units ? (u32)(a.body[(unit + 1) * 16] & 15u) : 15u;
        const u64 m = 0x1111111111111111ull;
        const u64 ns0 = (w0 >> 3) & (w0 >> 2) & m, ns1 = (w1 >> 3) & (w1 >> 2) & m;   // codes >= 12
        const u32 starts = (u32)__popcll(~ns0 & m) + (u32)__popcll(~ns1 & m);
        const u32 escs = (u32)__popcll(ns0 & (w0 >> 1) & ~w0) + (u32)__popcll(ns1 & (w1 >> 1) & ~w1);
        u32 total;
        const u32 x = starts | (escs << 16);
        const u32 ex = block_excl_sum<UP_NT>(x, sm, &total);
        if (w == 0) {
            const u64 e = lb_exclusive<OpSum>(a.status, tile, ((u64)(total & 0xffffu) << 31) | (total >> 16), a.err);
            if (lane == 0) s_excl = e;
        }
        __syncthreads();
        const u64 kbase = s_excl >> 31;          // first run of this tile
        u32 kl = ex & 0xffffu;                   // tile-local run index
        u64 e = (s_excl & 0x7fffffffull) + (ex >> 16);
        if ((w0 & 15u) == 14u) e++;  // belongs to the run that starts in the previous unit
        const u32 nextfirst = s_first[tid + 1];
        // runs are staged tile-locally (count | value << 28 would not hold escaped counts: two arrays)
#pragma unroll
        for (int p = 0; p < UP_NPT; p++) {
            const u32 code = (u32)((p < 16 ? w0 >> (4 * p) : w1 >> (4 * (p - 16))) & 15u);
            if (code < 12u) {
                const u32 nx = p == 31 ? nextfirst : (u32)((p + 1 < 16 ? w0 >> (4 * (p + 1)) : w1 >> (4 * (p - 15))) & 15u);
                u32 c = code >= 6u ? 2u : 1u;
                const u32 v = code >= 6u ? code - 6u : code;
                if (nx == 12u) c = 3u;
                else if (nx == 13u) c = 4u;
                else if (nx == 14u) {
                    c = e < a.nesc ? a.esc[e] : 0u;
                    e++;
                }
                s_cnt[kl] = c;
                s_val[kl] = (u8)v;
                kl++;
            }
        }
        __syncthreads();
        // coalesced write-out of the tile's runs
        const u32 truns = total & 0xffffu;
        for (u32 i = tid; i < truns; i += UP_NT) {
            const u64 k = kbase + i;
            if (k < a.nruns) {
                a.cnt[k] = s_cnt[i];
                a.val[k] = (u16)s_val[i];
            }
        }
        __syncthreads();  // s_first / s_excl reuse
    }
}

// ---- byte formats (6 < sigma) ------------------------------------------------------
// bytes per run: 1 (sigma <= 16) or 2 (value byte, count byte with escape; the ninth value bit
// of sigma = 257 rides in the count byte's top bit -> counts escape at 127)
// byte formats.  A block owns PR_TILE consecutive runs (a thread 8 of them); escapes are listed in
// run order -- pass 1 counts them per block, a scan gives every block its slot, pass 2 writes -- so
// that the packed bytes are a function of the runs alone.
#define PR_TILE 2048
__device__ __forceinline__ bool pr_escapes(int bpr, u32 c) { return bpr == 1 ? c >= 15 : c >= 127; }
__global__ __launch_bounds__(256) void pack_runs_count_kernel(const u32 *__restrict__ cnt, u64 nruns, int bpr,
                                                              u64 *__restrict__ tcnt) {
    __shared__ u32 s[4];
    const u64 base = (u64)blockIdx.x * PR_TILE;
    u32 e = 0;
    for (int k = 0; k < 8; k++) {
        const u64 i = base + (u64)k * 256 + threadIdx.x;
        if (i < nruns && pr_escapes(bpr, cnt[i])) e++;
    }
    for (int d = 32; d >= 1; d >>= 1) e += __shfl_xor(e, d, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0) tcnt[blockIdx.x] = (u64)s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void pack_runs_kernel(const u32 *__restrict__ cnt,
                                                        const u16 *__restrict__ val, u64 nruns,
                                                        int bpr, u8 *__restrict__ out,
                                                        u32 *__restrict__ esc, const u64 *__restrict__ tcnt,
                                                        u64 esc_cap) {
    __shared__ u32 s[4];
    const u64 base = (u64)blockIdx.x * PR_TILE + (u64)threadIdx.x * 8;
    u32 c[8], ne = 0;
    for (int k = 0; k < 8; k++) {
        const u64 i = base + k;
        c[k] = 0;
        if (i >= nruns) continue;
        c[k] = cnt[i];
        const u32 v = val[i];
        const bool e = pr_escapes(bpr, c[k]);
        ne += e;
        if (bpr == 1) {
            out[i] = (u8)((v & 15u) | ((e ? 15u : c[k]) << 4));
        } else {
            out[2 * i] = (u8)v;
            out[2 * i + 1] = (u8)((e ? 127u : c[k]) | ((v >> 8) << 7));
        }
    }
    u32 inc = ne;
    for (int d = 1; d < 64; d <<= 1) {
        const u32 t = __shfl_up(inc, d, 64);
        if ((int)(threadIdx.x & 63) >= d) inc += t;
    }
    if ((threadIdx.x & 63) == 63) s[threadIdx.x >> 6] = inc;
    __syncthreads();
    u64 slot = tcnt[blockIdx.x] + (inc - ne);
    for (int w = 0; w < (int)(threadIdx.x >> 6); w++) slot += s[w];
    if (ne) {
        for (int k = 0; k < 8; k++) {
            if (base + k < nruns && pr_escapes(bpr, c[k])) {
                if (slot < esc_cap) {
                    esc[2 * slot] = (u32)(base + k);
                    esc[2 * slot + 1] = c[k];
                }
                slot++;
            }
        }
    }
}
__global__ __launch_bounds__(256) void unpack_runs_kernel(const u8 *__restrict__ in, u64 nruns, int bpr,
                                                          u32 *__restrict__ cnt, u16 *__restrict__ val) {
    for (u64 k = (u64)blockIdx.x * 256 + threadIdx.x; k < nruns; k += (u64)gridDim.x * 256) {
        if (bpr == 1) {
            u8 b = in[k];
            val[k] = (u16)(b & 15);
            cnt[k] = (u32)(b >> 4);
        } else {
            u8 lo = in[2 * k], hi = in[2 * k + 1];
            val[k] = (u16)(lo | ((hi >> 7) << 8));
            cnt[k] = (u32)(hi & 127);
        }
    }
}
__global__ __launch_bounds__(256) void unpack_esc_kernel(const u32 *__restrict__ esc, u64 nesc,
                                                         u64 nruns, u32 *__restrict__ cnt) {
    u64 i = (u64)blockIdx.x * 256 + threadIdx.x;
    if (i < nesc && esc[2 * i] < nruns) cnt[esc[2 * i]] = esc[2 * i + 1];
}
