// tc_sa.hpp -- suffix array + BWT last column on the device.
//
// Replaces createSuffixArray (reference BWT/Internal.hs:110-134: comparison sort
// of all n+1 suffixes incl. the empty one) and saToBWT (:98-106).  Algorithm:
//   round 0  k-mer sort: every suffix gets a 64-bit key = its first h0 symbols in
//            a dense base-(sigma+1) code ('$' = 0 < every symbol, so a proper
//            prefix sorts first, Q1) plus, in the low byte, the byte that precedes
//            it in the text; one LSD radix sort over the key's used bits.  The
//            low byte makes the last column fall out of the sorted keys with no
//            gather: L[j] = low byte of key[j].
//   round r  prefix doubling on the still-tied suffixes only: sort the active set
//            by (group, rank[i + h]), split groups, h doubles.
// All distinct suffixes have distinct (infinite) keys, so the result is the unique
// order the reference's sort produces, whatever its algorithm.
#pragma once
#include "tc_radix.hpp"

#define SA_NT 256
#define SA_ITEMS 16
#define SA_TILE (SA_NT * SA_ITEMS)
#define KB_HALO 72  // >= P*s + s for every configuration (<= 7*8 + 8)

struct SaConfig {
    u32 sigma_text;  // distinct byte values present
    u32 B;           // sigma_text + 1 ('$' = code 0)
    u32 w;           // bits per field
    u32 s;           // symbols per field
    u32 P;           // fields sorted in round 0
    u32 h0;          // P * s
    u16 lut[256];    // byte -> code (1..sigma_text), 0 if absent
};

#ifdef __HIPCC__

// ---- byte histogram -----------------------------------------------------------
__global__ __launch_bounds__(256) void hist256_kernel(const u8 *__restrict__ text, u64 n,
                                                      u32 *__restrict__ counts) {
    __shared__ u32 s_h[4][256];
    for (int i = threadIdx.x; i < 1024; i += 256) (&s_h[0][0])[i] = 0;
    __syncthreads();
    u32 *h = s_h[threadIdx.x >> 6];
    const u64 nvec = ((uintptr_t)text & 15) ? 0 : n / 16;
    const uint4 *tv = reinterpret_cast<const uint4 *>(text);
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (u64)gridDim.x * 256) {
        uint4 v = tv[i];
        u32 x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            atomicAdd(&h[x[q] & 255], 1u);
            atomicAdd(&h[(x[q] >> 8) & 255], 1u);
            atomicAdd(&h[(x[q] >> 16) & 255], 1u);
            atomicAdd(&h[x[q] >> 24], 1u);
        }
    }
    for (u64 i = nvec * 16 + (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256)
        atomicAdd(&h[text[i]], 1u);
    __syncthreads();
    u32 c = s_h[0][threadIdx.x] + s_h[1][threadIdx.x] + s_h[2][threadIdx.x] + s_h[3][threadIdx.x];
    if (c) atomicAdd(&counts[threadIdx.x], c);
}

struct KeyBuildParams {
    u32 B, w, s, P;
    u16 lut[256];
    RadixPlanDev plan;
};

// ---- round-0 keys ---------------------------------------------------------------
// key(i) = fields G(i), G(i+s), ..  (G(p) = s symbols from p, base B) from bit 63
// down, w bits each; low byte = text[i-1] (0 for i == 0).  Also accumulates the
// digit histograms of every sort pass (saves one read of the keys).
__global__ __launch_bounds__(SA_NT) void keybuild_kernel(const u8 *__restrict__ text, u32 n,
                                                         KeyBuildParams kp,
                                                         u64 *__restrict__ keys,
                                                         u32 *__restrict__ hist) {
    __shared__ u16 s_c[SA_TILE + KB_HALO + 8];  // codes, slot 0 = position base-1
    __shared__ u16 s_g[SA_TILE + KB_HALO + 8];
    __shared__ u8 s_raw[SA_TILE + 8];            // raw bytes, slot 0 = position base-1
    __shared__ u32 s_h[RDX_MAX_PASSES * RDX_BINS];
    __shared__ u16 s_lut[256];
    const int tid = threadIdx.x;
    const u32 N = n + 1;
    const u64 base = (u64)blockIdx.x * SA_TILE;
    for (int i = tid; i < kp.plan.npass * RDX_BINS; i += SA_NT) s_h[i] = 0;
    s_lut[tid] = kp.lut[tid];
    __syncthreads();
    const u32 span = SA_TILE + kp.P * kp.s + kp.s;  // symbols needed from `base`
    for (u32 p = tid; p < span + 1; p += SA_NT) {
        i64 pos = (i64)base + (i64)p - 1;
        u16 c = 0;
        u8 raw = 0;
        if (pos >= 0 && pos < (i64)n) {
            raw = text[pos];
            c = s_lut[raw];
        }
        s_c[p] = c;
        if (p <= SA_TILE) s_raw[p] = raw;
    }
    __syncthreads();
    for (u32 p = tid; p < SA_TILE + kp.P * kp.s; p += SA_NT) {
        u32 g = 0;
        for (u32 j = 0; j < kp.s; j++) g = g * kp.B + s_c[p + 1 + j];
        s_g[p] = (u16)g;
    }
    __syncthreads();
#pragma unroll 4
    for (int k = 0; k < SA_ITEMS; k++) {
        u32 p = tid + k * SA_NT;
        u64 i = base + p;
        if (i < N) {
            u64 key = 0;
            int sh = 64;
            for (u32 f = 0; f < kp.P; f++) {
                sh -= kp.w;
                key |= (u64)s_g[p + f * kp.s] << sh;
            }
            key |= (u64)s_raw[p];
            keys[i] = key;
            for (int q = 0; q < kp.plan.npass; q++)
                atomicAdd(&s_h[q * RDX_BINS + (u32)((key >> kp.plan.shift[q]) & kp.plan.mask[q])],
                          1u);
        }
    }
    __syncthreads();
    for (int i = tid; i < kp.plan.npass * RDX_BINS; i += SA_NT) {
        u32 c = s_h[i];
        if (c) atomicAdd(&hist[i], c);
    }
}

// ---- group detection / re-ranking ---------------------------------------------
// INIT (round 0): element k sits at SA position k; tie <=> equal key bits above the
//   payload byte; writes L[k] from the key.
// REFINE (round r): element k of the sorted active set goes to SA position
//   slot[k]; tie <=> equal (group, rank[i+h]) key; writes SA, L (gathered).
// Both: rank[idx] = position of the group's first member; members of groups of
// size >= 2 are compacted into the next active set.
struct GroupArgs {
    const u64 *keys;  // sorted keys
    const u32 *idx;   // suffix start per element (INIT: the SA itself)
    const u32 *slot;  // REFINE: SA position of the k-th active element (increasing)
    u32 count;
    const u8 *text;
    u32 *sa;
    u32 *isa;
    u8 *L;
    u32 *out_slot, *out_idx, *out_grp;  // next active set
    u64 *status_max, *status_sum;       // look-back granules, [tiles] each
    u32 *ticket;
    u64 *scalars;  // [0] primary, [1] active count
    u32 *err;
};

template <bool INIT>
__global__ __launch_bounds__(SA_NT) void group_kernel(GroupArgs a) {
    constexpr int NW = SA_NT / 64;
    __shared__ u64 s_wmax[NW];
    __shared__ u32 s_wsum[NW];
    __shared__ u64 s_pref[2];
    __shared__ u32 s_tile;
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
    if (tid == 0) s_tile = atomicAdd(a.ticket, 1u);
    __syncthreads();
    const u32 tile = s_tile;
    const u64 base = (u64)tile * SA_TILE + (u64)w * 64 * SA_ITEMS;
    const u64 KMASK = INIT ? ~0xffull : ~0ull;

    u64 key[SA_ITEMS];
    u32 idx[SA_ITEMS], pos[SA_ITEMS];
    u8 lowb[SA_ITEMS];
    int hb[SA_ITEMS];
#pragma unroll
    for (int k = 0; k < SA_ITEMS; k++) {
        u64 j = base + k * 64 + l;
        bool in = j < a.count;
        u64 raw = in ? a.keys[j] : 0;
        lowb[k] = (u8)(raw & 0xff);
        key[k] = raw & KMASK;
        idx[k] = in ? a.idx[j] : 0;
        pos[k] = INIT ? (u32)j : (in ? a.slot[j] : 0);
    }
    // head flags: first element, or key differs from its predecessor
    {
        u64 prevk = 0;
        bool has_prev = false;
        if (base > 0 && base < a.count) {
            prevk = a.keys[base - 1] & KMASK;  // same address in every lane: broadcast
            has_prev = true;
        }
#pragma unroll
        for (int k = 0; k < SA_ITEMS; k++) {
            u64 up = __shfl_up(key[k], 1, 64);
            u64 pk = (l == 0) ? prevk : up;
            bool hp = (l == 0) ? has_prev : true;
            hb[k] = (!hp || pk != key[k]) ? 1 : 0;
            prevk = __shfl(key[k], 63, 64);
            has_prev = true;
        }
    }
    // next-head flags (a virtual head sits just past the end)
    int nhb[SA_ITEMS];
    {
        int tail_next = 1;  // head flag of element base + 64*ITEMS
        u64 jn = base + (u64)64 * SA_ITEMS;
        if (jn < a.count) tail_next = ((a.keys[jn] & KMASK) != __shfl(key[SA_ITEMS - 1], 63, 64)) ? 1 : 0;
#pragma unroll
        for (int k = 0; k < SA_ITEMS; k++) {
            int dn = __shfl_down(hb[k], 1, 64);
            int nx0 = tail_next;
            if (k + 1 < SA_ITEMS) nx0 = __shfl(hb[(k + 1) % SA_ITEMS], 0, 64);
            int nh = (l < 63) ? dn : nx0;
            u64 j = base + k * 64 + l;
            if (j + 1 >= a.count) nh = 1;
            nhb[k] = nh;
        }
    }
    // local scans in (wave, item, lane) order
    u64 gmax[SA_ITEMS];
    u32 aexc[SA_ITEMS];
    bool act[SA_ITEMS];
    u64 cmax = 0;
    u32 csum = 0;
#pragma unroll
    for (int k = 0; k < SA_ITEMS; k++) {
        u64 j = base + k * 64 + l;
        bool in = j < a.count;
        // group start encoded +1 so that 0 is the identity
        u64 v = (in && hb[k]) ? (u64)pos[k] + 1 : 0;
        u64 inc = wave_incl_max64(v);
        inc = inc > cmax ? inc : cmax;
        gmax[k] = inc;
        cmax = __shfl(inc, 63, 64);
        act[k] = in && !(hb[k] && nhb[k]);
        u32 av = act[k] ? 1u : 0u;
        u32 ai = wave_incl_sum(av);
        aexc[k] = csum + ai - av;
        csum += __shfl(ai, 63, 64);
    }
    if (l == 63) {
        s_wmax[w] = cmax;
        s_wsum[w] = csum;
    }
    __syncthreads();
    u64 wpmax = 0, bmax = 0;
    u32 wpsum = 0, bsum = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        if (i < w) {
            wpmax = wpmax > s_wmax[i] ? wpmax : s_wmax[i];
            wpsum += s_wsum[i];
        }
        bmax = bmax > s_wmax[i] ? bmax : s_wmax[i];
        bsum += s_wsum[i];
    }
    if (w == 0) {
        u64 e = lb_exclusive<OpMax>(a.status_max, tile, bmax, a.err);
        if (l == 0) s_pref[0] = e;
    } else if (w == 1) {
        u64 e = lb_exclusive<OpSum>(a.status_sum, tile, bsum, a.err);
        if (l == 0) {
            s_pref[1] = e;
            if ((u64)(tile + 1) * SA_TILE >= a.count) a.scalars[1] = e + bsum;  // last tile
        }
    }
    __syncthreads();
    const u64 tpmax = s_pref[0] > wpmax ? s_pref[0] : wpmax;
    const u32 tpsum = (u32)s_pref[1] + wpsum;
#pragma unroll
    for (int k = 0; k < SA_ITEMS; k++) {
        u64 j = base + k * 64 + l;
        if (j >= a.count) continue;
        u64 g1 = gmax[k] > tpmax ? gmax[k] : tpmax;
        u32 g = (u32)(g1 - 1);
        u32 i = idx[k];
        a.isa[i] = g;
        if (INIT) {
            a.L[j] = lowb[k];
        } else {
            a.sa[pos[k]] = i;
            a.L[pos[k]] = i ? a.text[i - 1] : (u8)0;
        }
        if (i == 0) a.scalars[0] = pos[k];
        if (act[k]) {
            u32 o = tpsum + aexc[k];
            a.out_slot[o] = pos[k];
            a.out_idx[o] = i;
            a.out_grp[o] = g;
        }
    }
}

// key2[k] = group << 32 | rank[idx + h]
__global__ __launch_bounds__(256) void key2_kernel(const u32 *__restrict__ idx,
                                                   const u32 *__restrict__ grp,
                                                   const u32 *__restrict__ isa, u32 m, u32 h,
                                                   u32 N, u64 *__restrict__ keys) {
    u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    u64 p = (u64)idx[k] + h;
    u32 r = p < N ? isa[p] : 0u;  // p <= n always holds for a tied suffix
    keys[k] = ((u64)grp[k] << 32) | r;
}

__global__ __launch_bounds__(256) void copy_u32_kernel(const u32 *__restrict__ a,
                                                       u32 *__restrict__ b, u32 m) {
    u32 k = blockIdx.x * 256 + threadIdx.x;
    if (k < m) b[k] = a[k];
}

#endif  // __HIPCC__
