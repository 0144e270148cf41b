// Paged decode attention for gfx950: one query token per request, K/V gathered from the
// slot pool through req_to_token.  HBM-bandwidth bound (G flop per byte, G = Hq/Hkv <= 8),
// so the kernel is built around the load stream, not the arithmetic:
//
//   * a work group (8 waves) owns one (request, split) pair and ALL local kv heads, so the
//     waves of a group together read whole token rows ([Hkv, 128] bf16 = up to 2 KiB
//     contiguous) and whole pages;
//   * a wave owns one kv head and walks its share of the split in tiles of 16 tokens.  One
//     global_load_dwordx4 per lane covers 4 tokens: the 16 lanes of a DPP row read one
//     256-byte K (or V) row, the 4 rows of the wave read 4 tokens;
//   * K and V tiles live in VGPRs (no LDS round trip: nothing is shared between waves).
//     The next tile's K loads are issued as soon as the scores of the current tile are
//     formed, the next V loads as soon as P.V is done, and the slot indices are fetched
//     two tiles ahead, so every wave keeps 4-8 KiB in flight;
//   * the G query heads of the kv head stay in registers as packed bf16 and feed
//     v_dot2c_f32_bf16; partial dot products are summed across the 16 lanes of a row with
//     DPP adds (no LDS, no ds_bpermute);
//   * each DPP row keeps its own online-softmax state (running max, sum, 8 output
//     columns per lane and head) — rows are merged once, after the loop, and waves that
//     share a kv head are merged through LDS;
//   * split-K over the sequence: the grid is always bs * max_splits groups and a request
//     uses ceil(seq_len / chunk) of them, so the launch shape is independent of the
//     sequence lengths; partial results go to a caller-provided workspace and a second
//     small kernel merges them (requests that fit one split are written directly).
#include "msgl_common.h"
#include "decode_params.h"
#include "plan_sort.h"

#include <stdlib.h>

#include <mutex>
#include <type_traits>

namespace msgl {

__device__ __forceinline__ uint4 ld16(const uint16_t* p) { return *reinterpret_cast<const uint4*>(p); }
// K/V row load.  NT = non-temporal hint: rows that no other request re-reads in this launch should not
// displace the few that are shared (prefix pages of the radix cache) from L2 / the Infinity Cache.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
template <bool NT>
__device__ __forceinline__ uint4 ld16_kv(const uint16_t* p) {
    if (NT) {
        const u32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x4_t*>(p));
        return make_uint4(v.x, v.y, v.z, v.w);
    }
    return *reinterpret_cast<const uint4*>(p);
}

// Register ring depth: tiles of K and of V in flight per wave.
template <int G>
struct RingDepth {
    static constexpr int value = (G >= 8) ? 1 : 2;  // 64 accumulators at G = 8 leave room for one tile
};

template <int G, int HPB, bool PAGED16>
__global__ __launch_bounds__(kWaves * 64) void paged_decode_kernel(DecodeParams p) {
    constexpr int WPH = kWaves / HPB;  // waves sharing one kv head
    constexpr int DEPTH = RingDepth<G>::value;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* lds_o = reinterpret_cast<float*>(smem);                       // [kWaves][G][128]
    float* lds_ml = lds_o + kWaves * G * kD;                             // [kWaves][G][2]

    // split-major block order: consecutive blocks (dealt round-robin to the 8 XCDs) are different
    // requests of the same split, so the active blocks cover every XCD whatever the split count is
    int item = blockIdx.x;
    if (p.plan) {  // ragged batch: full splits first, the partial tails by falling length after them
        item = p.plan[blockIdx.x];
        if (item < 0) return;
    }
    const int split = item / p.bs;
    const int b = item - split * p.bs;
    const int chunk = split_chunk(p);
    // (the request's table row is asked for together with its lengths, ahead of the early exit below: one round trip
    // to memory less in front of the first page-table read)
    const int req_row = p.req_idx[b];
    const SplitInfo si = split_info(p, b, chunk);
    asm volatile("" ::"s"(req_row));
    const int S = si.S, nsplit = si.nsplit;
    if (split >= nsplit) return;
    const int t0 = si.start + split * chunk;
    const int t1 = (split == p.max_splits - 1) ? S : min(S, t0 + chunk);  // last split takes the tail
    const int ntiles = (t1 - t0 + kTile - 1) / kTile;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // Lane roles.  K rows are read by OCTETS: 8 lanes x 16 B cover one 128-byte half row, a load
    // instruction covers 8 tokens, and a score needs only a 3-step reduction.  V rows are read by DPP ROWS:
    // 16 lanes x 16 B cover one 256-byte row, so each lane accumulates 8 output columns per query head.
    // Row j of the wave owns tile tokens 8t + 2j + b (t, b in {0, 1}) in both layouts.
    const int j = lane >> 4;        // DPP row
    const int c = lane & 15;        // V: 16-byte column chunk of the 256-byte head row
    const int o = lane >> 3;        // K: octet = token 8t + o of the tile
    const int s8 = lane & 7;        // K: 16-byte chunk inside a 128-byte half row
    const int h = wave / WPH;       // local kv head
    const int sub = wave - h * WPH;
    // this wave's tiles are sub, sub + WPH, ...; "n" below counts them
    const int my_tiles = (ntiles > sub) ? (ntiles - sub + WPH - 1) / WPH : 0;

    const int32_t* r2t_row = p.r2t + (int64_t)req_row * p.r2t_stride;
    const uint16_t* kbase = p.kc + h * kD + s8 * 8;
    const uint16_t* vbase = p.vc + h * kD + c * 8;

    uint4 qv[G][2];  // query columns [64k + 8 s8, +8) of head g, packed bf16
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k = 0; k < 2; ++k) qv[g][k] = ld16(p.q + (int64_t)b * p.q_stride + (h * G + g) * kD + k * 64 + s8 * 8);

    float m[G], l[G], acc[G][8];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        m[g] = kNegBig;
        l[g] = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[g][e] = 0.f;
    }

    // Table entry (entries) of own tile n, always from a valid position so that the load is
    // unconditional: PAGED16 -> the slot of the tile's first position (wave uniform, a scalar load);
    // otherwise lane c holds the slot of the tile's c-th token.
    // A tile that starts past the end of the split is clamped to the split's last tile (PAGED16: to that tile's
    // first position, t0 being a multiple of 16, so that the 16 rows read from the entry stay inside its page).
    auto load_index = [&](int n) -> int {
        int tok = t0 + (sub + n * WPH) * kTile + (PAGED16 ? 0 : c);
        tok = min(tok, PAGED16 ? ((t1 - 1) & ~(kTile - 1)) : t1 - 1);
        return r2t_row[tok];
    };
    // Row addresses of the tile whose table entry is idx: K tokens 8t + o, V tokens 8t + 2j + b (instruction
    // i = 2t + b).  PAGED16: the 16 slots are consecutive, so an address is a wave-uniform base (scalar
    // arithmetic on idx) plus a byte offset that never changes for the lane, and the tile costs next to no
    // vector address arithmetic -- used where the vector pipe is the limit (G = 8; with fewer heads the kernel
    // waits on memory and measured the same either way).  Otherwise every lane multiplies its own slot.
    constexpr bool UNIFORM_BASE = PAGED16 && G >= 8;
    uint32_t klane[2], vlane[4];
#pragma unroll
    for (int t = 0; t < 2; ++t) klane[t] = (uint32_t)(((8 * t + o) * p.cache_stride + s8 * 8) * 2);
#pragma unroll
    for (int i = 0; i < 4; ++i) vlane[i] = (uint32_t)(((8 * (i >> 1) + 2 * j + (i & 1)) * p.cache_stride + c * 8) * 2);
    auto tile_rows = [&](int idx, const uint16_t* (&kp)[2], const uint16_t* (&vp)[4]) {
        if constexpr (UNIFORM_BASE) {
            const int64_t first = (int64_t)__builtin_amdgcn_readfirstlane(idx) * p.cache_stride + h * kD;
            const char* kt = reinterpret_cast<const char*>(p.kc + first);
            const char* vt = reinterpret_cast<const char*>(p.vc + first);
#pragma unroll
            for (int t = 0; t < 2; ++t) kp[t] = reinterpret_cast<const uint16_t*>(kt + klane[t]);
#pragma unroll
            for (int i = 0; i < 4; ++i) vp[i] = reinterpret_cast<const uint16_t*>(vt + vlane[i]);
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int tk = 8 * t + o;
                kp[t] = kbase + (int64_t)(PAGED16 ? idx + tk : __shfl(idx, tk & 15, 16)) * p.cache_stride;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tv = 8 * (i >> 1) + 2 * j + (i & 1);
                vp[i] = vbase + (int64_t)(PAGED16 ? idx + tv : __shfl(idx, tv, 16)) * p.cache_stride;
            }
        }
    };

    uint4 kb[DEPTH][4], vb[DEPTH][4];  // kb[.][2t + k]: half row k of token 8t + o; vb[.][i]: row of token 8(i>>1) + 2j + (i&1)
    // One tile: scores from kb[st]; refill kb[st] with own tile n + DEPTH; softmax; P.V from vb[st];
    // refill vb[st].  idx_refill is the table entry of tile n + DEPTH (REFILL only).
    // FULL: every token of the tile lies inside the split (all tiles but a wave's last one), no masking.
    auto tile_step = [&](auto st_tag, auto refill_tag, auto full_tag, int n, int idx_refill) {
        constexpr int st = decltype(st_tag)::value;
        constexpr int RF = decltype(refill_tag)::value;  // 0: no refill, 1: refill with cached loads, 2: non-temporal
        constexpr bool REFILL = RF != 0;
        constexpr bool NT = RF == 2;
        constexpr bool FULL = decltype(full_tag)::value && G >= 8;  // (as above: only pays where the vector pipe is the limit)
        const uint16_t* kp[2];
        const uint16_t* vp[4];
        if (REFILL) tile_rows(idx_refill, kp, vp);

        const int tok0 = t0 + (sub + n * WPH) * kTile + o;  // this lane's K token of half t is tok0 + 8t
        // query heads are taken GS at a time (scores -> softmax -> P.V per group) so that only a few
        // scores and probabilities are live beside the G * 8 accumulators
        constexpr int GS = G < 4 ? G : 4;
#pragma unroll
        for (int g0 = 0; g0 < G; g0 += GS) {
            float s[GS][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
#pragma unroll
                for (int gg = 0; gg < GS; ++gg) {
                    const int g = g0 + gg;
                    float d = dot2_bf16_first(qv[g][0].x, kb[st][2 * t].x);
                    d = dot2_bf16(qv[g][0].y, kb[st][2 * t].y, d);
                    d = dot2_bf16(qv[g][0].z, kb[st][2 * t].z, d);
                    d = dot2_bf16(qv[g][0].w, kb[st][2 * t].w, d);
                    d = dot2_bf16(qv[g][1].x, kb[st][2 * t + 1].x, d);
                    d = dot2_bf16(qv[g][1].y, kb[st][2 * t + 1].y, d);
                    d = dot2_bf16(qv[g][1].z, kb[st][2 * t + 1].z, d);
                    d = dot2_bf16(qv[g][1].w, kb[st][2 * t + 1].w, d);
                    d = octet_sum(d);
                    s[gg][t] = (FULL || tok0 + 8 * t < t1) ? d * p.scale_log2e : -INFINITY;
                }
            }
            if (REFILL && g0 + GS >= G) {
                // this stage's K registers are free: refill them before the remaining softmax / P.V work
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int k = 0; k < 2; ++k) kb[st][2 * t + k] = ld16_kv<NT>(kp[t] + k * 64);
                __builtin_amdgcn_sched_barrier(0);
            }
            float pr[GS][2];
#pragma unroll
            for (int gg = 0; gg < GS; ++gg) {
                const int g = g0 + gg;
                float mt = fmaxf(s[gg][0], s[gg][1]);
                mt = fmaxf(mt, dpp_f32<DPP_ROW_ROR8>(mt));  // the row's other octet
                const float mn = fmaxf(m[g], mt);
                const float alpha = fast_exp2(m[g] - mn);
                pr[gg][0] = fast_exp2(s[gg][0] - mn);
                pr[gg][1] = fast_exp2(s[gg][1] - mn);
                l[g] = l[g] * alpha + (pr[gg][0] + pr[gg][1]);  // this octet's tokens only; the two octets meet after the loop
                m[g] = mn;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[g][e] *= alpha;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const uint32_t w[4] = {vb[st][i].x, vb[st][i].y, vb[st][i].z, vb[st][i].w};
                float pb[GS];  // probability of token 8(i>>1) + 2j + (i&1): held by the row's octet (i&1)
#pragma unroll
                for (int gg = 0; gg < GS; ++gg)
                    pb[gg] = (i & 1) ? dpp_f32<DPP_ROW_BCAST8>(pr[gg][i >> 1]) : dpp_f32<DPP_ROW_BCAST0>(pr[gg][i >> 1]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = bf16_lo(w[e]), hi = bf16_hi(w[e]);
#pragma unroll
                    for (int gg = 0; gg < GS; ++gg) {
                        const int g = g0 + gg;
                        acc[g][2 * e] = fmaf(pb[gg], lo, acc[g][2 * e]);
                        acc[g][2 * e + 1] = fmaf(pb[gg], hi, acc[g][2 * e + 1]);
                    }
                }
            }
            if (g0 + GS < G) __builtin_amdgcn_sched_barrier(0);  // keep the groups' live ranges apart
        }
        if (REFILL) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) vb[st][i] = ld16_kv<NT>(vp[i]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using FULL_ = std::true_type;
    using EDGE_ = std::false_type;
    using F_ = std::integral_constant<int, 0>;   // no refill
    using T_ = std::integral_constant<int, 1>;   // refill, cached loads
    using NT_ = std::integral_constant<int, 2>;  // refill, non-temporal loads

    // prologue: fill the ring with own tiles 0 .. DEPTH-1 (tiles past the end re-read the last valid one)
    int idx_next;  // table entry of the next tile to refill
    {
#pragma unroll
        for (int st = 0; st < DEPTH; ++st) {
            const uint16_t* kp[2];
            const uint16_t* vp[4];
            tile_rows(load_index(st), kp, vp);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int k = 0; k < 2; ++k) kb[st][2 * t + k] = ld16_kv<false>(kp[t] + k * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) vb[st][i] = ld16_kv<false>(vp[i]);
        }
        idx_next = load_index(DEPTH);
    }

    // steady state: every tile of the group has a refill target inside the split.  The body is
    // branch free, so the compiler's wait counts stay exact and loads stay in flight across tiles.
    // Tiles whose refill target starts below reuse_len (positions other requests may share) use cached
    // loads, the rest stream with the non-temporal hint.
    const int reuse_len = !p.kv_stream ? 0x7fffffff : (p.reuse_lens ? p.reuse_lens[b] : 0);
    int n_cached = 0;  // own tiles [0, n_cached) refill with cached loads
    if (reuse_len > t0) {
        const int tiles_below = (min(reuse_len, t1) - t0 + kTile - 1) / kTile;      // split tiles starting below reuse_len
        const int own_below = (tiles_below > sub) ? (tiles_below - sub + WPH - 1) / WPH : 0;
        n_cached = max(0, own_below - DEPTH);
        n_cached = (n_cached + DEPTH - 1) / DEPTH * DEPTH;
    }
    int n0 = 0;
    for (; n0 + 2 * DEPTH <= my_tiles && n0 < n_cached; n0 += DEPTH) {
        if constexpr (DEPTH >= 1) {
            const int idx_r = idx_next;
            idx_next = load_index(n0 + 0 + DEPTH + 1);
            tile_step(std::integral_constant<int, 0>{}, T_{}, FULL_{}, n0 + 0, idx_r);
        }
        if constexpr (DEPTH >= 2) {
            const int idx_r = idx_next;
            idx_next = load_index(n0 + 1 + DEPTH + 1);
            tile_step(std::integral_constant<int, 1>{}, T_{}, FULL_{}, n0 + 1, idx_r);
        }
    }
    for (; n0 + 2 * DEPTH <= my_tiles; n0 += DEPTH) {
        if constexpr (DEPTH >= 1) {
            const int idx_r = idx_next;
            idx_next = load_index(n0 + 0 + DEPTH + 1);
            tile_step(std::integral_constant<int, 0>{}, NT_{}, FULL_{}, n0 + 0, idx_r);
        }
        if constexpr (DEPTH >= 2) {
            const int idx_r = idx_next;
            idx_next = load_index(n0 + 1 + DEPTH + 1);
            tile_step(std::integral_constant<int, 1>{}, NT_{}, FULL_{}, n0 + 1, idx_r);
        }
    }
    // tail: at most 2*DEPTH - 1 tiles; refill only where the target exists
    if constexpr (DEPTH >= 1) {
        if (n0 + 0 < my_tiles) {
            if (n0 + 0 + DEPTH < my_tiles) {
                const int idx_r = idx_next;
                idx_next = load_index(n0 + 0 + DEPTH + 1);
                tile_step(std::integral_constant<int, 0>{}, T_{}, EDGE_{}, n0 + 0, idx_r);
            } else {
                tile_step(std::integral_constant<int, 0>{}, F_{}, EDGE_{}, n0 + 0, 0);
            }
        }
    }
    if constexpr (DEPTH >= 2) {
        if (n0 + 1 < my_tiles) {
            if (n0 + 1 + DEPTH < my_tiles) {
                const int idx_r = idx_next;
                idx_next = load_index(n0 + 1 + DEPTH + 1);
                tile_step(std::integral_constant<int, 1>{}, T_{}, EDGE_{}, n0 + 1, idx_r);
            } else {
                tile_step(std::integral_constant<int, 1>{}, F_{}, EDGE_{}, n0 + 1, 0);
            }
        }
        if (n0 + 2 < my_tiles) tile_step(std::integral_constant<int, 0>{}, F_{}, EDGE_{}, n0 + 2, 0);
    }

    // merge the 4 DPP rows of the wave (tokens were dealt round-robin to rows)
#pragma unroll
    for (int g = 0; g < G; ++g) {
        float M = fmaxf(m[g], __shfl_xor(m[g], 16));
        M = fmaxf(M, __shfl_xor(M, 32));
        const float w = fast_exp2(m[g] - M);
        float lg = (l[g] + dpp_f32<DPP_ROW_ROR8>(l[g])) * w;  // both octets of the row, then the four rows
        lg += __shfl_xor(lg, 16);
        lg += __shfl_xor(lg, 32);
        l[g] = lg;
        m[g] = M;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = acc[g][e] * w;
            a += __shfl_xor(a, 16);
            a += __shfl_xor(a, 32);
            acc[g][e] = a;
        }
    }
    if (j == 0) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            float* dst = lds_o + (wave * G + g) * kD + c * 8;
            *reinterpret_cast<float4*>(dst) = make_float4(acc[g][0], acc[g][1], acc[g][2], acc[g][3]);
            *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[g][4], acc[g][5], acc[g][6], acc[g][7]);
            if (c == 0) {
                lds_ml[(wave * G + g) * 2] = m[g];
                lds_ml[(wave * G + g) * 2 + 1] = l[g];
            }
        }
    }
    __syncthreads();
    merge_waves_and_store<G, WPH>(p, lds_o, lds_ml, h, sub, lane, b, split, nsplit == 1 && si.extra == 0);
}

// Launch order for a ragged decode batch: item = split * bs + request, cost = its number of 16-token tiles.
// All full splits cost the same and go first; the partial last split of every request follows by falling
// length, so the work groups still running when the launch drains are the shortest ones.
// sorted_dev (optional, device): 0 = write the identity order instead (uniform batches keep the split-major launch
// whose neighbours are different requests of one split); lets a captured graph switch per replay.
__global__ __launch_bounds__(kPlanThreads) void decode_plan_kernel(DecodeParams p, const int32_t* __restrict__ sorted_dev,
                                                                   int32_t* __restrict__ plan) {
    __shared__ int bins[kPlanBins];
    __shared__ int part[kPlanThreads];
    if (sorted_dev && *sorted_dev == 0) {
        for (int i = threadIdx.x; i < p.bs * p.max_splits; i += kPlanThreads) plan[i] = i;
        return;
    }
    const int chunk = split_chunk(p);
    auto tiles = [&](int i) -> int {
        const int split = i / p.bs, b = i - split * p.bs;
        const SplitInfo si = split_info(p, b, chunk);
        if (split >= si.nsplit) return -1;
        const int t0 = si.start + split * chunk;
        const int t1 = (split == p.max_splits - 1) ? si.S : min(si.S, t0 + chunk);
        return (t1 - t0 + kTile - 1) / kTile;
    };
    plan_sort_descending(p.bs * p.max_splits, tiles, plan, bins, part);
}

// Merge the split-K partials of one (request, query head) when there are many of them (long contexts at small batch:
// few rows, up to thousands of splits).  DQ work groups share a row, each owning 128 / DQ output columns, so that even
// one request covers dozens of compute units; inside a group LPR lanes hold a split's columns (16 bytes each) and the
// 256 / LPR lane rows take every (256 / LPR)-th split, four splits per lane in flight.  One pass: every lane folds its
// splits with a running maximum of its own (online softmax over the partials: no first pass for the global maximum,
// so the (max, sum) pairs and the outputs are all requested together), then the lane rows are folded by shuffles and
// the four waves through LDS.
template <int DQ>
__global__ __launch_bounds__(256) void paged_decode_merge_kernel(DecodeParams p) {
    constexpr int COLS = kD / DQ;        // columns of this work group
    constexpr int LPR = COLS / 4;        // lanes per split
    constexpr int ROWS = 256 / LPR;      // splits per pass
    constexpr int UNR = 4;
    static_assert(LPR >= 1 && LPR <= 64 && (LPR & (LPR - 1)) == 0, "lanes per split");
    __shared__ float sh[4][LPR][6];
    const int dq = blockIdx.x % DQ;
    const int row = blockIdx.x / DQ;
    const int b = row / p.Hq;
    const int hq = row - b * p.Hq;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = tid % LPR, r = tid / LPR;
    const int64_t row0 = ((int64_t)b * p.max_splits) * p.Hq + hq;
    const float* src_o = p.ws_o + row0 * kD + dq * COLS + c * 4;
    const float* src_ml = p.ws_ml + row0 * 2;
    const int64_t step = (int64_t)p.Hq * kD, step_ml = (int64_t)p.Hq * 2;
    // The first ROWS * UNR slots are requested before the number of splits is known (it hangs on a load of the request's
    // length: one round trip to memory saved); every slot index below max_splits is inside the workspace, and what
    // an unused slot holds is discarded below.
    float4 o[UNR];
    float2 ml[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        const int s = min(r + u * ROWS, p.max_splits - 1);
        ml[u] = *reinterpret_cast<const float2*>(src_ml + (int64_t)s * step_ml);
        o[u] = *reinterpret_cast<const float4*>(src_o + (int64_t)s * step);
    }
    const SplitInfo si = split_info(p, b, split_chunk(p));
    const int nsplit = min(si.nsplit + si.extra, p.max_splits);  // this kernel's partials, then the shared-prefix kernel's
    if (nsplit <= 1 && si.extra == 0) return;  // written directly by the main kernel
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float m = kNegBig, l = 0.f;
    for (int s0 = r; s0 < nsplit; s0 += ROWS * UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int s = s0 + u * ROWS;
            const bool live = s < nsplit;
            if (s0 != r) {  // (the first pass has its values already)
                ml[u] = live ? *reinterpret_cast<const float2*>(src_ml + (int64_t)s * step_ml) : make_float2(kNegBig, 0.f);
                o[u] = live ? *reinterpret_cast<const float4*>(src_o + (int64_t)s * step) : make_float4(0.f, 0.f, 0.f, 0.f);
            } else if (!live) {
                ml[u] = make_float2(kNegBig, 0.f);
                o[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        float mn = m;
#pragma unroll
        for (int u = 0; u < UNR; ++u) mn = fmaxf(mn, ml[u].x);
        const float a = fast_exp2(m - mn);
        acc.x *= a, acc.y *= a, acc.z *= a, acc.w *= a;
        l *= a;
        m = mn;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const float w = fast_exp2(ml[u].x - mn);
            acc.x = fmaf(w, o[u].x, acc.x), acc.y = fmaf(w, o[u].y, acc.y);
            acc.z = fmaf(w, o[u].z, acc.z), acc.w = fmaf(w, o[u].w, acc.w);
            l = fmaf(w, ml[u].y, l);
        }
    }
    // fold two partial results (a: this lane's, b: the other's); both maxima at kNegBig (rows past the last split) give
    // weights of one on zero sums
    auto fold = [&](float mo, float lo, const float4& ao) {
        const float mn = fmaxf(m, mo);
        const float wa = fast_exp2(m - mn), wb = fast_exp2(mo - mn);
        acc.x = acc.x * wa + ao.x * wb, acc.y = acc.y * wa + ao.y * wb;
        acc.z = acc.z * wa + ao.z * wb, acc.w = acc.w * wa + ao.w * wb;
        l = l * wa + lo * wb;
        m = mn;
    };
#pragma unroll
    for (int off = LPR; off < 64; off <<= 1) {
        const float mo = __shfl_xor(m, off), lo = __shfl_xor(l, off);
        const float4 ao = make_float4(__shfl_xor(acc.x, off), __shfl_xor(acc.y, off), __shfl_xor(acc.z, off), __shfl_xor(acc.w, off));
        fold(mo, lo, ao);
    }
    if (lane < LPR) {
        float* d = sh[wave][lane];
        d[0] = acc.x, d[1] = acc.y, d[2] = acc.z, d[3] = acc.w, d[4] = m, d[5] = l;
    }
    __syncthreads();
    if (tid < LPR) {
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float* d = sh[w][tid];
            fold(d[4], d[5], make_float4(d[0], d[1], d[2], d[3]));
        }
        const float inv = 1.0f / l;
        *reinterpret_cast<uint2*>(p.out + (int64_t)b * p.out_stride + hq * kD + dq * COLS + tid * 4) =
            make_uint2(pack_bf16(acc.x * inv, acc.y * inv), pack_bf16(acc.z * inv, acc.w * inv));
    }
}

// The same merge for at most 16 splits: one WAVE per (request, query head), no LDS and no barrier -- lane s holds split
// s's (max, sum) pair, the weights stay in registers and are broadcast with a lane read.  The partial outputs of the
// first four splits are requested before the number of splits is known (it hangs on a load of the request's length;
// slots below max_splits are inside the workspace, and what an unused one holds is discarded), the rest four at a time.
__global__ __launch_bounds__(256) void paged_decode_merge_small_kernel(DecodeParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);  // (request, query head)
    if (row >= p.bs * p.Hq) return;
    const int b = row / p.Hq;
    const int hq = row - b * p.Hq;
    const int64_t row0 = ((int64_t)b * p.max_splits) * p.Hq + hq;
    const int d = lane * 2;
    const float* src = p.ws_o + row0 * kD + d;
    const int64_t step = (int64_t)p.Hq * kD;
    float2 o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) o[u] = *reinterpret_cast<const float2*>(src + (int64_t)min(u, p.max_splits - 1) * step);
    const float2 ml_raw = *reinterpret_cast<const float2*>(p.ws_ml + (row0 + (int64_t)min(lane, p.max_splits - 1) * p.Hq) * 2);
    const SplitInfo si = split_info(p, b, split_chunk(p));
    const int nsplit = min(si.nsplit + si.extra, p.max_splits);
    if (nsplit <= 1 && si.extra == 0) return;  // written directly by the main kernel
    const float2 ml = lane < nsplit ? ml_raw : make_float2(kNegBig, 0.f);
    float M = ml.x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) M = fmaxf(M, __shfl_xor(M, off));
    const float w = lane < nsplit ? fast_exp2(ml.x - M) : 0.f;
    const float L = wave_sum(w * ml.y);
    float o0 = 0.f, o1 = 0.f;
    for (int s0 = 0; s0 < nsplit; s0 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int s = s0 + u;
            if (s0 != 0 && s < nsplit) o[u] = *reinterpret_cast<const float2*>(src + (int64_t)s * step);
            const float f = s < nsplit ? __shfl(w, s & 63) : 0.f;
            o0 = fmaf(f, s < nsplit ? o[u].x : 0.f, o0);
            o1 = fmaf(f, s < nsplit ? o[u].y : 0.f, o1);
        }
    }
    const float inv = 1.0f / L;
    *reinterpret_cast<uint32_t*>(p.out + (int64_t)b * p.out_stride + hq * kD + d) = pack_bf16(o0 * inv, o1 * inv);
}

int launch_decode_mfma(const DecodeParams& p, int G, int Hkv, hipStream_t stream);  // paged_decode_mfma.hip

// Which decode calls on 16-slot runs take the matrix-core kernel.  8 query heads per kv head: always (the vector kernel
// is bound by its arithmetic there).  4 heads per kv head: ranks with 2 or 4 kv heads (TP 4 / 2 shards of a 32 q / 8 kv
// model), where it measures 1-4 % ahead of the gather kernel at every batch size tried, and ranks with one kv head
// (2-11 % ahead, most at small batches) unless the launch has a single split per request (256 requests x 8192 tokens,
// one group each: the gather kernel is 3 % ahead); with 8 kv heads the two tie.  Tests and measurements pin the choice
// with msgl_set_variant(MSGL_VARIANT_DECODE_MFMA, 0 / 8 / 4).
static bool use_mfma_decode(int G, int Hkv, int page_size, int max_splits) {
    if (page_size % kTile != 0 || (G != 8 && G != 4)) return false;
    const int forced = variant(MSGL_VARIANT_DECODE_MFMA);
    if (forced == 0) return false;
    if (G == 8) return true;
    if (forced == 8) return false;
    if (forced == 4) return true;
    return Hkv == 2 || Hkv == 4 || (Hkv == 1 && max_splits > 1);
}

template <int G, int HPB>
static int launch_decode(const DecodeParams& p, int page_size, hipStream_t stream) {
    const dim3 grid(p.bs * p.max_splits), block(kWaves * 64);
    const size_t lds = (size_t)kWaves * G * (kD + 2) * sizeof(float);
    int rc;
    if (use_mfma_decode(G, HPB, page_size, p.max_splits)) {
        rc = launch_decode_mfma(p, G, HPB, stream);
    } else {
        if (page_size % kTile == 0)
            hipLaunchKernelGGL((paged_decode_kernel<G, HPB, true>), grid, block, lds, stream, p);
        else
            hipLaunchKernelGGL((paged_decode_kernel<G, HPB, false>), grid, block, lds, stream, p);
        rc = check_launch("paged_decode_kernel");
    }
    if (rc) return rc;
    if (p.max_splits > 1) {
        if (p.max_splits <= 16)  // few splits: a wave per row, four rows per group
            hipLaunchKernelGGL(paged_decode_merge_small_kernel, dim3((p.bs * p.Hq + 3) / 4), dim3(256), 0, stream, p);
        else
            hipLaunchKernelGGL(paged_decode_merge_kernel<8>, dim3(p.bs * p.Hq * 8), dim3(256), 0, stream, p);
        rc = check_launch("paged_decode_merge_kernel");
    }
    return rc;
}

template <int G>
static int dispatch_hpb(const DecodeParams& p, int Hkv, int page_size, hipStream_t stream) {
    switch (Hkv) {
        case 1: return launch_decode<G, 1>(p, page_size, stream);
        case 2: return launch_decode<G, 2>(p, page_size, stream);
        case 4: return launch_decode<G, 4>(p, page_size, stream);
        case 8: return launch_decode<G, 8>(p, page_size, stream);
    }
    set_error("msgl_paged_decode: Hkv=%d unsupported (1, 2, 4 or 8 local kv heads)", Hkv);
    return MSGL_E_INVALID;
}

}  // namespace msgl

using namespace msgl;

extern "C" size_t msgl_paged_decode_workspace_bytes(int bs, int Hq, int D, int max_splits) {
    if (bs <= 0 || Hq <= 0 || D <= 0 || max_splits <= 0) return 0;
    return (size_t)bs * max_splits * Hq * (D + 2) * sizeof(float);
}

static int paged_decode_checked(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                                int64_t cache_stride, const int32_t* req_to_token, int64_t r2t_stride,
                                const int32_t* req_pool_indices, const int32_t* seq_lens, void* out,
                                int64_t out_stride, void* workspace, size_t workspace_bytes, int bs, int Hq, int Hkv,
                                int D, int page_size, float scale, int max_splits, int chunk_tokens,
                                const int32_t* chunk_dev, int kv_stream, const int32_t* reuse_lens, const int32_t* plan,
                                const int32_t* start_lens, const int32_t* extra_splits, void* stream) {
    if (bs == 0) return MSGL_OK;
    MSGL_REQUIRE(bs > 0, "msgl_paged_decode: bs=%d", bs);
    MSGL_REQUIRE(cache_stride <= (int64_t)1 << 27, "msgl_paged_decode: cache_stride %lld too large", (long long)cache_stride);
    MSGL_REQUIRE(D == kD, "msgl_paged_decode: head dim %d unsupported (128 only)", D);
    MSGL_REQUIRE(Hkv > 0 && Hq % Hkv == 0, "msgl_paged_decode: Hq=%d not a multiple of Hkv=%d", Hq, Hkv);
    MSGL_REQUIRE(page_size >= 1, "msgl_paged_decode: page_size=%d", page_size);
    MSGL_REQUIRE(max_splits >= 1 && max_splits <= 4096, "msgl_paged_decode: max_splits=%d", max_splits);
    MSGL_REQUIRE(extra_splits == nullptr || (start_lens != nullptr && max_splits >= 2),
                 "msgl_paged_decode: extra_splits needs start_lens and room for them (max_splits >= 2)");
    MSGL_REQUIRE(chunk_dev != nullptr || (chunk_tokens > 0 && chunk_tokens % kTile == 0),
                 "msgl_paged_decode: chunk_tokens=%d must be a positive multiple of %d", chunk_tokens, kTile);
    MSGL_REQUIRE(q && k_cache && v_cache && req_to_token && req_pool_indices && seq_lens && out,
                 "msgl_paged_decode: null pointer argument");
    MSGL_REQUIRE(aligned16(q) && aligned16(k_cache) && aligned16(v_cache) && aligned16(out),
                 "msgl_paged_decode: q, k_cache, v_cache and out must be 16-byte aligned");
    MSGL_REQUIRE(q_stride % 8 == 0 && cache_stride % 8 == 0 && out_stride % 4 == 0,
                 "msgl_paged_decode: strides must keep 16-byte (q, cache) / 8-byte (out) row alignment");
    MSGL_REQUIRE(q_stride >= (int64_t)Hq * D && out_stride >= (int64_t)Hq * D && cache_stride >= (int64_t)Hkv * D,
                 "msgl_paged_decode: stride smaller than a row");
    const size_t need = msgl_paged_decode_workspace_bytes(bs, Hq, D, max_splits);
    if (max_splits > 1) {
        if (!workspace || workspace_bytes < need || !aligned16(workspace)) {
            set_error("msgl_paged_decode: workspace of %zu bytes (16-byte aligned) required, got %zu", need,
                      workspace_bytes);
            return MSGL_E_WORKSPACE;
        }
    }
    DecodeParams p;
    p.q = static_cast<const uint16_t*>(q);
    p.q_stride = q_stride;
    p.kc = static_cast<const uint16_t*>(k_cache);
    p.vc = static_cast<const uint16_t*>(v_cache);
    p.cache_stride = cache_stride;
    p.r2t = req_to_token;
    p.r2t_stride = r2t_stride;
    p.req_idx = req_pool_indices;
    p.seq_lens = seq_lens;
    p.out = static_cast<uint16_t*>(out);
    p.out_stride = out_stride;
    p.ws_o = static_cast<float*>(workspace);
    p.ws_ml = p.ws_o ? p.ws_o + (size_t)bs * max_splits * Hq * D : nullptr;
    p.chunk_dev = chunk_dev;
    p.reuse_lens = reuse_lens;
    p.plan = plan;
    p.start_lens = start_lens;
    p.extra_splits = extra_splits;
    p.kv_stream = kv_stream ? 1 : 0;
    p.chunk = chunk_tokens;
    p.bs = bs;
    p.Hq = Hq;
    p.max_splits = max_splits;
    p.scale_log2e = scale * 1.4426950408889634f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (Hq / Hkv) {
        case 1: return dispatch_hpb<1>(p, Hkv, page_size, st);
        case 2: return dispatch_hpb<2>(p, Hkv, page_size, st);
        case 4: return dispatch_hpb<4>(p, Hkv, page_size, st);
        case 8: return dispatch_hpb<8>(p, Hkv, page_size, st);
    }
    set_error("msgl_paged_decode: Hq/Hkv=%d unsupported (1, 2, 4 or 8)", Hq / Hkv);
    return MSGL_E_INVALID;
}

extern "C" int msgl_paged_decode(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                                 int64_t cache_stride, const int32_t* req_to_token, int64_t r2t_stride,
                                 const int32_t* req_pool_indices, const int32_t* seq_lens, void* out,
                                 int64_t out_stride, void* workspace, size_t workspace_bytes, int bs, int Hq, int Hkv,
                                 int D, int page_size, float scale, int max_splits, int chunk_tokens,
                                 const int32_t* chunk_dev, int kv_stream, const int32_t* reuse_lens, const int32_t* plan,
                                 const int32_t* start_lens, const int32_t* extra_splits, void* stream) {
    return paged_decode_checked(q, q_stride, k_cache, v_cache, cache_stride, req_to_token, r2t_stride, req_pool_indices, seq_lens, out,
                                out_stride, workspace, workspace_bytes, bs, Hq, Hkv, D, page_size, scale, max_splits, chunk_tokens,
                                chunk_dev, kv_stream, reuse_lens, plan, start_lens, extra_splits, stream);
}

// ---- the same call for callers that cannot allocate: the split workspace lives in the code object ----
// One copy per GPU the library runs on, held until the process ends.  Host code never uses the array's name as an
// address (on the host it is only a handle): the device address is looked up once per device and kept.
static __device__ __align__(256) unsigned char scratch[MSGL_PAGED_DECODE_STATIC_WORKSPACE_BYTES];

static constexpr int kMaxDevices = 64;
static void* g_scratch_ptr[kMaxDevices];  // per device: address of that device's copy of `scratch`
static std::mutex g_scratch_mutex;

// Device address of the current device's scratch array; looked up on the first call on that device.
static int scratch_for_current_device(void** ptr_out) {
    int dev = -1;
    hipError_t err = hipGetDevice(&dev);
    if (err != hipSuccess || dev < 0 || dev >= kMaxDevices) {
        set_error("msgl_paged_decode_static: cannot tell the current device (%s, device %d)", hipGetErrorString(err), dev);
        return MSGL_E_LAUNCH;
    }
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[dev] == nullptr) {
        void* ptr = nullptr;
        err = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (err != hipSuccess || ptr == nullptr) {
            set_error("msgl_paged_decode_static: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(err));
            return MSGL_E_LAUNCH;
        }
        g_scratch_ptr[dev] = ptr;
    }
    *ptr_out = g_scratch_ptr[dev];
    return MSGL_OK;
}

extern "C" int msgl_paged_decode_static(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                                        int64_t cache_stride, const int32_t* req_to_token, int64_t r2t_stride,
                                        const int32_t* req_pool_indices, const int32_t* seq_lens, void* out,
                                        int64_t out_stride, int bs, int Hq, int Hkv, int D, int page_size, float scale,
                                        int max_splits, int chunk_tokens, const int32_t* chunk_dev, int kv_stream,
                                        const int32_t* reuse_lens, const int32_t* plan, void* stream) {
    if (bs == 0) return MSGL_OK;
    // (checked before the size below is formed: a negative count would wrap it)
    MSGL_REQUIRE(bs > 0 && Hq > 0 && D == kD && max_splits >= 1 && max_splits <= 4096,
                 "msgl_paged_decode_static: bs=%d Hq=%d D=%d max_splits=%d", bs, Hq, D, max_splits);
    const size_t need = max_splits > 1 ? msgl_paged_decode_workspace_bytes(bs, Hq, D, max_splits) : 0;
    if (need > (size_t)MSGL_PAGED_DECODE_STATIC_WORKSPACE_BYTES) {
        set_error("msgl_paged_decode_static: bs=%d x max_splits=%d x Hq=%d needs %zu bytes of split workspace, the built-in one holds "
                  "%zu; lower max_splits or use msgl_paged_decode with a workspace of your own",
                  bs, max_splits, Hq, need, (size_t)MSGL_PAGED_DECODE_STATIC_WORKSPACE_BYTES);
        return MSGL_E_WORKSPACE;
    }
    void* ws = nullptr;
    if (max_splits > 1) {
        const int rc = scratch_for_current_device(&ws);
        if (rc) return rc;
    }
    // nothing in the workspace has to start at zero: the merge reads exactly the partials this call's main kernel wrote
    return paged_decode_checked(q, q_stride, k_cache, v_cache, cache_stride, req_to_token, r2t_stride, req_pool_indices, seq_lens, out,
                                out_stride, ws, ws ? (size_t)MSGL_PAGED_DECODE_STATIC_WORKSPACE_BYTES : 0, bs, Hq, Hkv, D, page_size, scale,
                                max_splits, chunk_tokens, chunk_dev, kv_stream, reuse_lens, plan, nullptr, nullptr, stream);
}

extern "C" int msgl_decode_plan(const int32_t* seq_lens, int bs, int max_splits, int chunk_tokens, const int32_t* chunk_dev,
                                const int32_t* sorted_dev, const int32_t* start_lens, int32_t* plan, void* stream) {
    if (bs == 0) return MSGL_OK;
    MSGL_REQUIRE(bs > 0 && max_splits >= 1 && max_splits <= 4096, "msgl_decode_plan: bs=%d max_splits=%d", bs, max_splits);
    MSGL_REQUIRE((int64_t)bs * max_splits <= 0x7fffffff, "msgl_decode_plan: too many items");
    MSGL_REQUIRE(chunk_dev != nullptr || (chunk_tokens > 0 && chunk_tokens % kTile == 0),
                 "msgl_decode_plan: chunk_tokens=%d must be a positive multiple of %d", chunk_tokens, kTile);
    MSGL_REQUIRE(seq_lens && plan, "msgl_decode_plan: null pointer argument");
    DecodeParams p = {};
    p.seq_lens = seq_lens;
    p.start_lens = start_lens;
    p.chunk_dev = chunk_dev;
    p.chunk = chunk_tokens;
    p.bs = bs;
    p.max_splits = max_splits;
    hipLaunchKernelGGL(decode_plan_kernel, dim3(1), dim3(kPlanThreads), 0, static_cast<hipStream_t>(stream), p, sorted_dev, plan);
    return check_launch("decode_plan_kernel");
}
