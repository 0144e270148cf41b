// Paged extend (prefill) attention for gfx950, the one-wave-per-SIMD form for long key ranges.
//
// Same work item, LDS tile images, matrix products and softmax arithmetic as paged_extend_kernel<2, 3> (128 query
// rows of two query heads that share a kv head, 64-key tiles in a 3-deep LDS ring, one barrier per tile), but the
// work group is 4 waves instead of 8 and every wave owns 32 rows of BOTH heads, with the whole 512-entry register
// file of its SIMD.  What that buys:
//   - every K and V^T fragment read from LDS feeds two matrix instructions (one per head) instead of one;
//   - per matrix instruction half as many staging instructions (global loads, LDS writes, their addresses);
//   - no second wave on the SIMD that runs the same segment at the same time and competes for the matrix pipe and
//     the vector issue port in lockstep; the two heads of a wave are interleaved by program order instead.
// The two heads of a wave see the same rows, hence the same causal structure: one set of loop bounds.
#include <type_traits>
#include <utility>

#include "extend_common.h"

namespace msgl {

#ifdef EXT_STAMPS
__device__ unsigned long long g_stamps_wide[16];
#endif
// Largest sum of one lane's 32 probabilities of a tile that the steady-state loop accepts without looking at the
// scores' maximum: probabilities are taken relative to a reference maximum that only moves when this bound is crossed
// (2^21 keeps every probability below 2^21: as exact in bf16 as one below 1, and far from any f32 overflow).
constexpr float kTileSumLimit = 2097152.0f;

// O^T += A . B with the accumulator in the accumulation half of the register file (the other MFMAs of this file are
// compiled in their all-VGPR form, so that the softmax reads S^T without copies; O^T is touched by nothing but these
// instructions, the rare rescale and the epilogue, and 128 registers of it would not fit beside S^T otherwise).
// The compiler does not know that this statement is a matrix instruction: whoever reads the accumulator with anything
// else than this statement waits out the matrix pipe first (drain_matrix_pipe).  Volatile, so that it keeps its place
// relative to that wait (volatile statements are not reordered among themselves).
__device__ __forceinline__ void mfma_into_agpr(f32x16_t& acc, const u32x4_t a, const u32x4_t b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
// (every accumulator then passes through an empty statement of its own behind the wait: volatile statements keep their
// order, so no read of an accumulator can be moved in front of the wait)
__device__ __forceinline__ void drain_matrix_pipe(f32x16_t (&o)[2][4]) {
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int d = 0; d < 4; ++d) asm volatile("" : "+a"(o[e][d]));
}

// One v_add_f32 the compiler cannot pair with its neighbour into a v_pk_add_f32: beside matrix instructions the packed
// form costs more issue time than the two plain adds it replaces.  The compiler cannot see into the statement, so it is
// the caller's business that b is not the result of the instruction right before it when that is a transcendental
// (v_exp_f32): their consumer needs one wait state, and without it half the lanes read a stale value.
__device__ __forceinline__ float add_alone(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Diagnostic builds only (-DEXT_KO=mask, scripts/ab_extend.sh): parts of the steady-state loop left out to see what each
// costs; the results of such a build are wrong.  1 exponent fma, 2 exponential, 4 lane sums, 8 packs, 16 ring writes,
// 32 ring loads, 64 K fragment reads, 128 V^T fragment reads, 256 the barrier.
#ifndef EXT_KO
#define EXT_KO 0
#endif
constexpr int kKo = EXT_KO;
// Where in the 64 slices of an iteration the ring operations sit: unit u's LDS write in slice kRingOff + 8 u + kRingWrite, its
// reload in slice kRingOff + 8 u + kRingLoad, the next page-table lookup in slice kRingOff + 33.  Measured on 16k prompts
// (offset, write, load): (0, 1, 5) 3145 cycles / 1108 TFLOP/s, (8, 1, 5) 3195 / 1101, (24, 1, 5) 3070 / 1121,
// (24, 3, 7) 3078 / 1120, (24, 2, 6) 3181 / 1103, (0, 0, 4) 3288 / 1124 of 1148.
#ifndef EXT_RING_OFF
#define EXT_RING_OFF 24
#endif
#ifndef EXT_RING_WRITE
#define EXT_RING_WRITE 1
#endif
#ifndef EXT_RING_LOAD
#define EXT_RING_LOAD 5
#endif
constexpr int kRingOff = EXT_RING_OFF, kRingWrite = EXT_RING_WRITE, kRingLoad = EXT_RING_LOAD;
static_assert(kRingOff % 8 == 0 && kRingOff + 33 < 64 && kRingWrite < kRingLoad && kRingLoad < 8, "ring schedule");

// f(integral_constant<int, Lo>), ..., f(integral_constant<int, Hi - 1>): a loop whose index is a constant expression in
// the body, so that register arrays are indexed by constants before any optimisation runs
template <int Lo, int... I, typename F>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, Lo + I>{}), ...);
}
template <int Lo, int Hi, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    static_for_impl<Lo>(f, std::make_integer_sequence<int, Hi - Lo>{});
}

// Schedule of the steady-state loop's softmax over the 64 slices (one matrix instruction each) of an iteration.  Item k
// is element k / 2 of head k & 1 of the next tile's scores.  The first 32 keys' scores exist when the iteration starts,
// the other 32 are complete after slice 31: their 32 items go into slices 34..63, two slices taking two.
__host__ __device__ __forceinline__ constexpr int exp_slice(int k) {
    if (k < 32) return k + 1;
    const int j = k - 32;
    return 34 + j - (j >= 11 ? 1 : 0) - (j >= 22 ? 1 : 0);
}
// Pair pr (elements 8s + 2pr, + 1) of head e goes into P^T of k-step s two slices after its second exponential, and not
// before the fourth slice after the last matrix instruction that reads the old P^T of that k-step and head (k-steps 0,
// 1: slices 0..15; 2, 3: slices 32..47; head e on the odd or even ones).  The distance matters: a matrix instruction
// reads its B operand while it executes, not when it is issued - a pack one slice behind the last reader corrupted the
// columns of four of its eight passes.  A result past 63 means: after the last slice.
__host__ __device__ __forceinline__ constexpr int pack_slice(int e, int s, int pr) {
    const int ready = exp_slice(16 * s + 4 * pr + 2 + e) + 2;
    const int last_reader = (s < 2 ? 8 * s : 32 + 8 * (s - 2)) + 6 + e;
    return ready > last_reader + 4 ? ready : last_reader + 4;
}

// Staging of a 64-key tile by the 256 threads of the four-wave kernel: a thread owns CPT 16-byte chunks (16 / CPT
// chunks apart) of 4 / CPT keys, four units of work either way (unit u = key index * CPT + chunk index).  The more
// chunks of one key a thread owns, the less address work per tile: with CPT = 4 it is ONE page-table entry and ONE row
// address per thread and tile, the other three chunks are that address plus 64, 128 and 192 bytes (a quad of lanes reads
// 64 contiguous bytes of a K or V row and writes 64 contiguous bytes of its LDS image).  Measured on 16k-token prompts,
// cycles per steady-state iteration / TFLOP/s: CPT = 1 (four keys, four addresses) 3325 / 1061, CPT = 2 3271 / 1071,
// CPT = 4 3219 / 1095.
#ifndef EXT_CPT
#define EXT_CPT 4
#endif
template <int CPT>
struct WideStage {
    static constexpr int KPT = 4 / CPT;          // keys per thread and tile
    static constexpr int KSTEP = 16 * CPT;       // distance between a thread's keys
    static constexpr int LANES = 16 / CPT;       // threads per key row
    u32x4_t k[4], v[4];
    uint32_t slot[KPT];
    __device__ __forceinline__ static int chunk0(int tid) { return tid & (LANES - 1); }
    __device__ __forceinline__ static int skey(int tid) { return tid / LANES; }
    __device__ __forceinline__ void lookup(const int32_t* r2t_row, int kt, int sk, int kv_len) {
        const int base = kt * kBN + sk;
#pragma unroll
        for (int i = 0; i < KPT; ++i) slot[i] = (uint32_t)r2t_row[min(base + i * KSTEP, kv_len - 1)];
    }
    // kcol / vcol: the thread's first chunk of slot 0's row
    __device__ __forceinline__ void load_unit(const uint16_t* kcol, const uint16_t* vcol, uint32_t cache_stride, int u) {
        const int i = u / CPT, j = u % CPT;
        const uint64_t o = (uint64_t)slot[i] * cache_stride + j * (LANES * 8);
        k[u] = *reinterpret_cast<const u32x4_t*>(kcol + o);
        v[u] = *reinterpret_cast<const u32x4_t*>(vcol + o);
    }
    template <int KROW>
    __device__ __forceinline__ void write_unit(char* tile, int sk, int c0, int u) const {
        const int i = u / CPT, j = u % CPT;
        const int key = sk + i * KSTEP, chunk = c0 + j * LANES;
        *reinterpret_cast<u32x4_t*>(tile + key * KROW + (chunk << 4)) = k[u];
        *reinterpret_cast<u32x4_t*>(tile + kBN * KROW + key * kVRow + (chunk << 4)) = v[u];
    }
};

__global__ __launch_bounds__(256, 1) void paged_extend_wide_kernel(ExtendParams p) {
#ifndef EXT_KD
#define EXT_KD 3
#endif
    constexpr int KD = EXT_KD;  // K fragments are requested this many k-steps (of two slices) ahead of their use
    using Stage = WideStage<EXT_CPT>;
    // a ring slot: 64 K rows padded to kKRowPadded bytes (the chunks of a row in place: a lane's K fragments of the eight
    // k-steps are one address plus immediates), then 64 V rows of kVRow bytes
    constexpr int KROW = kKRowPadded, kKImage = kBN * KROW, kSlot = kKImage + kBN * kVRow;
    extern __shared__ __attribute__((aligned(256))) char lds[];  // 4 * kSlot
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;  // query column of the S^T / O^T tiles
    const int h = lane >> 5;

    // block order: see paged_extend_kernel
    const int hsets = p.group / 2;
    const int idx = blockIdx.x;
    const int hk = idx % p.hkv;
    const int hs = (idx / p.hkv) % hsets;
    const int g_split = (idx / (p.hkv * hsets)) % p.ksplit;
    const int per_item = p.hkv * hsets * p.ksplit;
    int b = (idx / per_item) % p.bs;
    int qt = p.tiles_per_req - 1 - idx / (per_item * p.bs);
    if (p.plan) {
        const int item = p.plan[idx / per_item];
        if (item < 0) return;
        b = item / p.tiles_per_req;
        qt = item - b * p.tiles_per_req;
    }
    const int q0 = p.cu_q[b];
    const int Sq = p.cu_q[b + 1] - q0;
    if (qt * kBM >= Sq) return;
    const int prefix = p.prefix[b];
    int kv_len = min(prefix + Sq, prefix + qt * kBM + kBM);
    int nkt = (kv_len + kBN - 1) / kBN;
    const int hq0 = hk * p.group + hs * 2;  // the wave's heads are hq0 and hq0 + 1
    const int row0 = qt * kBM + wave * 32;
    const int qrow = row0 + n;
    const bool qvalid = qrow < Sq;
    const int64_t out_row = q0 + qrow;
    int key0 = 0;
    if (p.ksplit > 1) {
        const int per_split = (nkt + p.ksplit - 1) / p.ksplit;
        key0 = g_split * per_split * kBN;
        if (key0 >= kv_len) {
            if (qvalid && h == 0) {
#pragma unroll
                for (int e = 0; e < 2; ++e)
                    *reinterpret_cast<float2*>(p.ws_ml + (((int64_t)out_row * (p.hkv * p.group) + hq0 + e) * p.ksplit + g_split) * 2) =
                        make_float2(kNegBigE, 0.f);
            }
            return;
        }
        kv_len = min(kv_len - key0, per_split * kBN);
        nkt = (kv_len + kBN - 1) / kBN;
    }
    const int qpos = prefix + qrow - key0;
    const int full_nkt = min(nkt, max(0, prefix + row0 + 1 - key0) / kBN);
    const int my_nkt = (row0 < Sq && prefix + row0 + 31 >= key0) ? min(nkt, (prefix + row0 + 31 - key0) / kBN + 1) : 0;
    const int32_t* r2t_row = p.r2t + (int64_t)p.req_idx[b] * p.r2t_stride + key0;
    const uint16_t* qp = p.q + (int64_t)(q0 + (qvalid ? qrow : 0)) * p.q_stride + hq0 * kHD + h * 8;

    uint4 qf[2][8];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
            qf[e][ks] = qvalid ? *reinterpret_cast<const uint4*>(qp + e * kHD + ks * 16) : make_uint4(0, 0, 0, 0);

    f32x16_t o[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[e][d][r] = 0.f;
    float m[2] = {kNegBigE, kNegBigE}, l[2] = {0.f, 0.f}, l_before[2] = {0.f, 0.f}, alpha_pend[2] = {1.0f, 1.0f};
    uint4 pf[2][4];
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int s = 0; s < 4; ++s) pf[e][s] = make_uint4(0, 0, 0, 0);

    const int g1 = (lane >> 4) & 1, iq = (lane & 15) >> 2, ip = lane & 3;
    const int k_lane = n * KROW + (h << 4);
    const int v_lane = kKImage + (4 * h + iq) * kVRow + (16 * g1 + 4 * ip) * 2;

    auto v_pair = [&](int tile, int s, int d0, short8_t (&vf)[4]) {  // V^T fragments of k-step s, output blocks d0 and d0 + 1
        const char* vp = lds + tile + v_lane + 16 * s * kVRow;
#pragma unroll
        for (int d = d0; d < d0 + 2; ++d) {
            vf[d].lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(vp + d * 64));
            vf[d].hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(vp + d * 64 + 8 * kVRow));
        }
    };
    auto v_step = [&](int tile, int s, short8_t (&vf)[4]) {
        v_pair(tile, s, 0, vf);
        v_pair(tile, s, 2, vf);
    };
    auto pv_step = [&](int s, const short8_t (&vf)[4]) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int e = 0; e < 2; ++e)
                mfma_into_agpr(o[e][d], __builtin_bit_cast(u32x4_t, vf[d]), __builtin_bit_cast(u32x4_t, pf[e][s]));
    };
    // S^T of both heads for the tile at LDS offset `tile`, no overlap with anything else
    auto qk_tile = [&](int tile, f32x16_t (&s)[2][2]) {
        const uint32_t kp = tile + k_lane;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[e][0][r] = 0.f, s[e][1][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const char* ka = lds + kp + (ks << 5);
            const uint4 a0 = *reinterpret_cast<const uint4*>(ka);
            const uint4 a1 = *reinterpret_cast<const uint4*>(ka + 32 * KROW);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                s[e][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0), as_bf16x8(qf[e][ks]), s[e][0], 0, 0, 0);
                s[e][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1), as_bf16x8(qf[e][ks]), s[e][1], 0, 0, 0);
            }
        }
    };
    auto mask_scores = [&](float (&st)[32], int k0) {
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            const int kpos = k0 + 32 * (r >> 4) + (r & 3) + 8 * ((r & 15) >> 2) + 4 * h;
            st[r] = (kpos > qpos) ? -INFINITY : st[r];
        }
    };
    auto row_max = [&](const float (&st)[32]) {
        // (compiler-visible max: this path reads the accumulators right behind the matrix instructions that write
        // them, and only for instructions it knows does the compiler insert the wait that takes; the hand-written
        // max3 chains of the steady-state loop sit several matrix instructions behind theirs)
        float mx = fmaxf(fmaxf(st[0], st[1]), st[2]);
#pragma unroll
        for (int r = 3; r < 31; r += 2) mx = fmaxf(fmaxf(mx, st[r]), st[r + 1]);
        return max_with_lane_xor32(fmaxf(mx, st[31]));
    };
    auto exp_range = [&](float (&st)[32], int lo, int hi, float mn) {
#pragma unroll
        for (int r = lo; r < hi; ++r) st[r] = fast_exp2(__builtin_fmaf(st[r], p.scale_log2e, -mn));
    };
    auto pack_step = [&](const float (&st)[32], int s) {
        return make_uint4(pack_bf16(st[8 * s + 0], st[8 * s + 1]), pack_bf16(st[8 * s + 2], st[8 * s + 3]),
                          pack_bf16(st[8 * s + 4], st[8 * s + 5]), pack_bf16(st[8 * s + 6], st[8 * s + 7]));
    };
    auto sum_step = [&](const uint4& f, float acc) {  // row sum of the ROUNDED probabilities, two per v_dot2c
        constexpr uint32_t kOnes = 0x3F803F80u;
        acc = dot2_bf16(f.x, kOnes, acc);
        acc = dot2_bf16(f.y, kOnes, acc);
        acc = dot2_bf16(f.z, kOnes, acc);
        return dot2_bf16(f.w, kOnes, acc);
    };
    auto sum_tile = [&](int e) { return sum_step(pf[e][0], sum_step(pf[e][2], 0.f)) + sum_step(pf[e][1], sum_step(pf[e][3], 0.f)); };
    auto rescale_o = [&](int e) {
        if (!__all(alpha_pend[e] == 1.0f)) {
            drain_matrix_pipe(o);
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[e][d][r] *= alpha_pend[e];
        }
    };
    // whole softmax of one head's tile with its exact maximum, no overlap (first tile, tiles on the diagonal, and a tile
    // the steady-state loop gave up on); leaves the factor O^T is still owed and the row sum up to date
    auto softmax_tile = [&](int e, const f32x16_t& s0, const f32x16_t& s1, int k0, bool masked) {
        float st[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = s0[r], st[16 + r] = s1[r];
        if (masked) mask_scores(st, k0);
        const float mn = fmaxf(m[e], row_max(st) * p.scale_log2e);
        alpha_pend[e] = fast_exp2(m[e] - mn);
        m[e] = mn;
        exp_range(st, 0, 32, mn);
#pragma unroll
        for (int s = 0; s < 4; ++s) pf[e][s] = pack_step(st, s);
        l[e] = l[e] * alpha_pend[e] + sum_tile(e);
    };

    const int c16 = Stage::chunk0(tid);
    const int skey = Stage::skey(tid);
    const uint16_t* kcol = p.kc + hk * kHD + c16 * 8;
    const uint16_t* vcol = p.vc + hk * kHD + c16 * 8;
    // The ring has four slots: in the steady state an iteration reads three tiles (V of tile t, the second half of K of
    // tile t+1, the first half of K of tile t+2) while tile t+3 is written, and a tile is first read two barriers after
    // its writes were issued.  s0 = slot of tile t, s1 = t+1, s2 = t+2, s3 = the one being written.
    int s0 = 0, s1 = kSlot, s2 = 2 * kSlot, s3 = 3 * kSlot;
    auto advance_ring = [&]() {
        const int c = s0;
        s0 = s1, s1 = s2, s2 = s3, s3 = c;
    };
    Stage stg;
    auto load_tile = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) stg.load_unit(kcol, vcol, p.cache_stride, u);
    };
    auto write_tile = [&](char* tile) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 4; ++u) stg.template write_unit<KROW>(tile, skey, c16, u);
    };
    stg.lookup(r2t_row, 0, skey, kv_len);
    load_tile();
#pragma unroll
    for (int k = 1; k <= 3; ++k) {
        stg.lookup(r2t_row, min(k, nkt - 1), skey, kv_len);
        write_tile(lds + (k - 1) * kSlot);
        load_tile();
    }
    stg.lookup(r2t_row, min(4, nkt - 1), skey, kv_len);
    __syncthreads();
    if (my_nkt > 0) {
        f32x16_t s[2][2];
        qk_tile(0, s);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            softmax_tile(e, s[e][0], s[e][1], 0, full_nkt == 0);
            alpha_pend[e] = 1.0f;  // O^T is still zero
        }
    }

    int t = 0;
    // ---- steady state: tiles t+1 and (where the loop goes on) t+2 are visible to every row of the wave ----
    // Every 64-key tile is two halves of 32 keys with an S^T accumulator pair (one per head) each, and the iteration is
    // four blocks of 16 matrix instructions:
    //   1. O^T += V^T P^T of tile t, k-steps 0 and 1     beside  the exponentials of tile t+1, first half (scores kept
    //   2. S^T of tile t+1, second half of its keys              from block 4 of the iteration before)
    //   3. O^T of tile t, k-steps 2 and 3                beside  the exponentials of tile t+1, second half
    //   4. S^T of tile t+2, first half of its keys
    // so the softmax arithmetic is dealt over all 64 matrix instructions of the iteration, one exponential per matrix
    // instruction, each matrix instruction a fenced slice of its own.  P^T of a k-step is packed into the registers of
    // the old one once that k-step's matrix instructions have been issued.
    //
    // The loop never looks at the maximum of a tile's scores.  Probabilities are taken relative to the reference maximum
    // m of the row as it stands (the exact maximum of the last tile that went through softmax_tile); what the loop
    // watches is each lane's sum of the tile's probabilities.  While that stays under kTileSumLimit nothing can overflow
    // and nothing loses precision, whatever the scores did.  The iteration that crosses the bound is the loop's last: its
    // probabilities are discarded, the tile is redone with its exact maximum outside (O^T and the row sums rescaled),
    // and the loop is entered again.
    f32x16_t sa[2], sb[2];  // S^T of a tile's first / second 32 keys, one accumulator per head
    short8_t vf[4];         // V^T fragments of k-step 0 of the tile whose O^T comes next
    auto k_frag = [&](uint32_t base, int ks) { return *reinterpret_cast<const uint4*>(lds + base + (ks << 5)); };
    if (1 < full_nkt) {
        const uint32_t kp = s1 + k_lane;
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) sa[e][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const uint4 a = k_frag(kp, ks);
#pragma unroll
            for (int e = 0; e < 2; ++e) sa[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(qf[e][ks]), sa[e], 0, 0, 0);
        }
        v_step(s0, 0, vf);
    }
#ifdef EXT_STAMPS
    unsigned long long acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
    int iters_ = 0;
    const unsigned long long loop_c0_ = last_, loop_r0_ = __builtin_amdgcn_s_memrealtime();
#endif
    for (;;) {
    // (one basic block: a second exit inside the body would let the compiler sink the softmax arithmetic, whose results
    // are only used an iteration later, behind the matrix instructions it is meant to run beside)
    bool go = t + 1 < full_nkt;
    bool within = true;
    while (go) {
        STAMP(0);
        // LDS operations of a wave complete in order, and the K fragments read in block 4 were waited for: every ring
        // write of the iteration before (all issued before those reads) has landed
        if (!(kKo & 256)) asm volatile("s_barrier" ::: "memory");
        STAMP(1);
        short8_t vg[4];
        uint4 ka[KD + 1];
        float ts[2][2];
        if (kKo) {  // whatever a diagnostic build leaves out still has a defined (and opaque) value
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(vg[i]));
#pragma unroll
            for (int i = 0; i < KD + 1; ++i) asm volatile("" : "=v"(ka[i]));
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "=v"(ts[i >> 1][i & 1]));
        }
        const uint32_t kp1 = s1 + k_lane + 32 * KROW, kp2 = s2 + k_lane;
        auto slice = [&](auto gc) __attribute__((always_inline)) {
            constexpr int g = decltype(gc)::value, blk = g >> 4, j = g & 15;
            // ---- LDS reads: V^T fragments a k-step or more ahead of their use, K fragments two k-steps ahead
            if constexpr (g == 0 && !(kKo & 128)) v_pair(s0, 1, 0, vg);
            if constexpr (g == 1 && !(kKo & 128)) v_pair(s0, 1, 2, vg);
            if constexpr (g == 8 && !(kKo & 128)) v_pair(s0, 2, 0, vf);
            if constexpr (g == 9 && !(kKo & 128)) v_pair(s0, 2, 2, vf);
            if constexpr (g == 16 && !(kKo & 128)) v_pair(s0, 3, 0, vg);
            if constexpr (g == 17 && !(kKo & 128)) v_pair(s0, 3, 2, vg);
            if constexpr (g == 56 && !(kKo & 128)) v_pair(s1, 0, 0, vf);
            if constexpr (g == 58 && !(kKo & 128)) v_pair(s1, 0, 2, vf);
            if constexpr (g >= 16 - 2 * KD && g <= 30 - 2 * KD && (g & 1) == 0 && !(kKo & 64))
                ka[((g - (16 - 2 * KD)) >> 1) % (KD + 1)] = k_frag(kp1, (g - (16 - 2 * KD)) >> 1);
            if constexpr (g >= 48 - 2 * KD && g <= 62 - 2 * KD && (g & 1) == 0 && !(kKo & 64))
                ka[((g - (48 - 2 * KD)) >> 1) % (KD + 1)] = k_frag(kp2, (g - (48 - 2 * KD)) >> 1);
            // ---- ring: tile t+3 into the free slot, tile t+4 into the registers that frees, then the slots of tile t+5
            if constexpr (g >= kRingOff && g < kRingOff + 32 && (g & 7) == kRingWrite && !(kKo & 16))
                stg.template write_unit<KROW>(lds + s3, skey, c16, (g - kRingOff) >> 3);
            if constexpr (g >= kRingOff && g < kRingOff + 32 && (g & 7) == kRingLoad && !(kKo & 32))
                stg.load_unit(kcol, vcol, p.cache_stride, (g - kRingOff) >> 3);
            if constexpr (g == kRingOff + 33) stg.lookup(r2t_row, min(t + 5, nkt - 1), skey, kv_len);
            // ---- the slice's matrix instruction
            if constexpr (blk == 0 || blk == 2) {
                constexpr int ks = (blk ? 2 : 0) + (j >> 3), d = (j & 7) >> 1, e = j & 1;
                mfma_into_agpr(o[e][d], __builtin_bit_cast(u32x4_t, (ks & 1) ? vg[d] : vf[d]), __builtin_bit_cast(u32x4_t, pf[e][ks]));
            } else {
                constexpr int ks = j >> 1, e = j & 1;
                f32x16_t& acc = blk == 1 ? sb[e] : sa[e];
                f32x16_t zero;
#pragma unroll
                for (int r = 0; r < 16; ++r) zero[r] = 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ka[ks % (KD + 1)]), as_bf16x8(qf[e][ks]), ks == 0 ? zero : acc, 0, 0, 0);
            }
            // ---- softmax of tile t+1, item k = element k / 2 of head k & 1: exponent formed in the slice before its
            // exponential (so that the two do not wait on each other), added to its lane sum (two chains per head) two
            // slices after it (a matrix instruction between the v_exp_f32 and the add: see add_alone), packed into P^T
            // two slices after its pair is complete
            static_for<0, 6>([&](auto dkc) __attribute__((always_inline)) {
                constexpr int k = g + decltype(dkc)::value - 4;
                if constexpr (k >= 0 && k < 64) {
                    constexpr int he = k & 1, el = k >> 1, c = el & 1;
                    if constexpr (exp_slice(k) - 1 == g && !(kKo & 1)) {
                        if constexpr (el < 16) sa[he][el] = __builtin_fmaf(sa[he][el], p.scale_log2e, -m[he]);
                        else sb[he][el - 16] = __builtin_fmaf(sb[he][el - 16], p.scale_log2e, -m[he]);
                    }
                    if constexpr (exp_slice(k) == g && !(kKo & 2)) {
                        if constexpr (el < 16) sa[he][el] = fast_exp2(sa[he][el]);
                        else sb[he][el - 16] = fast_exp2(sb[he][el - 16]);
                    }
                    if constexpr (exp_slice(k) + 2 == g && !(kKo & 4)) {
                        float x;
                        if constexpr (el < 16) x = sa[he][el];
                        else x = sb[he][el - 16];
                        if constexpr (el < 2) ts[he][c] = x;
                        else ts[he][c] = add_alone(ts[he][c], x);
                    }
                }
            });
            static_for<0, 32>([&](auto qc) __attribute__((always_inline)) {
                constexpr int q = decltype(qc)::value, he = q & 1, pr = (q >> 1) & 3, s = q >> 3, el = 8 * s + 2 * pr;
                if constexpr (pack_slice(he, s, pr) == g && !(kKo & 8)) {
                    float x0, x1;
                    if constexpr (el < 16) x0 = sa[he][el], x1 = sa[he][el + 1];
                    else x0 = sb[he][el - 16], x1 = sb[he][el + 1 - 16];
                    const uint32_t w = pack_bf16(x0, x1);
                    if constexpr (pr == 0) pf[he][s].x = w;
                    else if constexpr (pr == 1) pf[he][s].y = w;
                    else if constexpr (pr == 2) pf[he][s].z = w;
                    else pf[he][s].w = w;
                }
            });
            __builtin_amdgcn_sched_barrier(0);
        };
        static_for<0, 16>(slice);
        STAMP(2);
        static_for<16, 32>(slice);
        STAMP(3);
        static_for<32, 48>(slice);
        STAMP(4);
        static_for<48, 64>(slice);
        STAMP(5);
        // what did not fit: the adds of the last element of either head (plain adds: the compiler sees what precedes them)
        // and the last pair of either head (elements 30, 31) of k-step 3
        static_assert(exp_slice(63) == 63 && exp_slice(62) == 62 && exp_slice(61) == 61 && pack_slice(0, 3, 3) > 63 && pack_slice(0, 3, 2) <= 63,
                      "schedule");
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            // (plain C++ adds: the compiler pairs the two heads' into v_pk_add_f32 and sees the v_exp_f32 in front of them;
            // single v_add_f32 statements here measured 2 % slower on long prompts)
            ts[e][1] += sb[e][15];
            pf[e][3].w = pack_bf16(sb[e][14], sb[e][15]);
            l_before[e] = l[e];
            const float tile_sum = ts[e][0] + ts[e][1];
            l[e] += tile_sum;
            within = within && (tile_sum < kTileSumLimit);  // false for NaN and infinity as well
        }
        advance_ring();
        STAMP(6);
#ifdef EXT_STAMPS
        ++iters_;
#endif
        ++t;
        within = kKo ? true : __all(within) != 0;
        go = (t + 1 < full_nkt) & within;
    }
    if (within) break;
    // P^T in registers belongs to tile t (now the ring's first slot) and is not to be trusted: redo that tile
    drain_matrix_pipe(o);
    {
        f32x16_t s[2][2];
        qk_tile(s0, s);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            l[e] = l_before[e];
            softmax_tile(e, s[e][0], s[e][1], t * kBN, false);
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[e][d][r] *= alpha_pend[e];
            alpha_pend[e] = 1.0f;
        }
    }
    }
#ifdef EXT_STAMPS
    if (lane == 0 && wave == 0) {
        for (int i = 0; i < 10; ++i) atomicAdd(&g_stamps_wide[i], acc_[i]);
        atomicAdd(&g_stamps_wide[15], (unsigned long long)iters_);
        if (iters_ > 0) {  // the loop as a whole (with the rare exits to redo a tile)
            atomicAdd(&g_stamps_wide[10], (unsigned long long)__builtin_amdgcn_s_memtime() - loop_c0_);
            atomicAdd(&g_stamps_wide[11], (unsigned long long)__builtin_amdgcn_s_memrealtime() - loop_r0_);
        }
    }
#endif
    // ---- the diagonal tiles, then the tiles only other waves of the work group still need ----
    for (; t < nkt; ++t) {
        __syncthreads();
        write_tile(lds + s3);
        load_tile();  // past the last tile: a copy of it into a free slot
        stg.lookup(r2t_row, min(t + 5, nkt - 1), skey, kv_len);
        if (t < my_nkt) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                rescale_o(e);
                alpha_pend[e] = 1.0f;
            }
            f32x16_t s[2][2];
            const bool more = t + 1 < my_nkt;
            if (more) qk_tile(s1, s);
            short8_t vt[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                v_step(s0, ks, vt);
                pv_step(ks, vt);
            }
            if (more) {
#pragma unroll
                for (int e = 0; e < 2; ++e) softmax_tile(e, s[e][0], s[e][1], (t + 1) * kBN, true);
            }
        }
        advance_ring();
    }

    if (!qvalid) return;
    drain_matrix_pipe(o);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float le = l[e] + __shfl_xor(l[e], 32);
        const int hq = hq0 + e;
        if (p.ksplit > 1) {  // partial of (token, head, key run)
            const int64_t row = ((int64_t)out_row * (p.hkv * p.group) + hq) * p.ksplit + g_split;
            float* po = p.ws_o + row * kHD;
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
                    *reinterpret_cast<float4*>(po + d * 32 + 8 * rr + 4 * h) =
                        make_float4(o[e][d][4 * rr], o[e][d][4 * rr + 1], o[e][d][4 * rr + 2], o[e][d][4 * rr + 3]);
            if (h == 0) *reinterpret_cast<float2*>(p.ws_ml + row * 2) = make_float2(my_nkt > 0 ? m[e] : kNegBigE, le);
            continue;
        }
        const float inv = 1.0f / le;
        uint16_t* op = p.out + out_row * p.out_stride + hq * kHD;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int col = d * 32 + 8 * rr + 4 * h;
                *reinterpret_cast<uint2*>(op + col) = make_uint2(pack_bf16(o[e][d][4 * rr] * inv, o[e][d][4 * rr + 1] * inv),
                                                                pack_bf16(o[e][d][4 * rr + 2] * inv, o[e][d][4 * rr + 3] * inv));
            }
    }
}

int launch_extend_wide(const ExtendParams& p, int64_t blocks, hipStream_t stream) {
    constexpr int kLds = 4 * (kBN * kKRowPadded + kBN * kVRow);
    static PerDeviceOnce once;
    if (!once.ensure([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(paged_extend_wide_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kLds) == hipSuccess;
        })) {
        set_error("msgl_paged_extend: cannot reserve %d bytes of LDS", kLds);
        return MSGL_E_LAUNCH;
    }
    hipLaunchKernelGGL(paged_extend_wide_kernel, dim3((unsigned)blocks), dim3(256), kLds, stream, p);
    return check_launch("paged_extend_wide_kernel");
}

}  // namespace msgl

#ifdef EXT_STAMPS
extern "C" void msgl_extend_wide_stamps(unsigned long long* out) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(msgl::g_stamps_wide), sizeof(unsigned long long) * 16);
}
#endif
