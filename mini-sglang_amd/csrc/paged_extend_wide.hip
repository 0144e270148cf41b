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
// form costs more issue time than the two plain adds it replaces.
__device__ __forceinline__ float add_alone(float a, float b) {
    float r;
    asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__global__ __launch_bounds__(256, 1) void paged_extend_wide_kernel(ExtendParams p) {
    constexpr int KPT = 4;     // keys staged per thread and tile
    constexpr int KSTEP = 16;  // distance between a thread's staged keys
    constexpr int ST = 3;
    extern __shared__ __attribute__((aligned(256))) char lds[];  // ST * kTileBytes
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
    const int k_lane = n * kKRow + ((h ^ (n & 15)) << 4);
    const int v_lane = kKBytes + (4 * h + iq) * kVRow + (16 * g1 + 4 * ip) * 2;

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
            const char* ka = lds + (kp ^ (uint32_t)(ks << 5));
            const uint4 a0 = *reinterpret_cast<const uint4*>(ka);
            const uint4 a1 = *reinterpret_cast<const uint4*>(ka + 32 * kKRow);
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

    const int c16 = tid & 15;
    const int skey = tid >> 4;
    const uint16_t* kcol = p.kc + hk * kHD + c16 * 8;
    const uint16_t* vcol = p.vc + hk * kHD + c16 * 8;
    StageSlots<KPT> slots = stage_slots<KPT, KSTEP>(r2t_row, 0, skey, kv_len);
    StageRegs<KPT> stg = stage_load<KPT>(kcol, vcol, p.cache_stride, slots);
    slots = stage_slots<KPT, KSTEP>(r2t_row, min(1, nkt - 1), skey, kv_len);
    stage_write<KPT, KSTEP>(lds, skey, c16, stg);
    stg = stage_load<KPT>(kcol, vcol, p.cache_stride, slots);
    slots = stage_slots<KPT, KSTEP>(r2t_row, min(2, nkt - 1), skey, kv_len);
    stage_write<KPT, KSTEP>(lds + kTileBytes, skey, c16, stg);
    stg = stage_load<KPT>(kcol, vcol, p.cache_stride, slots);
    slots = stage_slots<KPT, KSTEP>(r2t_row, min(3, nkt - 1), skey, kv_len);
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

    int cur = 0, nxt = kTileBytes, nn = 2 * kTileBytes;
    auto advance_ring = [&]() {
        const int c = cur;
        cur = nxt, nxt = nn, nn = c;
    };
    int t = 0;
    // ---- steady state: tile t+1 is visible to every row of the wave ----
    // The loop never looks at the maximum of a tile's scores.  Probabilities are taken relative to the reference maximum
    // m of the row as it stands (the exact maximum of the last tile that went through softmax_tile); what the loop
    // watches is each lane's sum of the tile's probabilities.
    // While that stays under kTileSumLimit nothing can overflow and nothing loses precision, whatever the scores did.
    // The iteration that crosses the bound is the loop's last: its probabilities are discarded, the tile is redone with
    // its exact maximum outside (O^T and the row sums rescaled), and the loop is entered again.
#ifdef EXT_STAMPS
    unsigned long long acc_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
    int iters_ = 0;
#endif
    for (;;) {
    // (one basic block: a second exit inside the body would let the compiler sink the softmax arithmetic, whose results
    // are only used an iteration later, behind the matrix instructions it is meant to run beside)
    bool go = t + 1 < full_nkt;
    bool within = true;
    while (go) {
        STAMP(0);
        __syncthreads();
        STAMP(1);
        // S^T of tile t+1 for both heads in eight fenced slices (one per k-step: the K fragments of k-step ks + 2 are
        // requested before the four matrix instructions of k-step ks), with the ring-slot writes of tile t+2, the loads
        // of tile t+3 and the first V^T fragments of tile t in the slices.
        f32x16_t s[2][2];
        short8_t vf[4], vg[4];
        uint4 ka[3][2];
        const uint32_t kp = nxt + k_lane;
        auto k_read = [&](int ks) {
            const char* a = lds + (kp ^ (uint32_t)(ks << 5));
            ka[ks % 3][0] = *reinterpret_cast<const uint4*>(a);
            ka[ks % 3][1] = *reinterpret_cast<const uint4*>(a + 32 * kKRow);
        };
        k_read(0);
        k_read(1);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) s[e][0][r] = 0.f, s[e][1][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 2 < 8) k_read(ks + 2);
            if (ks < KPT) stage_write_one<KSTEP>(lds + nn, skey, c16, stg, ks);
            else stage_load_one(kcol, vcol, p.cache_stride, slots, stg, ks - KPT);
            if (ks == 7) slots = stage_slots<KPT, KSTEP>(r2t_row, min(t + ST + 1, nkt - 1), skey, kv_len);
            if (ks >= 6) v_pair(cur, 0, 2 * (ks - 6), vf);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                s[e][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ka[ks % 3][0]), as_bf16x8(qf[e][ks]), s[e][0], 0, 0, 0);
                s[e][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ka[ks % 3][1]), as_bf16x8(qf[e][ks]), s[e][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef EXT_STAMPS
            if (ks == 0) STAMP(7);
            if (ks == 3) STAMP(8);
#endif
        }
        STAMP(2);
        float st[2][32];
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int r = 0; r < 16; ++r) st[e][r] = s[e][0][r], st[e][16 + r] = s[e][1][r];
        // O^T of tile t: 32 matrix instructions (k-step, output block, head), one fenced slice each, with the softmax of
        // tile t+1 dealt over the slices: slice g exponentiates element g of either head, adds element g-1 to the lane's
        // tile sums (two chains per head), and from k-step 1 on packs one pair of the previous k-step's elements into
        // P^T, whose old value that k-step's matrix instructions have consumed by then.
        v_step(cur, 1, vg);
        float ts[2][2];
#pragma unroll
        for (int g = 0; g < 32; ++g) {
            const int ks = g >> 3, i = g & 7, d = i >> 1, e = i & 1;
            if (g == 8) v_step(cur, 2, vf);
            if (g == 16) v_step(cur, 3, vg);
            mfma_into_agpr(o[e][d], __builtin_bit_cast(u32x4_t, (ks & 1) ? vg[d] : vf[d]), __builtin_bit_cast(u32x4_t, pf[e][ks]));
#pragma unroll
            for (int he = 0; he < 2; ++he) {
                st[he][g] = fast_exp2(__builtin_fmaf(st[he][g], p.scale_log2e, -m[he]));
                if (g == 1 || g == 2) ts[he][(g - 1) & 1] = st[he][g - 1];
                else if (g > 2) ts[he][(g - 1) & 1] = add_alone(ts[he][(g - 1) & 1], st[he][g - 1]);
            }
            if (ks >= 1) {
                const int he = i >> 2, pr = i & 3, r = 8 * (ks - 1) + 2 * pr;
                const uint32_t w = pack_bf16(st[he][r], st[he][r + 1]);
                if (pr == 0) pf[he][ks - 1].x = w;
                else if (pr == 1) pf[he][ks - 1].y = w;
                else if (pr == 2) pf[he][ks - 1].z = w;
                else pf[he][ks - 1].w = w;
            }
            __builtin_amdgcn_sched_barrier(0);
#ifdef EXT_STAMPS
            if (g == 7) STAMP(3);
            if (g == 15) STAMP(4);
            if (g == 23) STAMP(5);
#endif
        }
        uint4 n3[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            n3[e] = pack_step(st[e], 3);
            l_before[e] = l[e];
            const float tile_sum = (ts[e][1] + st[e][31]) + ts[e][0];
            l[e] += tile_sum;
            within = within && (tile_sum < kTileSumLimit);  // false for NaN and infinity as well
        }
        __builtin_amdgcn_sched_barrier(0);
        pf[0][3] = n3[0];
        pf[1][3] = n3[1];
        advance_ring();
        STAMP(6);
#ifdef EXT_STAMPS
        ++iters_;
#endif
        ++t;
        within = __all(within) != 0;
        go = (t + 1 < full_nkt) & within;
    }
    if (within) break;
    // P^T in registers belongs to tile t (now the ring's current slot) and is not to be trusted: redo that tile
    drain_matrix_pipe(o);
    {
        f32x16_t s[2][2];
        qk_tile(cur, s);
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
    }
#endif
    // ---- the diagonal tiles, then the tiles only other waves of the work group still need ----
    for (; t < nkt; ++t) {
        __syncthreads();
        stage_write<KPT, KSTEP>(lds + nn, skey, c16, stg);
        stg = stage_load<KPT>(kcol, vcol, p.cache_stride, slots);  // past the last tile: a copy of it into a free slot
        slots = stage_slots<KPT, KSTEP>(r2t_row, min(t + ST + 1, nkt - 1), skey, kv_len);
        if (t < my_nkt) {
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                rescale_o(e);
                alpha_pend[e] = 1.0f;
            }
            f32x16_t s[2][2];
            const bool more = t + 1 < my_nkt;
            if (more) qk_tile(nxt, s);
            short8_t vf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                v_step(cur, ks, vf);
                pv_step(ks, vf);
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
    constexpr int kLds = 3 * kTileBytes;
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
