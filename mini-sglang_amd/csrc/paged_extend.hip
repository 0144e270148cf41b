// Paged extend (prefill) attention for gfx950: variable-length query runs, bottom-right
// causal mask, K/V gathered from the slot pool through req_to_token.  MFMA bound.
//
// Work group = 4*NH waves = 128 consecutive query rows of NH query heads that share one kv
// head (NH = 2 whenever the GQA group is even: both heads consume the same staged K/V tile);
// every wave owns 32 rows of one head.  The key axis is walked in tiles of 64 tokens staged
// through registers into a 3-deep LDS ring (one barrier per tile):
//   S^T = K . Q^T      v_mfma_f32_32x32x16_bf16, A = K rows from LDS (ds_read_b128 on an
//                      XOR-swizzled image), B = Q^T held in registers for the whole kernel;
//                      the accumulator has the QUERY on the lane and 2 x 16 KEYS in registers,
//                      so row max / row sum are in-lane plus one exchange with lane ^ 32;
//   O^T += V^T . P^T   the S^T accumulator, exponentiated and packed to bf16, is already
//                      the B operand (its register order fixes the key order inside a
//                      k-step); the matching A operand V^T comes from LDS through
//                      ds_read_b64_tr_b16 (hardware transpose) on a 320-byte-stride image
//                      that is bank-conflict free for that read.
// The loop is software pipelined inside each wave: iteration t issues S^T of tile t+1, then
// the 16 matrix instructions of O^T for tile t with the softmax arithmetic of tile t+1
// placed between them, so the vector and matrix pipes run side by side.
#include <stdlib.h>

#include <type_traits>

#include "extend_common.h"
#include "plan_sort.h"

namespace msgl {

#ifdef EXT_STAMPS
__device__ unsigned long long g_stamps[16];
#endif
constexpr int SG_VALU = 0x002, SG_MFMA = 0x008, SG_TRANS = 0x400;

// ST = LDS ring depth.  3: one barrier per tile, 108 KiB, one work group per CU — the throughput form.
// 2: the ring slot of tile t+1 is only free once every wave has finished tile t-1, so the write sits between two
// barriers; 72 KiB lets two 4-wave groups share a CU, which hides the serial start-up (page table -> rows ->
// LDS -> first S^T) that dominates when a request has only a handful of key tiles.
template <int NH, int ST, bool GROUPED = false>
__global__ __launch_bounds__(256 * NH, 2) void paged_extend_kernel(ExtendParams p, GroupedParams gp) {
    static_assert(!GROUPED || NH == 1, "rows of a shared-prefix work group already span the heads of the kv head");
    constexpr int KPT = 4 / NH;          // keys staged per thread and tile
    constexpr int KSTEP = 16 * NH;       // distance between a thread's staged keys
    extern __shared__ __attribute__((aligned(256))) char lds[];  // ST * kTileBytes
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;  // query column of the S^T / O^T tiles
    const int h = lane >> 5;
    // what the tile loop needs, set by either mode: keys are positions [0, kv_len) of r2t_row, row visibility ends at qpos
    int hk, hq, kv_len, nkt, qpos, full_nkt, my_nkt, g_member = 0, g_split = 0;
    int64_t out_row = 0;
    bool qvalid;
    const int32_t* r2t_row;
    const uint16_t* qp;
    if constexpr (GROUPED) {
        int idx = blockIdx.x;
        hk = idx % p.hkv, idx /= p.hkv;
        int rt, gi;
        if (gp.items) {
            const int item = gp.items[idx];
            if (item < 0) return;
            gi = item >> 20, rt = (item >> 14) & 63, g_split = item & 0x3fff;
        } else {
            g_split = idx % gp.key_splits, idx /= gp.key_splits;
            rt = idx % gp.row_tiles;
            gi = idx / gp.row_tiles;
        }
        const int key_chunk = gp.key_chunk_dev ? *gp.key_chunk_dev : gp.key_chunk;
        const int first = gp.first[gi], rows = gp.count[gi] * p.group, pre = gp.prefix[gi];
        const int key0 = g_split * key_chunk;
        if (rt * kBM >= rows || key0 >= pre) return;
        kv_len = min(pre - key0, key_chunk);  // positions are counted from key0
        nkt = (kv_len + kBN - 1) / kBN;
        const int row0 = rt * kBM + (wave & 3) * 32;
        const int row = row0 + n;
        qvalid = row < rows;
        const int member0 = gp.members ? gp.members[first] : first;
        g_member = qvalid ? (gp.members ? gp.members[first + row / p.group] : first + row / p.group) : member0;
        hq = hk * p.group + (qvalid ? row % p.group : 0);
        qpos = kv_len - 1;  // no causal structure: every row sees the whole slice
        full_nkt = kv_len / kBN;
        my_nkt = row0 < rows ? nkt : 0;
        r2t_row = p.r2t + (int64_t)p.req_idx[member0] * p.r2t_stride + key0;
        qp = p.q + (int64_t)g_member * p.q_stride + hq * kHD + h * 8;
    } else {

    // Block order: the last (heaviest under the causal mask) query tile of every request and head first,
    // so the long blocks start early and the short ones fill the tail.  The kv head varies fastest: with
    // 8 kv heads the blocks that share a K/V stream run on the same XCD back to back.
    const int hsets = p.group / NH;  // work groups per (kv head, query tile)
    const int idx = blockIdx.x;
    hk = idx % p.hkv;
    const int hs = (idx / p.hkv) % hsets;
    g_split = (idx / (p.hkv * hsets)) % p.ksplit;  // which run of key tiles (0 when the keys are not split)
    const int per_item = p.hkv * hsets * p.ksplit;
    int b = (idx / per_item) % p.bs;
    int qt = p.tiles_per_req - 1 - idx / (per_item * p.bs);
    if (p.plan) {  // ragged batches: items sorted by their number of key tiles, longest first
        const int item = p.plan[idx / per_item];
        if (item < 0) return;
        b = item / p.tiles_per_req;
        qt = item - b * p.tiles_per_req;
    }
    const int q0 = p.cu_q[b];
    const int Sq = p.cu_q[b + 1] - q0;
    if (qt * kBM >= Sq) return;
    const int prefix = p.prefix[b];
    kv_len = min(prefix + Sq, prefix + qt * kBM + kBM);  // keys any row of this block may see
    nkt = (kv_len + kBN - 1) / kBN;
    hq = hk * p.group + hs * NH + (wave >> 2);
    const int row0 = qt * kBM + (wave & 3) * 32;  // first query row of this wave
    const int qrow = row0 + n;
    qvalid = qrow < Sq;
    out_row = q0 + qrow;
    // this group's keys are positions [key0, key0 + kv_len); everything below counts positions from key0
    int key0 = 0;
    if (p.ksplit > 1) {
        const int per_split = (nkt + p.ksplit - 1) / p.ksplit;
        key0 = g_split * per_split * kBN;
        if (key0 >= kv_len) {  // nothing left for this run: an empty partial (the merge skips it by its running max)
            if (qvalid && h == 0) *reinterpret_cast<float2*>(p.ws_ml + (((int64_t)out_row * (p.hkv * p.group) + hq) * p.ksplit + g_split) * 2) =
                    make_float2(kNegBigE, 0.f);
            return;
        }
        kv_len = min(kv_len - key0, per_split * kBN);
        nkt = (kv_len + kBN - 1) / kBN;
    }
    qpos = prefix + qrow - key0;  // last key position this lane's query may see
    // tiles [0, full_nkt) are visible to every row of the wave, [full_nkt, my_nkt) partly, the rest not at all
    full_nkt = min(nkt, max(0, prefix + row0 + 1 - key0) / kBN);
    my_nkt = (row0 < Sq && prefix + row0 + 31 >= key0) ? min(nkt, (prefix + row0 + 31 - key0) / kBN + 1) : 0;
    r2t_row = p.r2t + (int64_t)p.req_idx[b] * p.r2t_stride + key0;
    qp = p.q + (int64_t)(q0 + (qvalid ? qrow : 0)) * p.q_stride + hq * kHD + h * 8;
    }

    // Q^T fragments: B operand, lane (n, h) holds Q[n][16*ks + 8*h .. +8)
    uint4 qf[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = qvalid ? *reinterpret_cast<const uint4*>(qp + ks * 16) : make_uint4(0, 0, 0, 0);

    f32x16_t o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m = kNegBigE, l = 0.f;  // m is in the scaled (log2) domain
    float alpha_pend = 1.0f;      // factor O^T still has to take before the next tile is added to it
    uint4 pf[4];                  // P^T of the tile whose O^T update is pending: k-step s = keys 16s .. 16s+15
#pragma unroll
    for (int s = 0; s < 4; ++s) pf[s] = make_uint4(0, 0, 0, 0);

    const int g1 = (lane >> 4) & 1, iq = (lane & 15) >> 2, ip = lane & 3;
    // lane parts of the LDS read addresses (tile base added per tile)
    const int k_lane = n * kKRow + ((h ^ (n & 15)) << 4);                          // ^ (ks << 5), + 32 rows for block 1
    const int v_lane = kKBytes + (4 * h + iq) * kVRow + (16 * g1 + 4 * ip) * 2;    // + key rows and column block

    // ---- S^T (both 32-key blocks) of the tile at LDS offset `tile` ----
    auto qk_tile = [&](int tile, f32x16_t& s0, f32x16_t& s1) {
        const uint32_t kp = tile + k_lane;  // 256-byte aligned tile base: the chunk swizzle stays an XOR of the offset
#pragma unroll
        for (int r = 0; r < 16; ++r) s0[r] = 0.f, s1[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const char* ka = lds + (kp ^ (uint32_t)(ks << 5));
            const uint4 a0 = *reinterpret_cast<const uint4*>(ka);
            const uint4 a1 = *reinterpret_cast<const uint4*>(ka + 32 * kKRow);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a0), as_bf16x8(qf[ks]), s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a1), as_bf16x8(qf[ks]), s1, 0, 0, 0);
        }
    };
    // V^T fragments of output block d (32 head dims) for the 4 k-steps of the tile
    auto v_frags = [&](int tile, int d, short8_t (&vf)[4]) {
        const char* vp = lds + tile + v_lane + d * 64;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            vf[s].lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(vp + 16 * s * kVRow));
            vf[s].hi =
                __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(vp + (16 * s + 8) * kVRow));
        }
    };
    auto pv_block = [&](int d, const short8_t (&vf)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf[s]), as_bf16x8(pf[s]), o[d], 0, 0, 0);
    };
    // the same product k-step major: V^T fragments of k-step s (16 keys) for the 4 output blocks
    auto v_step = [&](int tile, int s, short8_t (&vf)[4]) {
        const char* vp = lds + tile + v_lane + 16 * s * kVRow;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            vf[d].lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(vp + d * 64));
            vf[d].hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(vp + d * 64 + 8 * kVRow));
        }
    };
    auto v_half = [&](int tile, int s, int d0, short8_t (&vf)[4]) {  // output blocks d0, d0 + 1 of the same
        const char* vp = lds + tile + v_lane + 16 * s * kVRow;
#pragma unroll
        for (int d = d0; d < d0 + 2; ++d) {
            vf[d].lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(vp + d * 64));
            vf[d].hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(vp + d * 64 + 8 * kVRow));
        }
    };
    auto pv_step = [&](int s, const short8_t (&vf)[4]) {
#pragma unroll
        for (int d = 0; d < 4; ++d)
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf[d]), as_bf16x8(pf[s]), o[d], 0, 0, 0);
    };
    // softmax pieces on the raw scores st[0..31] (st[16 b + r]: block b, register r); scale folded into the exp
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
    auto pack_step = [&](const float (&st)[32], int s) {  // registers 8s..8s+7 are k-step s; element j is key 16s + 8(j>>2) + 4h + (j&3)
        return make_uint4(pack_bf16(st[8 * s + 0], st[8 * s + 1]), pack_bf16(st[8 * s + 2], st[8 * s + 3]),
                          pack_bf16(st[8 * s + 4], st[8 * s + 5]), pack_bf16(st[8 * s + 6], st[8 * s + 7]));
    };
    // row sum of the ROUNDED probabilities (the values the matrix product uses), two per v_dot2c
    auto sum_step = [&](const uint4& f, float acc) {
        constexpr uint32_t kOnes = 0x3F803F80u;
        acc = dot2_bf16(f.x, kOnes, acc);
        acc = dot2_bf16(f.y, kOnes, acc);
        acc = dot2_bf16(f.z, kOnes, acc);
        return dot2_bf16(f.w, kOnes, acc);
    };
    auto sum_tile = [&]() { return sum_step(pf[0], sum_step(pf[2], 0.f)) + sum_step(pf[1], sum_step(pf[3], 0.f)); };
    auto rescale_o = [&](float alpha) {
        if (!__all(alpha == 1.0f)) {  // the running max settles after a few tiles: usually nothing to rescale
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
    };
    // whole softmax of one tile, no overlap (first tile and tiles on the diagonal); returns alpha
    auto softmax_tile = [&](const f32x16_t& s0, const f32x16_t& s1, int k0, bool masked) {
        float st[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = s0[r], st[16 + r] = s1[r];
        if (masked) mask_scores(st, k0);
        const float mn = fmaxf(m, row_max(st) * p.scale_log2e);
        const float alpha = fast_exp2(m - mn);
        m = mn;
        exp_range(st, 0, 32, mn);
#pragma unroll
        for (int s = 0; s < 4; ++s) pf[s] = pack_step(st, s);
        return alpha;  // the row sum of the tile is added when its O^T update is issued
    };

    // staging: thread owns 16-byte chunk c16 of keys skey + i * KSTEP of a tile
    const int c16 = tid & 15;
    const int skey = tid >> 4;
    const uint16_t* kcol = p.kc + hk * kHD + c16 * 8;
    const uint16_t* vcol = p.vc + hk * kHD + c16 * 8;
    // tile 0 (and, with three slots, tile 1) go into the ring before the first barrier, the next tile waits in registers
    StageSlots<KPT> slots = stage_slots<KPT, KSTEP>(r2t_row, 0, skey, kv_len);
    StageRegs<KPT> stg = stage_load<KPT>(kcol, vcol, p.cache_stride, slots);
    slots = stage_slots<KPT, KSTEP>(r2t_row, min(1, nkt - 1), skey, kv_len);
    stage_write<KPT, KSTEP>(lds, skey, c16, stg);
    stg = stage_load<KPT>(kcol, vcol, p.cache_stride, slots);  // (past the last tile: harmless re-read of it)
    slots = stage_slots<KPT, KSTEP>(r2t_row, min(2, nkt - 1), skey, kv_len);
    if constexpr (ST == 3) {
        stage_write<KPT, KSTEP>(lds + kTileBytes, skey, c16, stg);
        stg = stage_load<KPT>(kcol, vcol, p.cache_stride, slots);
        slots = stage_slots<KPT, KSTEP>(r2t_row, min(3, nkt - 1), skey, kv_len);
    }
    __syncthreads();
    if (my_nkt > 0) {
        f32x16_t s0, s1;
        qk_tile(0, s0, s1);
        softmax_tile(s0, s1, 0, full_nkt == 0);
    }

    int cur = 0, nxt = kTileBytes, nn = 2 * kTileBytes;  // LDS offsets of tiles t, t+1, t+2 in the ring
    auto advance_ring = [&]() {
        const int c = cur;
        if constexpr (ST == 3) cur = nxt, nxt = nn, nn = c;
        else cur = nxt, nxt = c;
    };
    // staging at the top of iteration t: three slots -> tile t+2 into the slot tile t-1 was read from, after the
    // barrier and beside the matrix work; two slots -> tile t+1 between two barriers
    auto stage_step = [&](int t) {
        __syncthreads();
        if constexpr (ST == 2) {
            stage_write<KPT, KSTEP>(lds + nxt, skey, c16, stg);
            __syncthreads();
        }
    };
    auto stage_refill = [&](int t) {
        if constexpr (ST == 3) stage_write<KPT, KSTEP>(lds + nn, skey, c16, stg);
        stg = stage_load<KPT>(kcol, vcol, p.cache_stride, slots);
        slots = stage_slots<KPT, KSTEP>(r2t_row, min(t + ST + 1, nkt - 1), skey, kv_len);
    };
    // Every iteration of either loop below: one barrier (tile t+1, written during iteration t-1, becomes
    // visible and the ring slot tile t-1 was read from becomes free); tile t+2 goes from registers into that
    // slot and the loads of tile t+3 are issued, all beside the matrix work; the pending factor and row sum
    // of tile t (whose P^T is in pf) are applied; then the matrix work.  Waves of one work group leave the
    // first loop at different t; both loops have one barrier per iteration, so the counts agree.
    int t = 0;
    // ---- steady state: tile t+1 is visible to every row of the wave ----
#ifdef EXT_STAMPS
    unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(last_)::"memory");
    int iters_ = 0;
#endif
    // One pipelined iteration: S^T and softmax of tile t+1 beside O^T of tile t.  MASKED: tile t+1 lies on the wave's
    // diagonal, its scores pass through the causal mask first (64 more vector instructions, not overlapped).
    auto pipelined = [&](auto masked_tag) {
        constexpr bool MASKED = decltype(masked_tag)::value;
        STAMP(0);
        stage_step(t);
        STAMP(1);
        rescale_o(alpha_pend);  // a branch of its own, ahead of the block below, so that nothing of that block sinks behind it

        // S^T of tile t+1, ordered by hand in eight slices (one per k-step, fenced from each other): the K fragments of
        // k-step ks + 2 are requested before the matrix instructions of k-step ks, and the row sum of tile t, the
        // ring-slot writes of tile t+2, the loads of tile t+3 and the first V^T fragments of tile t fill the slices.
        f32x16_t s0, s1;
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
        for (int r = 0; r < 16; ++r) s0[r] = 0.f, s1[r] = 0.f;
        float part[4];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            if (ks + 2 < 8) k_read(ks + 2);
            if (ks < 4) part[ks] = sum_step(pf[ks], 0.f);
            if (ks == 4) l = l * alpha_pend + ((part[0] + part[2]) + (part[1] + part[3]));
            if (ks < KPT) {
                if constexpr (ST == 3) stage_write_one<KSTEP>(lds + nn, skey, c16, stg, ks);
            } else if (ks < 2 * KPT) {
                stage_load_one(kcol, vcol, p.cache_stride, slots, stg, ks - KPT);
            }
            if (ks == min(2 * KPT, 7)) slots = stage_slots<KPT, KSTEP>(r2t_row, min(t + ST + 1, nkt - 1), skey, kv_len);
            if (ks >= 6) v_half(cur, 0, 2 * (ks - 6), vf);
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ka[ks % 3][0]), as_bf16x8(qf[ks]), s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(ka[ks % 3][1]), as_bf16x8(qf[ks]), s1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(2);
        float st[32];
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = s0[r], st[16 + r] = s1[r];
        if constexpr (MASKED) mask_scores(st, (t + 1) * kBN);
        // O^T of tile t, k-step by k-step (the consumed P^T registers free up for the new ones), with the
        // softmax of tile t+1 in its shadow.  k-step 0 beside the row max (asm max3: ordered by hand).
        v_step(cur, 1, vg);
        float mx = max3_raw(st[0], st[1], st[2]);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vf[d]), as_bf16x8(pf[0]), o[d], 0, 0, 0);
            const int r = 3 + 8 * d;  // d = 3 runs past the end: st[30] repeated, the last value (st[31]) is taken below
            mx = max9_raw(mx, st[r], st[r + 1], st[r + 2], st[r + 3], st[min(r + 4, 30)], st[min(r + 5, 30)], st[min(r + 6, 30)],
                          st[min(r + 7, 30)]);
            __builtin_amdgcn_sched_barrier(0);
        }
        STAMP(3);
        const float mn = fmaxf(m, max_with_lane_xor32(fmaxf(mx, st[31])) * p.scale_log2e);
        alpha_pend = fast_exp2(m - mn);
        m = mn;
        // k-step 1 beside the first 12 exps; P^T of k-step 0 is consumed, its registers take the new one
        v_step(cur, 2, vf);
        pv_step(1, vg);
        exp_range(st, 0, 12, mn);
        pf[0] = pack_step(st, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VALU, 4, 0);
            __builtin_amdgcn_sched_group_barrier(SG_TRANS, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(4);
        // k-step 2 beside the next 12
        v_step(cur, 3, vg);
        pv_step(2, vf);
        exp_range(st, 12, 24, mn);
        pf[1] = pack_step(st, 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VALU, 4, 0);
            __builtin_amdgcn_sched_group_barrier(SG_TRANS, 3, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        STAMP(5);
        // k-step 3 beside the last 8
        pv_step(3, vg);
        exp_range(st, 24, 32, mn);
        pf[2] = pack_step(st, 2);
        const uint4 n3 = pack_step(st, 3);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(SG_MFMA, 1, 0);
            __builtin_amdgcn_sched_group_barrier(SG_VALU, 4, 0);
            __builtin_amdgcn_sched_group_barrier(SG_TRANS, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        pf[3] = n3;
        advance_ring();
        STAMP(6);
#ifdef EXT_STAMPS
        ++iters_;
#endif
    
    };
    for (; t + 1 < full_nkt; ++t) pipelined(std::false_type{});
    // ---- the wave's diagonal tiles, same pipeline with the mask ----
    for (; t + 1 < my_nkt; ++t) pipelined(std::true_type{});
#ifdef EXT_STAMPS
    if (lane == 0 && (wave == 0 || wave == 5)) {
        for (int i = 0; i < 7; ++i) atomicAdd(&g_stamps[i], acc_[i]);
        atomicAdd(&g_stamps[15], (unsigned long long)iters_);
    }
#endif
    // ---- the wave's last tile, then the tiles only other waves of the work group still need ----
    for (; t < nkt; ++t) {
        stage_step(t);
        stage_refill(t);  // past the last tile: a copy of it into a free slot
        if (t < my_nkt) {
            rescale_o(alpha_pend);
            l = l * alpha_pend + sum_tile();
            alpha_pend = 1.0f;
            f32x16_t s0, s1;
            const bool more = t + 1 < my_nkt;
            if (more) qk_tile(nxt, s0, s1);
            short8_t vf[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                v_frags(cur, d, vf);
                pv_block(d, vf);
            }
            if (more) alpha_pend = softmax_tile(s0, s1, (t + 1) * kBN, true);
        }
        advance_ring();
    }

    l += __shfl_xor(l, 32);
    if (!qvalid) return;
    if constexpr (GROUPED) {
        // split partial of (member, head): slot = the member's own split count + this key slice
        const DecodeParams& dp = gp.dec;
        const SplitInfo si = split_info(dp, g_member, split_chunk(dp));
        if (si.nsplit + g_split >= dp.max_splits) return;  // caller error (no room for this slice): stay inside the member's slots
        const int64_t row = ((int64_t)g_member * dp.max_splits + si.nsplit + g_split) * dp.Hq + hq;
        float* po = dp.ws_o + row * kHD;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                *reinterpret_cast<float4*>(po + d * 32 + 8 * rr + 4 * h) =
                    make_float4(o[d][4 * rr], o[d][4 * rr + 1], o[d][4 * rr + 2], o[d][4 * rr + 3]);
        if (h == 0) *reinterpret_cast<float2*>(dp.ws_ml + row * 2) = make_float2(m, l);
        return;
    }
    if (p.ksplit > 1) {  // partial of (token, head, key run)
        const int64_t row = ((int64_t)out_row * (p.hkv * p.group) + hq) * p.ksplit + g_split;
        float* po = p.ws_o + row * kHD;
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                *reinterpret_cast<float4*>(po + d * 32 + 8 * rr + 4 * h) =
                    make_float4(o[d][4 * rr], o[d][4 * rr + 1], o[d][4 * rr + 2], o[d][4 * rr + 3]);
        if (h == 0) *reinterpret_cast<float2*>(p.ws_ml + row * 2) = make_float2(my_nkt > 0 ? m : kNegBigE, l);
        return;
    }
    const float inv = 1.0f / l;
    uint16_t* op = p.out + out_row * p.out_stride + hq * kHD;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int col = d * 32 + 8 * rr + 4 * h;  // registers 4rr..4rr+3 are 4 consecutive columns
            *reinterpret_cast<uint2*>(op + col) = make_uint2(pack_bf16(o[d][4 * rr] * inv, o[d][4 * rr + 1] * inv),
                                                            pack_bf16(o[d][4 * rr + 2] * inv, o[d][4 * rr + 3] * inv));
        }
}

// Launch order for a ragged batch.  Item i = b * tiles_per_req + qt stands for query rows [128 qt, 128 qt + 128)
// of request b; its cost is its number of 64-key tiles (prefix included), which the host cannot know without
// reading the lengths back.  One work group: counting sort of the items by that number, longest first, so the
// hardware's in-order dispatch becomes longest-processing-time-first scheduling; items past a request's
// length are dropped and the rest of `plan` is filled with -1.  Items of equal cost land in arbitrary order
// (LDS atomics); the attention output does not depend on the order.
__global__ __launch_bounds__(kPlanThreads) void extend_plan_kernel(const int32_t* __restrict__ cu_q, const int32_t* __restrict__ prefix, int bs,
                                                                   int tiles_per_req, int32_t* __restrict__ plan) {
    __shared__ int bins[kPlanBins];
    __shared__ int part[kPlanThreads];
    auto key_tiles = [&](int i) -> int {
        const int b = i / tiles_per_req, qt = i - b * tiles_per_req;
        const int Sq = cu_q[b + 1] - cu_q[b];
        if (qt * kBM >= Sq) return -1;
        const int pre = prefix[b];
        const int kv_len = min(pre + Sq, pre + qt * kBM + kBM);
        return (kv_len + kBN - 1) / kBN;
    };
    plan_sort_descending(bs * tiles_per_req, key_tiles, plan, bins, part);
}

// Joins the key runs of a split launch: one wave per (token, head), lane l owns output columns 2l and 2l + 1.
__global__ __launch_bounds__(256) void extend_merge_kernel(const float* __restrict__ ws_o, const float* __restrict__ ws_ml, uint16_t* __restrict__ out,
                                                           int64_t out_stride, int64_t rows, int Hq, int ksplit) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // (token, head)
    if (r >= rows * Hq) return;
    const int64_t tok = r / Hq;
    const int hq = (int)(r - tok * Hq);
    float M = kNegBigE;
    for (int s = 0; s < ksplit; ++s) M = fmaxf(M, ws_ml[(r * ksplit + s) * 2]);
    float L = 0.f, o0 = 0.f, o1 = 0.f;
    for (int s = 0; s < ksplit; ++s) {
        const float2 ml = *reinterpret_cast<const float2*>(ws_ml + (r * ksplit + s) * 2);
        if (ml.x <= 0.5f * kNegBigE) continue;  // an empty run: its output slots were never written
        const float f = fast_exp2(ml.x - M);
        const float2 o = *reinterpret_cast<const float2*>(ws_o + (r * ksplit + s) * kHD + 2 * lane);
        L += f * ml.y;
        o0 += f * o.x;
        o1 += f * o.y;
    }
    const float inv = 1.0f / L;
    *reinterpret_cast<uint32_t*>(out + tok * out_stride + hq * kHD + 2 * lane) = pack_bf16(o0 * inv, o1 * inv);
}

template <int NH, int ST, bool GROUPED = false>
static int launch_extend(const ExtendParams& p, int64_t blocks, hipStream_t stream, const GroupedParams& gp = GroupedParams()) {
    constexpr int kLds = ST * kTileBytes;
    static PerDeviceOnce once;  // the ring is larger than the 64 KiB a kernel gets by default
    if (!once.ensure([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(paged_extend_kernel<NH, ST, GROUPED>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess;
        })) {
        set_error("msgl_paged_extend: cannot reserve %d bytes of LDS", kLds);
        return MSGL_E_LAUNCH;
    }
    hipLaunchKernelGGL((paged_extend_kernel<NH, ST, GROUPED>), dim3((unsigned)blocks), dim3(256 * NH), kLds, stream, p, gp);
    return check_launch("paged_extend_kernel");
}

int launch_extend_wide(const ExtendParams& p, int64_t blocks, hipStream_t stream);  // paged_extend_wide.hip

}  // namespace msgl

using namespace msgl;

constexpr int kMaxKeySplit = 8;

#ifdef EXT_STAMPS
extern "C" void msgl_extend_stamps(unsigned long long* out, int reset) {
    (void)reset;
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(msgl::g_stamps), sizeof(unsigned long long) * 16);
}
#endif
extern "C" size_t msgl_paged_extend_workspace_bytes(int64_t total_tokens, int Hq, int D) {
    if (total_tokens <= 0 || Hq <= 0 || D <= 0) return 0;
    return (size_t)total_tokens * Hq * kMaxKeySplit * (D + 2) * sizeof(float);
}

// Key runs per work item the launch of msgl_paged_extend takes for this shape when its workspace allows (1 = no split):
// few work items with long key ranges (a short prompt on a long cached prefix, few heads per GPU) cut every item's key
// tiles until the launch covers the chip, at least eight key tiles per run.
static int extend_key_splits(int bs, int max_seqlen_q, int max_seqlen_k, int Hq, int Hkv) {
    const int group = Hq / Hkv;
    int nh = (group % 2 == 0) ? 2 : 1;
    const int nh_forced = variant(MSGL_VARIANT_EXTEND_HEADS);
    if (nh_forced > 0) nh = (nh_forced == 2 && group % 2 == 0) ? 2 : 1;
    const int64_t items = (int64_t)bs * ((max_seqlen_q + kBM - 1) / kBM) * (Hq / nh);
    const int cus = device_cus();
    if (!(max_seqlen_k >= 2048 && items < cus)) return 1;
    int ks = (int)((cus + items - 1) / items);
    const int by_keys = (max_seqlen_k / kBN) / 8;
    ks = ks > kMaxKeySplit ? kMaxKeySplit : ks;
    ks = ks > by_keys ? by_keys : ks;
    if (const int f = variant(MSGL_VARIANT_EXTEND_KEY_SPLIT); f > 0) ks = f > kMaxKeySplit ? kMaxKeySplit : f;
    return ks < 1 ? 1 : ks;
}

extern "C" int msgl_paged_extend_key_splits(int bs, int max_seqlen_q, int max_seqlen_k, int Hq, int Hkv) {
    if (bs <= 0 || max_seqlen_q <= 0 || max_seqlen_k <= 0 || Hq <= 0 || Hkv <= 0 || Hq % Hkv != 0) return 1;
    return extend_key_splits(bs, max_seqlen_q, max_seqlen_k, Hq, Hkv);
}

extern "C" size_t msgl_paged_extend_workspace_bytes_for(int64_t total_tokens, int Hq, int D, int key_splits) {
    if (total_tokens <= 0 || Hq <= 0 || D <= 0 || key_splits <= 1) return 0;
    const int ks = key_splits > kMaxKeySplit ? kMaxKeySplit : key_splits;
    return (size_t)total_tokens * Hq * ks * (D + 2) * sizeof(float);
}

extern "C" int msgl_paged_extend(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                                 int64_t cache_stride, const int32_t* req_to_token, int64_t r2t_stride,
                                 const int32_t* req_pool_indices, const int32_t* cu_seqlens_q, const int32_t* prefix_lens,
                                 void* out, int64_t out_stride, int bs, int max_seqlen_q, int max_seqlen_k, const int32_t* plan,
                                 int Hq, int Hkv, int D, int page_size, float scale, void* workspace, size_t workspace_bytes,
                                 int64_t total_tokens, void* stream) {
    (void)page_size;  // the table is read per token, which is valid for every page size
    if (bs == 0 || max_seqlen_q == 0) return MSGL_OK;
    MSGL_REQUIRE(bs > 0 && max_seqlen_q > 0 && max_seqlen_k >= 0, "msgl_paged_extend: bs=%d max_seqlen_q=%d max_seqlen_k=%d", bs,
                 max_seqlen_q, max_seqlen_k);
    MSGL_REQUIRE(D == kHD, "msgl_paged_extend: head dim %d unsupported (128 only)", D);
    MSGL_REQUIRE(Hkv > 0 && Hq > 0 && Hq % Hkv == 0 && Hq <= 65535, "msgl_paged_extend: Hq=%d Hkv=%d", Hq, Hkv);
    MSGL_REQUIRE(q && k_cache && v_cache && req_to_token && req_pool_indices && cu_seqlens_q && prefix_lens && out,
                 "msgl_paged_extend: null pointer argument");
    MSGL_REQUIRE(aligned16(q) && aligned16(k_cache) && aligned16(v_cache) && aligned16(out),
                 "msgl_paged_extend: q, k_cache, v_cache and out must be 16-byte aligned");
    MSGL_REQUIRE(q_stride % 8 == 0 && cache_stride % 8 == 0 && out_stride % 4 == 0,
                 "msgl_paged_extend: strides must keep 16-byte (q, cache) / 8-byte (out) row alignment");
    MSGL_REQUIRE(q_stride >= (int64_t)Hq * D && out_stride >= (int64_t)Hq * D && cache_stride >= (int64_t)Hkv * D,
                 "msgl_paged_extend: stride smaller than a row");
    ExtendParams p;
    p.q = static_cast<const uint16_t*>(q);
    p.q_stride = q_stride;
    p.kc = static_cast<const uint16_t*>(k_cache);
    p.vc = static_cast<const uint16_t*>(v_cache);
    MSGL_REQUIRE(cache_stride < (int64_t)1 << 31, "msgl_paged_extend: cache_stride %lld too large", (long long)cache_stride);
    p.cache_stride = (uint32_t)cache_stride;
    p.r2t = req_to_token;
    p.r2t_stride = r2t_stride;
    p.req_idx = req_pool_indices;
    p.cu_q = cu_seqlens_q;
    p.prefix = prefix_lens;
    p.out = static_cast<uint16_t*>(out);
    p.out_stride = out_stride;
    p.tiles_per_req = (max_seqlen_q + kBM - 1) / kBM;
    p.bs = bs;
    p.hkv = Hkv;
    p.group = Hq / Hkv;
    p.scale_log2e = scale * 1.4426950408889634f;
    p.plan = plan;
    p.ksplit = 1;
    p.ws_o = nullptr, p.ws_ml = nullptr;
    // query heads per work group: 2 share the staged K/V tile and give every SIMD two waves
    int nh = (p.group % 2 == 0) ? 2 : 1;
    const int nh_forced = variant(MSGL_VARIANT_EXTEND_HEADS);  // tests / measurements only
    if (nh_forced > 0) nh = (nh_forced == 2 && p.group % 2 == 0) ? 2 : 1;
    // Few work items with long key ranges (a short prompt on a long cached prefix, few heads per GPU): with a workspace
    // the key tiles of every item are cut into runs until the launch covers the chip; without one, one head per group
    // doubles the number of groups instead (one wave per SIMD: about a fifth slower per group).
    const int64_t items = (int64_t)bs * p.tiles_per_req * (Hq / nh);
    const int cus = device_cus();
    const int ks_wanted = extend_key_splits(bs, max_seqlen_q, max_seqlen_k, Hq, Hkv);
    if (ks_wanted > 1 && workspace != nullptr && total_tokens > 0) {
        MSGL_REQUIRE(aligned16(workspace), "msgl_paged_extend: workspace must be 16-byte aligned");
        const size_t per_split = (size_t)total_tokens * Hq * (D + 2) * sizeof(float);
        int ks = ks_wanted;
        if ((size_t)ks * per_split > workspace_bytes) ks = (int)(workspace_bytes / per_split);
        if (ks > 1) {
            p.ksplit = ks;
            p.ws_o = static_cast<float*>(workspace);
            p.ws_ml = p.ws_o + (size_t)total_tokens * Hq * ks * D;
        }
    }
    if (p.ksplit == 1 && nh_forced <= 0 && nh == 2 && items <= cus) nh = 1;
    const int64_t gx = (int64_t)bs * p.tiles_per_req * (Hq / nh) * p.ksplit;
    MSGL_REQUIRE(gx <= 0x7fffffff, "msgl_paged_extend: grid too large");
    // Requests of at most a dozen key tiles never reach the steady state the three-slot ring is built for; two
    // work groups per CU (two-slot ring, one head per group) overlap their start-up phases instead.
    const int ring_forced = variant(MSGL_VARIANT_EXTEND_RING);
    const bool two_slots = ring_forced > 0 ? ring_forced == 2 : (max_seqlen_k > 0 && max_seqlen_k <= kShortKv);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (two_slots) {
        p.ksplit = 1;
        const int64_t g1 = (int64_t)bs * p.tiles_per_req * Hq;
        MSGL_REQUIRE(g1 <= 0x7fffffff, "msgl_paged_extend: grid too large");
        return launch_extend<1, 2>(p, g1, st);
    }
    // Two heads per work group: 4 waves that own both heads each (one wave per SIMD, 512 registers) when the key ranges
    // are long enough for the steady state to dominate (measured +4..6 %); requests known to end within two dozen key
    // tiles spend most of their time on the diagonal, where 8 waves that own one head each are a few percent ahead.
    const int wide_forced = variant(MSGL_VARIANT_EXTEND_WIDE);
    const bool wide = nh == 2 && (wide_forced >= 0 ? wide_forced != 0 : !(max_seqlen_k > 0 && max_seqlen_k <= kWideMinKv));
    const int rc = wide ? launch_extend_wide(p, gx, st) : nh == 2 ? launch_extend<2, 3>(p, gx, st) : launch_extend<1, 3>(p, gx, st);
    if (rc || p.ksplit == 1) return rc;
    const int64_t rows = total_tokens * Hq;
    hipLaunchKernelGGL(extend_merge_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p.ws_o, p.ws_ml, p.out, p.out_stride, total_tokens, Hq,
                       p.ksplit);
    return check_launch("extend_merge_kernel");
}

extern "C" int64_t msgl_extend_plan_items(int bs, int max_seqlen_q) {
    if (bs <= 0 || max_seqlen_q <= 0) return 0;
    return (int64_t)bs * ((max_seqlen_q + kBM - 1) / kBM);
}

extern "C" int msgl_extend_plan(const int32_t* cu_seqlens_q, const int32_t* prefix_lens, int bs, int max_seqlen_q, int32_t* plan,
                                void* stream) {
    if (bs == 0 || max_seqlen_q == 0) return MSGL_OK;
    MSGL_REQUIRE(bs > 0 && max_seqlen_q > 0, "msgl_extend_plan: bs=%d max_seqlen_q=%d", bs, max_seqlen_q);
    MSGL_REQUIRE(cu_seqlens_q && prefix_lens && plan, "msgl_extend_plan: null pointer argument");
    const int64_t n = msgl_extend_plan_items(bs, max_seqlen_q);
    MSGL_REQUIRE(n <= 0x7fffffff, "msgl_extend_plan: %lld items", (long long)n);
    hipLaunchKernelGGL(extend_plan_kernel, dim3(1), dim3(kPlanThreads), 0, static_cast<hipStream_t>(stream), cu_seqlens_q, prefix_lens, bs,
                       (max_seqlen_q + kBM - 1) / kBM, plan);
    return check_launch("extend_plan_kernel");
}

extern "C" int msgl_shared_prefix_decode(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache, int64_t cache_stride,
                                         const int32_t* req_to_token, int64_t r2t_stride, const int32_t* req_pool_indices,
                                         const int32_t* seq_lens, const int32_t* start_lens, const int32_t* group_first,
                                         const int32_t* group_count, const int32_t* group_prefix, const int32_t* group_members,
                                         int num_groups, int max_members,
                                         int max_prefix, int key_chunk, void* workspace, size_t workspace_bytes, int bs, int Hq,
                                         int Hkv, int D, float scale, int max_splits, int chunk_tokens, const int32_t* chunk_dev,
                                         const int32_t* items, int num_items, const int32_t* key_chunk_dev, void* stream) {
    if (items == nullptr && (num_groups == 0 || max_members == 0 || max_prefix == 0)) return MSGL_OK;
    if (items != nullptr && num_items == 0) return MSGL_OK;
    MSGL_REQUIRE(bs > 0 && (items != nullptr ? num_items > 0 && num_items <= 65536
                                             : (num_groups > 0 && num_groups < 2048 && max_members > 0 && max_prefix > 0)),
                 "msgl_shared_prefix_decode: groups=%d members=%d prefix=%d items=%d bs=%d", num_groups, max_members, max_prefix, num_items, bs);
    MSGL_REQUIRE(D == kHD, "msgl_shared_prefix_decode: head dim %d unsupported (128 only)", D);
    MSGL_REQUIRE(Hkv > 0 && Hq > 0 && Hq % Hkv == 0, "msgl_shared_prefix_decode: Hq=%d Hkv=%d", Hq, Hkv);
    MSGL_REQUIRE(key_chunk_dev != nullptr || (key_chunk > 0 && key_chunk % kBN == 0),
                 "msgl_shared_prefix_decode: key_chunk %d must be a positive multiple of %d", key_chunk, kBN);
    MSGL_REQUIRE(max_splits >= 2 && max_splits <= 4096, "msgl_shared_prefix_decode: max_splits=%d", max_splits);
    MSGL_REQUIRE(chunk_dev != nullptr || (chunk_tokens > 0 && chunk_tokens % 16 == 0),
                 "msgl_shared_prefix_decode: chunk_tokens=%d must be a positive multiple of 16", chunk_tokens);
    MSGL_REQUIRE(q && k_cache && v_cache && req_to_token && req_pool_indices && seq_lens && start_lens && group_first && group_count &&
                     group_prefix && workspace,
                 "msgl_shared_prefix_decode: null pointer argument");
    MSGL_REQUIRE(aligned16(q) && aligned16(k_cache) && aligned16(v_cache) && aligned16(workspace),
                 "msgl_shared_prefix_decode: q, k_cache, v_cache and workspace must be 16-byte aligned");
    MSGL_REQUIRE(q_stride % 8 == 0 && cache_stride % 8 == 0 && cache_stride < (int64_t)1 << 31 && q_stride >= (int64_t)Hq * D &&
                     cache_stride >= (int64_t)Hkv * D,
                 "msgl_shared_prefix_decode: bad stride");
    const size_t need = msgl_paged_decode_workspace_bytes(bs, Hq, D, max_splits);
    if (workspace_bytes < need) {
        set_error("msgl_shared_prefix_decode: workspace of %zu bytes required, got %zu", need, workspace_bytes);
        return MSGL_E_WORKSPACE;
    }
    ExtendParams p = {};
    p.q = static_cast<const uint16_t*>(q);
    p.q_stride = q_stride;
    p.kc = static_cast<const uint16_t*>(k_cache);
    p.vc = static_cast<const uint16_t*>(v_cache);
    p.cache_stride = (uint32_t)cache_stride;
    p.r2t = req_to_token;
    p.r2t_stride = r2t_stride;
    p.req_idx = req_pool_indices;
    p.bs = bs;
    p.hkv = Hkv;
    p.group = Hq / Hkv;
    p.ksplit = 1;
    p.scale_log2e = scale * 1.4426950408889634f;
    GroupedParams g = {};
    g.first = group_first, g.count = group_count, g.prefix = group_prefix, g.members = group_members;
    g.key_chunk = key_chunk;
    g.key_chunk_dev = key_chunk_dev;
    g.items = items;
    g.key_splits = items ? 1 : (max_prefix + key_chunk - 1) / key_chunk;
    g.row_tiles = items ? 1 : (max_members * p.group + kBM - 1) / kBM;
    g.dec.seq_lens = seq_lens;
    g.dec.start_lens = start_lens;
    g.dec.chunk_dev = chunk_dev;
    g.dec.chunk = chunk_tokens;
    g.dec.bs = bs, g.dec.Hq = Hq, g.dec.max_splits = max_splits;
    g.dec.ws_o = static_cast<float*>(workspace);
    g.dec.ws_ml = g.dec.ws_o + (size_t)bs * max_splits * Hq * D;
    const int64_t blocks = (items ? (int64_t)num_items : (int64_t)num_groups * g.row_tiles * g.key_splits) * Hkv;
    MSGL_REQUIRE(blocks <= 0x7fffffff, "msgl_shared_prefix_decode: grid too large");
    // a slice is 8-16 key tiles: the two-slot ring (two work groups per CU) measured slightly ahead of the three-slot one
    return launch_extend<1, 2, true>(p, blocks, static_cast<hipStream_t>(stream), g);
}
