// Paged extend (prefill) attention for gfx950: variable-length query runs, bottom-right
// causal mask, K/V gathered from the slot pool through req_to_token.  MFMA bound.
//
// Work group = 4 waves = 128 consecutive query rows of one (request, query head); every
// wave owns 32 rows.  The key axis is walked in tiles of 64 tokens staged in LDS:
//   S^T = K . Q^T      v_mfma_f32_32x32x16_bf16, A = K rows from LDS (ds_read_b128 on an
//                      XOR-swizzled image), B = Q^T held in registers for the whole kernel;
//                      the accumulator has the QUERY on the lane and 16 KEYS in registers,
//                      so row max / row sum are in-lane plus one exchange with lane ^ 32;
//   O^T += V^T . P^T   the S^T accumulator, exponentiated and packed to bf16, is already
//                      the B operand (its register order fixes the key order inside a
//                      k-step); the matching A operand V^T comes from LDS through
//                      ds_read_b64_tr_b16 (hardware transpose) on a 320-byte-stride image
//                      that is bank-conflict free for that read.
// The next tile's rows are fetched into registers before the MFMAs of the current tile
// and written to LDS after them, so HBM/L2 latency hides under the matrix work.
#include "msgl_common.h"

namespace msgl {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

constexpr int kBM = 128;          // query rows per work group
constexpr int kBN = 64;           // keys per LDS tile
constexpr int kHD = 128;          // head dim
constexpr int kKRow = 256;        // bytes per K row in LDS (swizzled 16-byte chunks)
constexpr int kVRow = 320;        // bytes per V row in LDS (padded: conflict-free transposed reads)
constexpr float kNegBigE = -1.0e30f;

struct ExtendParams {
    const uint16_t* q;
    int64_t q_stride;
    const uint16_t* kc;
    const uint16_t* vc;
    uint32_t cache_stride;
    const int32_t* r2t;
    int64_t r2t_stride;
    const int32_t* req_idx;
    const int32_t* cu_q;
    const int32_t* prefix;
    uint16_t* out;
    int64_t out_stride;
    int tiles_per_req;
    int bs;
    int hkv;
    int group;  // query heads per kv head
    float scale_log2e;
};

__device__ __forceinline__ bf16x8_t as_bf16x8(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }

// Staging of one 64-key tile: thread owns 16-byte chunk c16 of keys skey, skey+16, skey+32, skey+48.
// Positions past the last visible key re-read that key (always a valid slot), so the loads are
// unconditional; the causal mask discards those columns.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
struct StageRegs {
    u32x4_t k0, k1, k2, k3, v0, v1, v2, v3;
};
struct StageSlots {
    uint32_t s0, s1, s2, s3;
};

// Pool slots of the four keys this thread stages for tile kt.  Fetched one tile ahead of the rows
// themselves, so the row loads never wait on the page-table read.
__device__ __forceinline__ StageSlots stage_slots(const int32_t* r2t_row, int kt, int skey, int kv_len) {
    const int base = kt * kBN + skey;
    StageSlots r;
    r.s0 = (uint32_t)r2t_row[min(base, kv_len - 1)];
    r.s1 = (uint32_t)r2t_row[min(base + 16, kv_len - 1)];
    r.s2 = (uint32_t)r2t_row[min(base + 32, kv_len - 1)];
    r.s3 = (uint32_t)r2t_row[min(base + 48, kv_len - 1)];
    return r;
}

__device__ __forceinline__ StageRegs stage_load(const uint16_t* kcol, const uint16_t* vcol, uint32_t cache_stride,
                                                const StageSlots& sl) {
    const uint64_t o0 = (uint64_t)sl.s0 * cache_stride;
    const uint64_t o1 = (uint64_t)sl.s1 * cache_stride;
    const uint64_t o2 = (uint64_t)sl.s2 * cache_stride;
    const uint64_t o3 = (uint64_t)sl.s3 * cache_stride;
    StageRegs r;
    r.k0 = *reinterpret_cast<const u32x4_t*>(kcol + o0);
    r.v0 = *reinterpret_cast<const u32x4_t*>(vcol + o0);
    r.k1 = *reinterpret_cast<const u32x4_t*>(kcol + o1);
    r.v1 = *reinterpret_cast<const u32x4_t*>(vcol + o1);
    r.k2 = *reinterpret_cast<const u32x4_t*>(kcol + o2);
    r.v2 = *reinterpret_cast<const u32x4_t*>(vcol + o2);
    r.k3 = *reinterpret_cast<const u32x4_t*>(kcol + o3);
    r.v3 = *reinterpret_cast<const u32x4_t*>(vcol + o3);
    return r;
}

__device__ __forceinline__ void stage_write(char* lds_k, char* lds_v, int skey, int c16, const StageRegs& r) {
    // keys skey + 16 i: (key & 15) == (skey & 15) for every i
    char* kd = lds_k + skey * kKRow + ((c16 ^ (skey & 15)) << 4);
    char* vd = lds_v + skey * kVRow + (c16 << 4);
    *reinterpret_cast<u32x4_t*>(kd) = r.k0;
    *reinterpret_cast<u32x4_t*>(kd + 16 * kKRow) = r.k1;
    *reinterpret_cast<u32x4_t*>(kd + 32 * kKRow) = r.k2;
    *reinterpret_cast<u32x4_t*>(kd + 48 * kKRow) = r.k3;
    *reinterpret_cast<u32x4_t*>(vd) = r.v0;
    *reinterpret_cast<u32x4_t*>(vd + 16 * kVRow) = r.v1;
    *reinterpret_cast<u32x4_t*>(vd + 32 * kVRow) = r.v2;
    *reinterpret_cast<u32x4_t*>(vd + 48 * kVRow) = r.v3;
}

// max over the two lanes that share a query column (lane, lane ^ 32): one v_permlane32_swap
__device__ __forceinline__ float max_with_lane_xor32(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return fmaxf(__builtin_bit_cast(float, r[0]), __builtin_bit_cast(float, r[1]));
}

template <bool B>
struct BoolTag {
    static constexpr bool value = B;
};

__global__ __launch_bounds__(256, 2) void paged_extend_kernel(ExtendParams p) {
    __shared__ __attribute__((aligned(16))) char lds[kBN * kKRow + kBN * kVRow];
    char* lds_k = lds;
    char* lds_v = lds + kBN * kKRow;

    // Block order: the last (heaviest under the causal mask) query tile of every request and head first,
    // so the long blocks start early and the short ones fill the tail.  The kv head varies fastest: with
    // 8 kv heads the `group` blocks that share a K/V stream run on the same XCD back to back.
    const int idx = blockIdx.x;
    const int hk = idx % p.hkv;
    const int g = (idx / p.hkv) % p.group;
    const int hq = hk * p.group + g;
    const int b = (idx / (p.hkv * p.group)) % p.bs;
    const int qt = p.tiles_per_req - 1 - idx / (p.hkv * p.group * p.bs);
    const int q0 = p.cu_q[b];
    const int Sq = p.cu_q[b + 1] - q0;
    if (qt * kBM >= Sq) return;
    const int prefix = p.prefix[b];
    const int kv_len = min(prefix + Sq, prefix + qt * kBM + kBM);  // keys any row of this block may see
    const int nkt = (kv_len + kBN - 1) / kBN;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31;  // query column of the S^T / O^T tiles
    const int h = lane >> 5;
    const int row0 = qt * kBM + wave * 32;  // first query row of this wave
    const int qrow = row0 + n;
    const bool qvalid = qrow < Sq;
    const int qpos = prefix + qrow;  // last key position this lane's query may see

    const int32_t* r2t_row = p.r2t + (int64_t)p.req_idx[b] * p.r2t_stride;

    // Q^T fragments: B operand, lane (n, h) holds Q[n][16*ks + 8*h .. +8)
    uint4 qf[8];
    {
        const uint16_t* qp = p.q + (int64_t)(q0 + (qvalid ? qrow : 0)) * p.q_stride + hq * kHD + h * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) qf[ks] = qvalid ? *reinterpret_cast<const uint4*>(qp + ks * 16) : make_uint4(0, 0, 0, 0);
    }

    f32x16_t o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m = kNegBigE, l = 0.f;  // m is in the scaled (log2) domain

    const int g1 = (lane >> 4) & 1, iq = (lane & 15) >> 2, ip = lane & 3;
    const char* k_lane = lds_k + n * kKRow;                                         // + sb*32 rows, swizzled chunk
    const char* v_lane = lds_v + (4 * h + iq) * kVRow + (16 * g1 + 4 * ip) * 2;     // + key-row and column offsets

    // one 32-key block: S^T, online softmax, O^T update
    auto block = [&](auto masked, int sb, int kb) {
        constexpr bool MASKED = decltype(masked)::value;
        // ---- S^T = K . Q^T ----
        f32x16_t st;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = 0.f;
        const char* krow = k_lane + sb * 32 * kKRow;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const uint4 a = *reinterpret_cast<const uint4*>(krow + (((ks * 2 + h) ^ (n & 15)) << 4));
            st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(qf[ks]), st, 0, 0, 0);
        }
        // ---- mask + online softmax (query on the lane, keys in registers); raw scores, scale folded into the exp ----
        if constexpr (MASKED) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kpos = kb + (r & 3) + 8 * (r >> 2) + 4 * h;
                st[r] = (kpos > qpos) ? -INFINITY : st[r];
            }
        }
        float mx = fmaxf(st[0], st[1]);
#pragma unroll
        for (int r = 2; r < 16; r += 2) mx = fmaxf(mx, fmaxf(st[r], st[r + 1]));
        mx = max_with_lane_xor32(mx);
        const float mn = fmaxf(m, mx * p.scale_log2e);
        const float alpha = fast_exp2(m - mn);
        m = mn;
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = fast_exp2(__builtin_fmaf(st[r], p.scale_log2e, -mn));
        // P^T fragments: registers 8s..8s+7 are k-step s; element j is key 16s + 8(j>>2) + 4h + (j&3)
        uint4 pf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
            pf[s] = make_uint4(pack_bf16(st[8 * s + 0], st[8 * s + 1]), pack_bf16(st[8 * s + 2], st[8 * s + 3]),
                               pack_bf16(st[8 * s + 4], st[8 * s + 5]), pack_bf16(st[8 * s + 6], st[8 * s + 7]));
        // row sum of the ROUNDED probabilities (the values the matrix product uses), two per v_dot2c
        constexpr uint32_t kOnes = 0x3F803F80u;
        float ps0 = dot2_bf16(pf[0].x, kOnes, 0.f), ps1 = dot2_bf16(pf[1].x, kOnes, 0.f);
        ps0 = dot2_bf16(pf[0].y, kOnes, ps0), ps1 = dot2_bf16(pf[1].y, kOnes, ps1);
        ps0 = dot2_bf16(pf[0].z, kOnes, ps0), ps1 = dot2_bf16(pf[1].z, kOnes, ps1);
        ps0 = dot2_bf16(pf[0].w, kOnes, ps0), ps1 = dot2_bf16(pf[1].w, kOnes, ps1);
        l = l * alpha + (ps0 + ps1);  // per-lane partial sum; halves are added after the loop
        if (!__all(alpha == 1.0f)) {  // the running max settles after a few tiles: usually nothing to rescale
#pragma unroll
            for (int d = 0; d < 4; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
        // ---- O^T += V^T . P^T ----
#pragma unroll
        for (int d = 0; d < 4; ++d) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const char* va = v_lane + (sb * 32 + 16 * s) * kVRow + d * 64;
                const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(va));
                const short4_t hi =
                    __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(va + 8 * kVRow));
                typedef __attribute__((ext_vector_type(8))) short short8_t;
                short8_t a;
                a.lo = lo;
                a.hi = hi;
                o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), as_bf16x8(pf[s]), o[d], 0, 0, 0);
            }
        }
    };

    // staging: thread owns 16-byte chunk c16 of keys skey, skey+16, skey+32, skey+48 of a tile
    const int c16 = tid & 15;
    const int skey = tid >> 4;
    const uint16_t* kcol = p.kc + hk * kHD + c16 * 8;
    const uint16_t* vcol = p.vc + hk * kHD + c16 * 8;
    StageSlots slots = stage_slots(r2t_row, 0, skey, kv_len);
    StageRegs stg = stage_load(kcol, vcol, p.cache_stride, slots);
    slots = stage_slots(r2t_row, min(1, nkt - 1), skey, kv_len);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();  // every wave is done reading the previous tile
        stage_write(lds_k, lds_v, skey, c16, stg);
        __syncthreads();
        stg = stage_load(kcol, vcol, p.cache_stride, slots);  // tile kt+1 (last tile: harmless re-read)
        slots = stage_slots(r2t_row, min(kt + 2, nkt - 1), skey, kv_len);

        const int k0 = kt * kBN;
        if (k0 + kBN - 1 <= prefix + row0) {  // wave-uniform: every key of the tile is visible to every row
#pragma unroll 1  // one copy of the block: unrolled, the two copies' operand reads interleave and spill
            for (int sb = 0; sb < 2; ++sb) block(BoolTag<false>{}, sb, k0 + sb * 32);
        } else {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int kb = k0 + sb * 32;            // first key position of this 32-key block
                if (kb > prefix + row0 + 31) continue;  // wave-uniform: block entirely above the diagonal
                block(BoolTag<true>{}, sb, kb);
            }
        }
    }

    l += __shfl_xor(l, 32);
    if (!qvalid) return;
    const float inv = 1.0f / l;
    uint16_t* op = p.out + (int64_t)(q0 + qrow) * p.out_stride + hq * kHD;
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int col = d * 32 + 8 * rr + 4 * h;  // registers 4rr..4rr+3 are 4 consecutive columns
            *reinterpret_cast<uint2*>(op + col) = make_uint2(pack_bf16(o[d][4 * rr] * inv, o[d][4 * rr + 1] * inv),
                                                            pack_bf16(o[d][4 * rr + 2] * inv, o[d][4 * rr + 3] * inv));
        }
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_paged_extend(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                                 int64_t cache_stride, const int32_t* req_to_token, int64_t r2t_stride,
                                 const int32_t* req_pool_indices, const int32_t* cu_seqlens_q, const int32_t* prefix_lens,
                                 void* out, int64_t out_stride, int bs, int max_seqlen_q, int Hq, int Hkv, int D,
                                 int page_size, float scale, void* stream) {
    (void)page_size;  // the table is read per token, which is valid for every page size
    if (bs == 0 || max_seqlen_q == 0) return MSGL_OK;
    MSGL_REQUIRE(bs > 0 && max_seqlen_q > 0, "msgl_paged_extend: bs=%d max_seqlen_q=%d", bs, max_seqlen_q);
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
    const int64_t gx = (int64_t)bs * p.tiles_per_req * Hq;
    MSGL_REQUIRE(gx <= 0x7fffffff, "msgl_paged_extend: grid too large");
    hipLaunchKernelGGL(paged_extend_kernel, dim3((unsigned)gx), dim3(256), 0, static_cast<hipStream_t>(stream), p);
    return check_launch("paged_extend_kernel");
}
