// Single-pass, HBM-bound layer kernels for gfx950: RMSNorm, fused residual-add RMSNorm,
// SiluAndMul, neox RoPE, KV-pool store, the fused per-head q/k-norm -> RoPE -> KV-store
// kernel, and a greedy argmax.  Every kernel moves 16 bytes per lane per access and
// touches each input element exactly once.
//
// Rounding follows the reference's torch path op for op (bf16 in, fp32 inside one op,
// one rounding per torch op), so outputs are comparable bit for bit up to the
// reduction order of the row sums and the last ulp of exp/rsqrt.
#include "msgl_common.h"

namespace msgl {

__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) {
    f[0] = bf16_lo(v.x); f[1] = bf16_hi(v.x);
    f[2] = bf16_lo(v.y); f[3] = bf16_hi(v.y);
    f[4] = bf16_lo(v.z); f[5] = bf16_hi(v.z);
    f[6] = bf16_lo(v.w); f[7] = bf16_hi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
    return make_uint4(pack_bf16(f[0], f[1]), pack_bf16(f[2], f[3]), pack_bf16(f[4], f[5]), pack_bf16(f[6], f[7]));
}

// 1/sqrt(x) to within an ulp: hardware estimate plus one Newton step.
__device__ __forceinline__ float rsqrt_nr(float x) {
    float r = __builtin_amdgcn_rsqf(x);
    return r * fmaf(-0.5f * x * r, r, 1.5f);
}

// Sum across the threads of a 256-thread block; every thread gets the total.
__device__ __forceinline__ float block256_sum(float x, float* red /* [4] in LDS */) {
    x = wave_sum(x);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = x;
    __syncthreads();
    const float t = red[0] + red[1] + red[2] + red[3];
    __syncthreads();
    return t;
}

// ---- RMSNorm, wide rows: one 256-thread block per row, NCH 16-byte chunks per thread ----
template <int NCH, bool ADD>
__global__ __launch_bounds__(256) void rmsnorm_wide_kernel(const uint16_t* x, int64_t x_stride, uint16_t* res,
                                                           const uint16_t* w, uint16_t* out, int64_t out_stride,
                                                           int cols, float eps) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const uint16_t* xr = x + (int64_t)row * x_stride;
    uint16_t* rr = ADD ? res + (int64_t)row * cols : nullptr;
    float v[NCH][8];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int col = (k * 256 + threadIdx.x) * 8;
        if (col < cols) {
            unpack8(*reinterpret_cast<const uint4*>(xr + col), v[k]);
            if (ADD) {
                float r[8];
                unpack8(*reinterpret_cast<const uint4*>(rr + col), r);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[k][e] += r[e];
                *reinterpret_cast<uint4*>(rr + col) = pack8(v[k]);  // residual' = bf16(sum); norm keeps fp32 sum
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = fmaf(v[k][e], v[k][e], ss);
        }
    }
    ss = block256_sum(ss, red);
    const float r = rsqrt_nr(ss / (float)cols + eps);
    uint16_t* orow = out + (int64_t)row * out_stride;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int col = (k * 256 + threadIdx.x) * 8;
        if (col < cols) {
            float wf[8], o[8];
            unpack8(*reinterpret_cast<const uint4*>(w + col), wf);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (v[k][e] * r) * wf[e];
            *reinterpret_cast<uint4*>(orow + col) = pack8(o);
        }
    }
}

// ---- RMSNorm, 128-wide rows (per-head q/k norm): 16 lanes per row, 16 rows per block ----
__global__ __launch_bounds__(256) void rmsnorm_d128_kernel(const uint16_t* x, int64_t x_stride, const uint16_t* w,
                                                           uint16_t* out, int64_t out_stride, int rows, float eps) {
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c = threadIdx.x & 15;
    const bool live = row < rows;
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) unpack8(*reinterpret_cast<const uint4*>(x + (int64_t)row * x_stride + c * 8), v);
    float ss = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
    ss = row16_sum(ss);
    const float r = rsqrt_nr(ss / 128.0f + eps);
    if (live) {
        float wf[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(w + c * 8), wf);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[e] * r) * wf[e];
        *reinterpret_cast<uint4*>(out + (int64_t)row * out_stride + c * 8) = pack8(o);
    }
}

// ---- SiluAndMul ----
__device__ __forceinline__ float silu_f32(float x) { return x / (1.0f + expf(-x)); }

__global__ __launch_bounds__(256) void silu_and_mul_kernel(const uint16_t* x, int64_t x_stride, uint16_t* out, int rows,
                                                           int n) {
    const int chunks = n >> 3;
    const int64_t total = (int64_t)rows * chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int row = (int)(i / chunks);
        const int col = (int)(i - (int64_t)row * chunks) * 8;
        const uint16_t* xr = x + (int64_t)row * x_stride + col;
        float a[8], b[8], o[8];
        unpack8(*reinterpret_cast<const uint4*>(xr), a);
        unpack8(*reinterpret_cast<const uint4*>(xr + n), b);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = round_bf16(silu_f32(a[e])) * b[e];
        *reinterpret_cast<uint4*>(out + (int64_t)row * n + col) = pack8(o);
    }
}

// ---- neox RoPE on 8 (x1, x2) pairs ----
// ROUND_REF: every torch bf16 op of the reference rounds (cos/sin cast, two products, one
// sum); otherwise fp32 throughout with one final rounding by the caller's pack8.
template <bool ROUND_REF>
__device__ __forceinline__ void rope8(const float (&x1)[8], const float (&x2)[8], const float* cs_lo /* cos */,
                                      const float* sn_lo /* sin */, float (&y1)[8], float (&y2)[8]) {
    const float4 c0 = *reinterpret_cast<const float4*>(cs_lo), c1 = *reinterpret_cast<const float4*>(cs_lo + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(sn_lo), s1 = *reinterpret_cast<const float4*>(sn_lo + 4);
    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        if (ROUND_REF) {
            const float cb = round_bf16(cs[e]), sb = round_bf16(sn[e]);
            y1[e] = round_bf16(x1[e] * cb) - round_bf16(x2[e] * sb);
            y2[e] = round_bf16(x1[e] * sb) + round_bf16(x2[e] * cb);
        } else {
            y1[e] = x1[e] * cs[e] - x2[e] * sn[e];
            y2[e] = x1[e] * sn[e] + x2[e] * cs[e];
        }
    }
}

// One thread = 8 pairs of one head of one token; 8 threads per 128-wide head.
template <bool ROUND_REF>
__global__ __launch_bounds__(256) void rope_neox_kernel(const int32_t* positions, uint16_t* q, int64_t q_stride,
                                                        uint16_t* k, int64_t k_stride, const float* cache, int T, int Hq,
                                                        int Hkv, int D) {
    const int tpt = (D >> 4);  // threads per head
    const int H = Hq + Hkv;
    const int64_t total = (int64_t)T * H * tpt;
    const int half = D >> 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int c = (int)(i % tpt);
        const int64_t th = i / tpt;
        const int hh = (int)(th % H);
        const int t = (int)(th / H);
        uint16_t* base = (hh < Hq) ? q + (int64_t)t * q_stride + hh * D : k + (int64_t)t * k_stride + (hh - Hq) * D;
        const float* cs = cache + (int64_t)positions[t] * D + c * 8;
        float x1[8], x2[8], y1[8], y2[8];
        unpack8(*reinterpret_cast<const uint4*>(base + c * 8), x1);
        unpack8(*reinterpret_cast<const uint4*>(base + half + c * 8), x2);
        rope8<ROUND_REF>(x1, x2, cs, cs + half, y1, y2);
        *reinterpret_cast<uint4*>(base + c * 8) = pack8(y1);
        *reinterpret_cast<uint4*>(base + half + c * 8) = pack8(y2);
    }
}

// ---- KV-pool store: 16 bytes per thread ----
__global__ __launch_bounds__(256) void store_kv_kernel(const uint16_t* k, int64_t k_stride, const uint16_t* v,
                                                       int64_t v_stride, const int32_t* loc, uint16_t* kc, uint16_t* vc,
                                                       int64_t cache_stride, int T, int row_chunks) {
    const int64_t total = (int64_t)T * row_chunks;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int t = (int)(i / row_chunks);
        const int col = (int)(i - (int64_t)t * row_chunks) * 8;
        const int64_t dst = (int64_t)loc[t] * cache_stride + col;
        *reinterpret_cast<uint4*>(kc + dst) = *reinterpret_cast<const uint4*>(k + (int64_t)t * k_stride + col);
        *reinterpret_cast<uint4*>(vc + dst) = *reinterpret_cast<const uint4*>(v + (int64_t)t * v_stride + col);
    }
}

// ---- fused per-head RMSNorm -> RoPE -> KV store over one layer's qkv rows (D = 128) ----
// 16 lanes per (token, head): lane c holds columns 8c..8c+7; the rotate-half partner of
// lane c is lane c^8, reached with a DPP row rotate by 8.
template <bool ROUND_REF>
__global__ __launch_bounds__(256) void qknorm_rope_store_kernel(uint16_t* qkv, int64_t qkv_stride, const uint16_t* qw,
                                                                const uint16_t* kw, float eps, const int32_t* positions,
                                                                const float* cache, const int32_t* loc, uint16_t* kc,
                                                                uint16_t* vc, int64_t cache_stride, int T, int Hq,
                                                                int Hkv) {
    constexpr int D = 128;
    const int H = Hq + 2 * Hkv;
    const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);  // (token, head)
    const int c = threadIdx.x & 15;
    const bool live = item < (int64_t)T * H;
    const int t = live ? (int)(item / H) : 0;
    const int hh = live ? (int)(item - (int64_t)t * H) : 0;
    uint16_t* src = qkv + (int64_t)t * qkv_stride + hh * D + c * 8;
    uint4 raw = make_uint4(0, 0, 0, 0);
    if (live) raw = *reinterpret_cast<const uint4*>(src);
    const bool is_v = hh >= Hq + Hkv;
    const bool is_k = !is_v && hh >= Hq;
    if (!is_v) {  // uniform within the 16-lane row; DPP below stays inside the row
        float v[8];
        unpack8(raw, v);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
        ss = row16_sum(ss);
        const float r = rsqrt_nr(ss / 128.0f + eps);
        float wf[8], xn[8], xp[8];
        unpack8(*reinterpret_cast<const uint4*>((is_k ? kw : qw) + c * 8), wf);
#pragma unroll
        for (int e = 0; e < 8; ++e) xn[e] = round_bf16((v[e] * r) * wf[e]);  // q_norm / k_norm output is bf16
#pragma unroll
        for (int e = 0; e < 8; ++e) xp[e] = dpp_f32<0x128>(xn[e]);  // row_ror:8 -> partner half
        const float* cs = cache + (int64_t)positions[t] * D + (c & 7) * 8;
        float y1[8], y2[8];
        if (c < 8)
            rope8<ROUND_REF>(xn, xp, cs, cs + 64, y1, y2);  // own = x1, partner = x2 -> keep y1
        else
            rope8<ROUND_REF>(xp, xn, cs, cs + 64, y1, y2);  // own = x2, partner = x1 -> keep y2
        raw = (c < 8) ? pack8(y1) : pack8(y2);
        if (live) *reinterpret_cast<uint4*>(src) = raw;
    }
    if (live && loc != nullptr && (is_k || is_v)) {
        const int hk = is_k ? hh - Hq : hh - Hq - Hkv;
        uint16_t* dst = (is_k ? kc : vc) + (int64_t)loc[t] * cache_stride + hk * D + c * 8;
        *reinterpret_cast<uint4*>(dst) = raw;
    }
}

// The same work for large T: a 16-lane row owns HPR consecutive heads of one token, so the position's
// cos / sin entries and the norm weights are read once per row instead of once per head, and the HPR
// 16-byte loads of the row are all in flight before the first head is processed.
template <bool ROUND_REF, int HPR>
__global__ __launch_bounds__(256) void qknorm_rope_store_rows_kernel(uint16_t* qkv, int64_t qkv_stride, const uint16_t* qw,
                                                                     const uint16_t* kw, float eps, const int32_t* positions,
                                                                     const float* cache, const int32_t* loc, uint16_t* kc,
                                                                     uint16_t* vc, int64_t cache_stride, int T, int Hq,
                                                                     int Hkv) {
    constexpr int D = 128;
    const int H = Hq + 2 * Hkv;
    const int groups = (H + HPR - 1) / HPR;  // rows per token
    const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int c = threadIdx.x & 15;
    if (item >= (int64_t)T * groups) return;  // whole 16-lane rows leave together
    const int t = (int)(item / groups);
    const int h0 = (int)(item - (int64_t)t * groups) * HPR;
    uint16_t* src = qkv + (int64_t)t * qkv_stride + h0 * D + c * 8;
    uint4 raw[HPR];
#pragma unroll
    for (int i = 0; i < HPR; ++i) raw[i] = (h0 + i < H) ? *reinterpret_cast<const uint4*>(src + i * D) : make_uint4(0, 0, 0, 0);
    float wq[8], wk[8], cs[8], sn[8];
    unpack8(*reinterpret_cast<const uint4*>(qw + c * 8), wq);
    unpack8(*reinterpret_cast<const uint4*>(kw + c * 8), wk);
    {
        const float* cp = cache + (int64_t)positions[t] * D + (c & 7) * 8;
        const float4 c0 = *reinterpret_cast<const float4*>(cp), c1 = *reinterpret_cast<const float4*>(cp + 4);
        const float4 s0 = *reinterpret_cast<const float4*>(cp + 64), s1 = *reinterpret_cast<const float4*>(cp + 68);
        const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        const float ss[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[e] = cc[e], sn[e] = ss[e];
    }
    const int64_t slot = loc != nullptr ? (int64_t)loc[t] * cache_stride : 0;
#pragma unroll
    for (int i = 0; i < HPR; ++i) {
        const int hh = h0 + i;
        if (hh >= H) break;
        const bool is_v = hh >= Hq + Hkv;
        const bool is_k = !is_v && hh >= Hq;
        uint4 r4 = raw[i];
        if (!is_v) {
            float v[8];
            unpack8(r4, v);
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
            ss = row16_sum(ss);
            const float r = rsqrt_nr(ss / 128.0f + eps);
            float xn[8], xp[8], y1[8], y2[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) xn[e] = round_bf16((v[e] * r) * (is_k ? wk[e] : wq[e]));
#pragma unroll
            for (int e = 0; e < 8; ++e) xp[e] = dpp_f32<0x128>(xn[e]);
            if (c < 8)
                rope8<ROUND_REF>(xn, xp, cs, sn, y1, y2);
            else
                rope8<ROUND_REF>(xp, xn, cs, sn, y1, y2);
            r4 = (c < 8) ? pack8(y1) : pack8(y2);
            *reinterpret_cast<uint4*>(src + i * D) = r4;
        }
        if (loc != nullptr && (is_k || is_v)) {
            const int hk = is_k ? hh - Hq : hh - Hq - Hkv;
            *reinterpret_cast<uint4*>((is_k ? kc : vc) + slot + hk * D + c * 8) = r4;
        }
    }
}

// ---- greedy argmax over a vocab row: one 256-thread block per row ----
template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<uint16_t>(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }

// total order used by torch.argmax: NaN is the largest value, ties go to the lower index
__device__ __forceinline__ bool argmax_better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;
    if (vn) return i < bi;
    return v > bv || (v == bv && i < bi);
}

// One 1024-thread block per row; 16 bytes per thread per step when the row is 16-byte aligned.
template <typename T>
__global__ __launch_bounds__(1024) void argmax_kernel(const T* logits, int64_t row_stride, int32_t* out, float* out_val, int vocab,
                                                      int vec_ok) {
    __shared__ float bv[16];
    __shared__ int bi[16];
    constexpr int VEC = 16 / sizeof(T);
    const T* row = logits + (int64_t)blockIdx.x * row_stride;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    int done = 0;
    if (vec_ok) {
        const int nvec = vocab / VEC;
        for (int v = threadIdx.x; v < nvec; v += 1024) {
            const uint4 raw = *reinterpret_cast<const uint4*>(row + (int64_t)v * VEC);
            const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float x;
                if (sizeof(T) == 2) x = (e & 1) ? bf16_hi(w[e >> 1]) : bf16_lo(w[e >> 1]);
                else x = __builtin_bit_cast(float, w[e]);
                const int i = v * VEC + e;
                if (argmax_better(x, i, best, besti)) {
                    best = x;
                    besti = i;
                }
            }
        }
        done = nvec * VEC;
    }
    for (int i = done + threadIdx.x; i < vocab; i += 1024) {
        const float x = to_f32<T>(row[i]);
        if (argmax_better(x, i, best, besti)) {
            best = x;
            besti = i;
        }
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(besti, off);
        if (argmax_better(ov, oi, best, besti)) {
            best = ov;
            besti = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        bv[threadIdx.x >> 6] = best;
        bi[threadIdx.x >> 6] = besti;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (argmax_better(bv[w], bi[w], best, besti)) {
                best = bv[w];
                besti = bi[w];
            }
        out[blockIdx.x] = besti == 0x7fffffff ? 0 : besti;
        if (out_val != nullptr) out_val[blockIdx.x] = best;
    }
}

static inline int grid_for(int64_t items, int per_block) {
    int64_t g = (items + per_block - 1) / per_block;
    return (int)(g < 1 ? 1 : (g > 8192 ? 8192 : g));
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_rmsnorm(const void* x, int64_t x_stride, const void* w, void* out, int64_t out_stride, int rows,
                            int cols, float eps, void* stream) {
    if (rows == 0) return MSGL_OK;
    MSGL_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 16384, "msgl_rmsnorm: rows=%d cols=%d unsupported", rows,
                 cols);
    MSGL_REQUIRE(x && w && out && aligned16(x) && aligned16(w) && aligned16(out), "msgl_rmsnorm: null or unaligned pointer");
    MSGL_REQUIRE(x_stride % 8 == 0 && out_stride % 8 == 0 && x_stride >= cols && out_stride >= cols,
                 "msgl_rmsnorm: bad row stride");
    auto xs = static_cast<const uint16_t*>(x);
    auto ws = static_cast<const uint16_t*>(w);
    auto os = static_cast<uint16_t*>(out);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (cols == 128) {
        hipLaunchKernelGGL(rmsnorm_d128_kernel, dim3((rows + 15) / 16), dim3(256), 0, st, xs, x_stride, ws, os, out_stride,
                           rows, eps);
    } else {
        const int nch = (cols + 2047) / 2048;
#define MSGL_RMS(N) \
    hipLaunchKernelGGL((rmsnorm_wide_kernel<N, false>), dim3(rows), dim3(256), 0, st, xs, x_stride, nullptr, ws, os, \
                       out_stride, cols, eps)
        if (nch <= 1) MSGL_RMS(1);
        else if (nch == 2) MSGL_RMS(2);
        else if (nch == 3) MSGL_RMS(3);
        else if (nch == 4) MSGL_RMS(4);
        else MSGL_RMS(8);
#undef MSGL_RMS
    }
    return check_launch("rmsnorm");
}

extern "C" int msgl_fused_add_rmsnorm(void* x, void* residual, const void* w, int rows, int cols, float eps, void* stream) {
    if (rows == 0) return MSGL_OK;
    MSGL_REQUIRE(rows > 0 && cols > 0 && cols % 8 == 0 && cols <= 16384, "msgl_fused_add_rmsnorm: rows=%d cols=%d unsupported",
                 rows, cols);
    MSGL_REQUIRE(x && residual && w && aligned16(x) && aligned16(residual) && aligned16(w),
                 "msgl_fused_add_rmsnorm: null or unaligned pointer");
    auto xs = static_cast<uint16_t*>(x);
    auto rs = static_cast<uint16_t*>(residual);
    auto ws = static_cast<const uint16_t*>(w);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nch = (cols + 2047) / 2048;
#define MSGL_RMS(N) \
    hipLaunchKernelGGL((rmsnorm_wide_kernel<N, true>), dim3(rows), dim3(256), 0, st, xs, (int64_t)cols, rs, ws, xs, \
                       (int64_t)cols, cols, eps)
    if (nch <= 1) MSGL_RMS(1);
    else if (nch == 2) MSGL_RMS(2);
    else if (nch == 3) MSGL_RMS(3);
    else if (nch == 4) MSGL_RMS(4);
    else MSGL_RMS(8);
#undef MSGL_RMS
    return check_launch("fused_add_rmsnorm");
}

extern "C" int msgl_silu_and_mul(const void* x, int64_t x_stride, void* out, int rows, int n, void* stream) {
    if (rows == 0) return MSGL_OK;
    MSGL_REQUIRE(rows > 0 && n > 0 && n % 8 == 0, "msgl_silu_and_mul: rows=%d n=%d unsupported", rows, n);
    MSGL_REQUIRE(x && out && aligned16(x) && aligned16(out), "msgl_silu_and_mul: null or unaligned pointer");
    MSGL_REQUIRE(x_stride % 8 == 0 && x_stride >= 2 * (int64_t)n, "msgl_silu_and_mul: bad row stride");
    hipLaunchKernelGGL(silu_and_mul_kernel, dim3(grid_for((int64_t)rows * (n / 8), 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint16_t*>(x), x_stride, static_cast<uint16_t*>(out),
                       rows, n);
    return check_launch("silu_and_mul");
}

extern "C" int msgl_rope_neox(const int32_t* positions, void* q, int64_t q_stride, void* k, int64_t k_stride,
                              const float* cos_sin_cache, int T, int Hq, int Hkv, int D, int rounding, void* stream) {
    if (T == 0) return MSGL_OK;
    MSGL_REQUIRE(T > 0 && Hq >= 0 && Hkv >= 0 && Hq + Hkv > 0, "msgl_rope_neox: T=%d Hq=%d Hkv=%d", T, Hq, Hkv);
    MSGL_REQUIRE(D >= 16 && D % 16 == 0 && D <= 512, "msgl_rope_neox: head dim %d unsupported (multiple of 16, <= 512)", D);
    MSGL_REQUIRE(positions && cos_sin_cache && (q || Hq == 0) && (k || Hkv == 0), "msgl_rope_neox: null pointer");
    MSGL_REQUIRE(aligned16(q) && aligned16(k) && aligned16(cos_sin_cache) && q_stride % 8 == 0 && k_stride % 8 == 0,
                 "msgl_rope_neox: q, k, cache must be 16-byte aligned with 16-byte aligned rows");
    const int64_t items = (int64_t)T * (Hq + Hkv) * (D / 16);
    const dim3 grid(grid_for(items, 256)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (rounding == 0)
        hipLaunchKernelGGL((rope_neox_kernel<true>), grid, block, 0, st, positions, static_cast<uint16_t*>(q), q_stride,
                           static_cast<uint16_t*>(k), k_stride, cos_sin_cache, T, Hq, Hkv, D);
    else
        hipLaunchKernelGGL((rope_neox_kernel<false>), grid, block, 0, st, positions, static_cast<uint16_t*>(q), q_stride,
                           static_cast<uint16_t*>(k), k_stride, cos_sin_cache, T, Hq, Hkv, D);
    return check_launch("rope_neox");
}

extern "C" int msgl_store_kv(const void* k, int64_t k_stride, const void* v, int64_t v_stride, const int32_t* loc,
                             void* k_cache, void* v_cache, int64_t cache_stride, int T, int Hkv, int D, void* stream) {
    if (T == 0) return MSGL_OK;
    MSGL_REQUIRE(T > 0 && Hkv > 0 && D > 0 && (Hkv * D) % 8 == 0, "msgl_store_kv: T=%d Hkv=%d D=%d unsupported", T, Hkv, D);
    MSGL_REQUIRE(k && v && loc && k_cache && v_cache, "msgl_store_kv: null pointer");
    MSGL_REQUIRE(aligned16(k) && aligned16(v) && aligned16(k_cache) && aligned16(v_cache) && k_stride % 8 == 0 &&
                     v_stride % 8 == 0 && cache_stride % 8 == 0,
                 "msgl_store_kv: tensors must be 16-byte aligned with 16-byte aligned rows");
    MSGL_REQUIRE(cache_stride >= (int64_t)Hkv * D, "msgl_store_kv: cache stride smaller than a row");
    const int row_chunks = Hkv * D / 8;
    hipLaunchKernelGGL(store_kv_kernel, dim3(grid_for((int64_t)T * row_chunks, 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), static_cast<const uint16_t*>(k), k_stride,
                       static_cast<const uint16_t*>(v), v_stride, loc, static_cast<uint16_t*>(k_cache),
                       static_cast<uint16_t*>(v_cache), cache_stride, T, row_chunks);
    return check_launch("store_kv");
}

extern "C" int msgl_qknorm_rope_store(void* qkv, int64_t qkv_stride, const void* q_norm_w, const void* k_norm_w, float eps,
                                      const int32_t* positions, const float* cos_sin_cache, const int32_t* loc,
                                      void* k_cache, void* v_cache, int64_t cache_stride, int T, int Hq, int Hkv, int D,
                                      int rounding, void* stream) {
    if (T == 0) return MSGL_OK;
    MSGL_REQUIRE(D == 128, "msgl_qknorm_rope_store: head dim %d unsupported (128 only)", D);
    MSGL_REQUIRE(T > 0 && Hq > 0 && Hkv > 0, "msgl_qknorm_rope_store: T=%d Hq=%d Hkv=%d", T, Hq, Hkv);
    MSGL_REQUIRE(qkv && q_norm_w && k_norm_w && positions && cos_sin_cache, "msgl_qknorm_rope_store: null pointer");
    MSGL_REQUIRE(loc == nullptr || (k_cache && v_cache), "msgl_qknorm_rope_store: loc given without caches");
    MSGL_REQUIRE(aligned16(qkv) && aligned16(q_norm_w) && aligned16(k_norm_w) && aligned16(cos_sin_cache) &&
                     aligned16(k_cache) && aligned16(v_cache) && qkv_stride % 8 == 0 && cache_stride % 8 == 0,
                 "msgl_qknorm_rope_store: tensors must be 16-byte aligned with 16-byte aligned rows");
    MSGL_REQUIRE(qkv_stride >= (int64_t)(Hq + 2 * Hkv) * D, "msgl_qknorm_rope_store: qkv stride smaller than a row");
    const int64_t items = (int64_t)T * (Hq + 2 * Hkv);
    const dim3 grid((unsigned)((items + 15) / 16)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto qp = static_cast<uint16_t*>(qkv);
    auto qw = static_cast<const uint16_t*>(q_norm_w);
    auto kw = static_cast<const uint16_t*>(k_norm_w);
    auto kc = static_cast<uint16_t*>(k_cache);
    auto vc = static_cast<uint16_t*>(v_cache);
    constexpr int kHeadsPerRow = 8;
    const int64_t row_items = (int64_t)T * ((Hq + 2 * Hkv + kHeadsPerRow - 1) / kHeadsPerRow);
    if (row_items >= 16 * 2048) {  // enough rows to fill the chip several times over: amortise the per-token loads
        const dim3 rgrid((unsigned)((row_items + 15) / 16));
        if (rounding == 0)
            hipLaunchKernelGGL((qknorm_rope_store_rows_kernel<true, kHeadsPerRow>), rgrid, block, 0, st, qp, qkv_stride, qw, kw, eps,
                               positions, cos_sin_cache, loc, kc, vc, cache_stride, T, Hq, Hkv);
        else
            hipLaunchKernelGGL((qknorm_rope_store_rows_kernel<false, kHeadsPerRow>), rgrid, block, 0, st, qp, qkv_stride, qw, kw, eps,
                               positions, cos_sin_cache, loc, kc, vc, cache_stride, T, Hq, Hkv);
        return check_launch("qknorm_rope_store_rows");
    }
    if (rounding == 0)
        hipLaunchKernelGGL((qknorm_rope_store_kernel<true>), grid, block, 0, st, qp, qkv_stride, qw, kw, eps, positions,
                           cos_sin_cache, loc, kc, vc, cache_stride, T, Hq, Hkv);
    else
        hipLaunchKernelGGL((qknorm_rope_store_kernel<false>), grid, block, 0, st, qp, qkv_stride, qw, kw, eps, positions,
                           cos_sin_cache, loc, kc, vc, cache_stride, T, Hq, Hkv);
    return check_launch("qknorm_rope_store");
}

extern "C" int msgl_argmax_val(const void* logits, int64_t row_stride, int elem_bytes, int32_t* out, float* out_val, int bs,
                               int vocab, void* stream) {
    if (bs == 0) return MSGL_OK;
    MSGL_REQUIRE(bs > 0 && vocab > 0 && logits && out, "msgl_argmax: bs=%d vocab=%d or null pointer", bs, vocab);
    MSGL_REQUIRE(elem_bytes == 2 || elem_bytes == 4, "msgl_argmax: elem_bytes=%d (2 = bf16, 4 = fp32)", elem_bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int vec_ok = aligned16(logits) && (row_stride * elem_bytes) % 16 == 0;
    if (elem_bytes == 2)
        hipLaunchKernelGGL((argmax_kernel<uint16_t>), dim3(bs), dim3(1024), 0, st, static_cast<const uint16_t*>(logits),
                           row_stride, out, out_val, vocab, vec_ok);
    else
        hipLaunchKernelGGL((argmax_kernel<float>), dim3(bs), dim3(1024), 0, st, static_cast<const float*>(logits), row_stride,
                           out, out_val, vocab, vec_ok);
    return check_launch("argmax");
}

extern "C" int msgl_argmax(const void* logits, int64_t row_stride, int elem_bytes, int32_t* out, int bs, int vocab,
                           void* stream) {
    return msgl_argmax_val(logits, row_stride, elem_bytes, out, nullptr, bs, vocab, stream);
}
