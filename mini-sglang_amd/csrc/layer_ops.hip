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

// 16-byte global accesses, with the non-temporal hint when the tensor is streamed once and far larger than the caches
typedef __attribute__((ext_vector_type(4))) unsigned int lu32x4_t;
template <bool STREAM>
__device__ __forceinline__ uint4 load16(const uint16_t* p) {
    if (!STREAM) return *reinterpret_cast<const uint4*>(p);
    const lu32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const lu32x4_t*>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <bool STREAM>
__device__ __forceinline__ void store16(uint16_t* p, const uint4& v) {
    if (!STREAM) {
        *reinterpret_cast<uint4*>(p) = v;
    } else {
        const lu32x4_t w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<lu32x4_t*>(p));
    }
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

// ---- RMSNorm, many rows: one WAVE per row, NCH 16-byte chunks per lane (cols = NCH * 512) ----
// No barrier and no LDS: the row sum is a wave reduction, every load of the row (and the weights) is in flight before
// the first is consumed (NCH * 16 bytes per lane, twice that with the residual), and the four waves of a work group
// run their rows independently.  STREAM: tensors far larger than the caches go by with the non-temporal hint.
template <int NCH, bool ADD, bool STREAM>
__global__ __launch_bounds__(256) void rmsnorm_rows_kernel(const uint16_t* x, int64_t x_stride, uint16_t* res,
                                                           const uint16_t* w, uint16_t* out, int64_t out_stride,
                                                           int rows, float eps) {
    constexpr int cols = NCH * 512;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const uint16_t* xr = x + (int64_t)row * x_stride + lane * 8;
    uint16_t* rr = ADD ? res + (int64_t)row * cols + lane * 8 : nullptr;
    uint4 rx[NCH], rres[ADD ? NCH : 1], rw[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) rx[k] = load16<STREAM>(xr + k * 512);
    if (ADD) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) rres[k] = load16<STREAM>(rr + k * 512);
    }
#pragma unroll
    for (int k = 0; k < NCH; ++k) rw[k] = *reinterpret_cast<const uint4*>(w + lane * 8 + k * 512);
    float v[NCH][8];
    float ss = 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        unpack8(rx[k], v[k]);
        if (ADD) {
            float r[8];
            unpack8(rres[k], r);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[k][e] += r[e];
            store16<STREAM>(rr + k * 512, pack8(v[k]));  // residual' = bf16(sum); the norm keeps the fp32 sum
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(v[k][e], v[k][e], ss);
    }
    ss = wave_sum(ss);
    const float r = rsqrt_nr(ss / (float)cols + eps);
    uint16_t* orow = out + (int64_t)row * out_stride + lane * 8;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        float wf[8], o[8];
        unpack8(rw[k], wf);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[k][e] * r) * wf[e];
        store16<STREAM>(orow + k * 512, pack8(o));
    }
}

// ---- RMSNorm, 128-wide rows (per-head q/k norm): 16 lanes per row, RPT rows per lane group ----
template <int RPT, bool STREAM>
__global__ __launch_bounds__(256) void rmsnorm_d128_kernel(const uint16_t* x, int64_t x_stride, const uint16_t* w,
                                                           uint16_t* out, int64_t out_stride, int rows, float eps) {
    // the 16 lane groups of a work group take rows r0 + g, r0 + 16 + g, ...: consecutive groups read consecutive rows
    const int row0 = blockIdx.x * (16 * RPT) + (threadIdx.x >> 4);
    const int c = threadIdx.x & 15;
    uint4 raw[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = row0 + 16 * i;
        raw[i] = row < rows ? load16<STREAM>(x + (int64_t)row * x_stride + c * 8) : make_uint4(0, 0, 0, 0);
    }
    float wf[8];
    unpack8(*reinterpret_cast<const uint4*>(w + c * 8), wf);
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const int row = row0 + 16 * i;
        float v[8], o[8];
        unpack8(raw[i], v);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
        ss = row16_sum(ss);
        const float r = rsqrt_nr(ss / 128.0f + eps);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (v[e] * r) * wf[e];
        if (row < rows) store16<STREAM>(out + (int64_t)row * out_stride + c * 8, pack8(o));
    }
}

// ---- SiluAndMul ----
// x / (1 + exp(-x)) on the hardware's exp2 and reciprocal estimates, each corrected to about an ulp of fp32: the bf16
// rounding of this value feeds a second rounded product, where one bf16 ulp of difference here can become two there,
// so the value has to agree with a correctly rounded evaluation (the reference's expf and division) to within the
// last fp32 bits, not merely to bf16 precision.
//   exp(-x) = exp2(t) * (1 + ln2 * e),  t = fl(-x log2e),  e = the part of -x log2e that t's rounding dropped
//   quotient: one Newton step on x * rcp(d)
__device__ __forceinline__ float silu_f32(float x) {
    constexpr float kL2eHi = -1.4426950216293335f;  // -(log2 e) rounded to fp32, and what that rounding dropped
    constexpr float kL2eLo = -1.92596298909109e-8f;
    const float t = x * kL2eHi;
    float e = __builtin_fmaf(x, kL2eHi, -t);
    e = __builtin_fmaf(x, kL2eLo, e);
    float p = fast_exp2(t);
    p = __builtin_fmaf(p, e * 0.6931471805599453f, p);
    const float d = 1.0f + p;
    const float r = fast_rcp(d);
    const float q = x * r;
    // (d = inf: r = 0, q = -0 and the correction term is NaN-free only if skipped; exp2 overflows beyond x < -88)
    const float c = __builtin_fmaf(-d, q, x);
    return p > 3.0e38f ? q : __builtin_fmaf(c, r, q);
}

// One thread = RPT rows x one 16-byte column chunk: all 2 * RPT loads are issued before the first result is formed
// (64 bytes in flight per lane at RPT = 2), and no index arithmetic beyond a multiply.  grid = (column blocks, row
// groups; the row groups stride over the grid when there are more than it holds).  STREAM: tensors far larger than the
// caches (prefill) are read and written with the non-temporal hint; decode-size tensors stay cacheable for the GEMM
// that reads the result next.
template <int RPT, bool STREAM>
__global__ __launch_bounds__(256) void silu_and_mul_kernel(const uint16_t* x, int64_t x_stride, uint16_t* out, int rows,
                                                           int n) {
    const int col = (blockIdx.x * 256 + threadIdx.x) * 8;
    if (col >= n) return;
    for (int row0 = blockIdx.y * RPT; row0 < rows; row0 += gridDim.y * RPT) {
        uint4 ra[RPT], rb[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const uint16_t* xr = x + (int64_t)min(row0 + r, rows - 1) * x_stride + col;
            ra[r] = load16<STREAM>(xr);
            rb[r] = load16<STREAM>(xr + n);
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            if (row0 + r >= rows) break;
            float a[8], b[8], o[8];
            unpack8(ra[r], a);
            unpack8(rb[r], b);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = round_bf16(silu_f32(a[e])) * b[e];
            store16<STREAM>(out + (int64_t)(row0 + r) * n + col, pack8(o));
        }
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

// cos / sin entries of a position for 8 pairs, rounded to bf16 first when the reference's torch fallback is reproduced
template <bool ROUND_REF>
__device__ __forceinline__ void load_cos_sin8(const float* cs_lo, const float* sn_lo, float (&cb)[8], float (&sb)[8]) {
    const float4 c0 = *reinterpret_cast<const float4*>(cs_lo), c1 = *reinterpret_cast<const float4*>(cs_lo + 4);
    const float4 s0 = *reinterpret_cast<const float4*>(sn_lo), s1 = *reinterpret_cast<const float4*>(sn_lo + 4);
    const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        cb[e] = ROUND_REF ? round_bf16(cs[e]) : cs[e];
        sb[e] = ROUND_REF ? round_bf16(sn[e]) : sn[e];
    }
}
// Two values after one rounding to bf16 each, through one packed conversion
__device__ __forceinline__ void round_bf16_pair(float a, float b, float& ra, float& rb) {
    const uint32_t p = pack_bf16(a, b);
    ra = bf16_lo(p);
    rb = bf16_hi(p);
}
// One half of the rotation: y = own * cos + partner * ssin, with ssin = -sin for the first half of a head (y1 = x1 cos
// - x2 sin) and +sin for the second (y2 = x2 cos + x1 sin).  Negating an operand commutes with every rounding, so this
// gives the bits of the two-sided form while a lane that keeps one half computes only that half.
template <bool ROUND_REF>
__device__ __forceinline__ void rope_half8(const float (&own)[8], const float (&partner)[8], const float (&cb)[8],
                                           const float (&ssin)[8], float (&y)[8]) {
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        if (ROUND_REF) {
            float a0, a1, b0, b1;
            round_bf16_pair(own[e] * cb[e], own[e + 1] * cb[e + 1], a0, a1);
            round_bf16_pair(partner[e] * ssin[e], partner[e + 1] * ssin[e + 1], b0, b1);
            y[e] = a0 + b0;
            y[e + 1] = a1 + b1;
        } else {
            y[e] = own[e] * cb[e] + partner[e] * ssin[e];
            y[e + 1] = own[e + 1] * cb[e + 1] + partner[e + 1] * ssin[e + 1];
        }
    }
}

// neox RoPE over many tokens (D = 128): 8 lanes per (token, run of HPT heads).  The position's cos / sin entries are read
// and rounded once per run instead of once per head, and the run's 2 * HPT loads are in flight together.
template <bool ROUND_REF, int HPT, bool STREAM>
__global__ __launch_bounds__(256) void rope_neox_rows_kernel(const int32_t* positions, uint16_t* q, int64_t q_stride, uint16_t* k,
                                                             int64_t k_stride, const float* cache, int T, int Hq, int Hkv) {
    constexpr int D = 128;
    const int H = Hq + Hkv;
    const int runs = (H + HPT - 1) / HPT;
    const int item = blockIdx.x * 32 + (threadIdx.x >> 3);
    const int c = threadIdx.x & 7;
    if (item >= T * runs) return;
    const int t = item / runs;
    const int h0 = (item - t * runs) * HPT;
    uint16_t* base[HPT];
    uint4 r1[HPT], r2[HPT];
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        const int hh = min(h0 + i, H - 1);
        base[i] = ((hh < Hq) ? q + (int64_t)t * q_stride + hh * D : k + (int64_t)t * k_stride + (hh - Hq) * D) + c * 8;
        r1[i] = load16<STREAM>(base[i]);
        r2[i] = load16<STREAM>(base[i] + 64);
    }
    float cb[8], sb[8], nsb[8];
    const float* cp = cache + (int64_t)positions[t] * D + c * 8;
    load_cos_sin8<ROUND_REF>(cp, cp + 64, cb, sb);
#pragma unroll
    for (int e = 0; e < 8; ++e) nsb[e] = -sb[e];
#pragma unroll
    for (int i = 0; i < HPT; ++i) {
        if (h0 + i >= H) break;
        float x1[8], x2[8], y1[8], y2[8];
        unpack8(r1[i], x1);
        unpack8(r2[i], x2);
        rope_half8<ROUND_REF>(x1, x2, cb, nsb, y1);
        rope_half8<ROUND_REF>(x2, x1, cb, sb, y2);
        store16<STREAM>(base[i], pack8(y1));
        store16<STREAM>(base[i] + 64, pack8(y2));
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

// The same for many tokens: a thread owns one 16-byte column chunk of TPT consecutive tokens' K and V rows, so that
// 2 * TPT loads (and the TPT slot indices) are in flight before the first store; grid = (column blocks, token groups).
template <int TPT>
__global__ __launch_bounds__(256) void store_kv_rows_kernel(const uint16_t* k, int64_t k_stride, const uint16_t* v, int64_t v_stride,
                                                            const int32_t* loc, uint16_t* kc, uint16_t* vc, int64_t cache_stride,
                                                            int T, int row_chunks) {
    const int chunk = blockIdx.x * 64 + (threadIdx.x & 63);
    if (chunk >= row_chunks) return;
    const int col = chunk * 8;
    for (int t0 = (blockIdx.y * 4 + (threadIdx.x >> 6)) * TPT; t0 < T; t0 += gridDim.y * 4 * TPT) {
        uint4 rk[TPT], rv[TPT];
        int32_t slot[TPT];
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
            const int t = min(t0 + i, T - 1);
            slot[i] = loc[t];
            rk[i] = load16<true>(k + (int64_t)t * k_stride + col);
            rv[i] = load16<true>(v + (int64_t)t * v_stride + col);
        }
#pragma unroll
        for (int i = 0; i < TPT; ++i) {
            if (t0 + i >= T) break;
            const int64_t dst = (int64_t)slot[i] * cache_stride + col;
            *reinterpret_cast<uint4*>(kc + dst) = rk[i];
            *reinterpret_cast<uint4*>(vc + dst) = rv[i];
        }
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
        float cb[8], ssin[8], y[8];
        load_cos_sin8<ROUND_REF>(cs, cs + 64, cb, ssin);
        if (c < 8) {  // first half of the head: y1 = x1 cos - x2 sin; second half: y2 = x2 cos + x1 sin
#pragma unroll
            for (int e = 0; e < 8; ++e) ssin[e] = -ssin[e];
        }
        rope_half8<ROUND_REF>(xn, xp, cb, ssin, y);
        raw = pack8(y);
        if (live) *reinterpret_cast<uint4*>(src) = raw;
    }
    if (live && loc != nullptr && (is_k || is_v)) {
        const int hk = is_k ? hh - Hq : hh - Hq - Hkv;
        uint16_t* dst = (is_k ? kc : vc) + (int64_t)loc[t] * cache_stride + hk * D + c * 8;
        *reinterpret_cast<uint4*>(dst) = raw;
    }
}

// The same work for large T: a 16-lane row owns a run of up to HPR consecutive heads of one token, all of one kind
// (query heads, key heads or value heads: one weight vector per run, and value runs are a plain copy).  The position's cos / sin entries and the norm weight are read once
// per run instead of once per head, and the run's loads are all in flight before the first head is processed.
template <bool ROUND_REF, int HPR, bool STREAM>
__global__ __launch_bounds__(256) void qknorm_rope_store_rows_kernel(uint16_t* qkv, int64_t qkv_stride, const uint16_t* qw,
                                                                     const uint16_t* kw, float eps, const int32_t* positions,
                                                                     const float* cache, const int32_t* loc, uint16_t* kc,
                                                                     uint16_t* vc, int64_t cache_stride, int T, int Hq,
                                                                     int Hkv) {
    constexpr int D = 128;
    const int q_runs = (Hq + HPR - 1) / HPR, kv_runs = (Hkv + HPR - 1) / HPR;
    // items in token-major order (all runs of a token are neighbours: a work group covers a contiguous stretch of qkv
    // rows; run-major order measured 15 % slower)
    const int runs = q_runs + 2 * kv_runs;
    const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int t = (int)(item / runs);
    const int run = (int)(item - (int64_t)t * runs);
    const int kind = run < q_runs ? 0 : (run < q_runs + kv_runs ? 1 : 2);  // query, key, value heads
    const int first = (kind == 0 ? run : kind == 1 ? run - q_runs : run - q_runs - kv_runs) * HPR;  // within its kind
    const int n = min(HPR, (kind == 0 ? Hq : Hkv) - first);
    const int c = threadIdx.x & 15;
    if (t >= T) return;  // whole 16-lane rows leave together
    uint16_t* src = qkv + (int64_t)t * qkv_stride + ((kind == 0 ? 0 : kind == 1 ? Hq : Hq + Hkv) + first) * D + c * 8;
    uint4 raw[HPR];
#pragma unroll
    for (int i = 0; i < HPR; ++i) raw[i] = (i < n) ? load16<STREAM>(src + i * D) : make_uint4(0, 0, 0, 0);
    uint16_t* pool = nullptr;
    if (loc != nullptr && kind != 0) pool = (kind == 1 ? kc : vc) + (int64_t)loc[t] * cache_stride + first * D + c * 8;
    if (kind == 2) {  // value heads: a copy into the pool
        if (pool != nullptr) {
#pragma unroll
            for (int i = 0; i < HPR; ++i)
                if (i < n) store16<false>(pool + i * D, raw[i]);
        }
        return;
    }
    float w[8], cb[8], ssin[8];
    unpack8(*reinterpret_cast<const uint4*>((kind == 1 ? kw : qw) + c * 8), w);
    {
        const float* cp = cache + (int64_t)positions[t] * D + (c & 7) * 8;
        load_cos_sin8<ROUND_REF>(cp, cp + 64, cb, ssin);
        if (c < 8) {  // this lane keeps the first half of every head: y1 = x1 cos - x2 sin
#pragma unroll
            for (int e = 0; e < 8; ++e) ssin[e] = -ssin[e];
        }
    }
#pragma unroll
    for (int i = 0; i < HPR; ++i) {
        if (i >= n) break;
        float v[8];
        unpack8(raw[i], v);
        float ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = fmaf(v[e], v[e], ss);
        ss = row16_sum(ss);
        const float r = rsqrt_nr(ss / 128.0f + eps);
        float xn[8], xp[8], y[8];
#pragma unroll
        for (int e = 0; e < 8; e += 2) round_bf16_pair((v[e] * r) * w[e], (v[e + 1] * r) * w[e + 1], xn[e], xn[e + 1]);
#pragma unroll
        for (int e = 0; e < 8; ++e) xp[e] = dpp_f32<0x128>(xn[e]);
        rope_half8<ROUND_REF>(xn, xp, cb, ssin, y);
        const uint4 r4 = pack8(y);
        store16<STREAM>(src + i * D, r4);
        if (pool != nullptr) store16<false>(pool + i * D, r4);
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

// Tensors of at least this many bytes (inputs + outputs of one launch) are streamed with the non-temporal hint.
#ifndef LAYER_STREAM_MB
#define LAYER_STREAM_MB 192
#endif
constexpr int64_t kStreamBytes = (int64_t)LAYER_STREAM_MB << 20;
// From this many rows on, RMSNorm over wide rows runs a wave per row (fewer rows: a work group per row, which spreads
// a small launch over more compute units).
constexpr int kRowsPerWaveMin = 4096;

// Wave-per-row RMSNorm for the row widths it is built for (multiples of 512 up to 8192 that occur in the Qwen3 family);
// false = no build for this width, the caller takes the work-group-per-row kernel.
template <bool ADD>
static bool launch_rmsnorm_rows(const uint16_t* x, int64_t x_stride, uint16_t* res, const uint16_t* w, uint16_t* out, int64_t out_stride,
                                int rows, int cols, float eps, bool stream_once, hipStream_t st) {
    const dim3 grid((rows + 3) / 4), block(256);
#define MSGL_ROWS(N)                                                                                                               \
    case N * 512:                                                                                                                  \
        if (stream_once)                                                                                                           \
            hipLaunchKernelGGL((rmsnorm_rows_kernel<N, ADD, true>), grid, block, 0, st, x, x_stride, res, w, out, out_stride, rows, eps);  \
        else                                                                                                                       \
            hipLaunchKernelGGL((rmsnorm_rows_kernel<N, ADD, false>), grid, block, 0, st, x, x_stride, res, w, out, out_stride, rows, eps); \
        return true;
    switch (cols) {
        MSGL_ROWS(2)
        MSGL_ROWS(4)
        MSGL_ROWS(5)
        MSGL_ROWS(8)
        MSGL_ROWS(10)
        MSGL_ROWS(16)
    }
#undef MSGL_ROWS
    return false;
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
    const bool stream_once = (int64_t)rows * cols * 4 >= kStreamBytes;  // in + out far beyond the caches
    if (cols == 128) {
        if (rows >= 16 * 4 * 2048) {  // enough rows to fill the chip with four rows per lane group
            if (stream_once)
                hipLaunchKernelGGL((rmsnorm_d128_kernel<4, true>), dim3((rows + 63) / 64), dim3(256), 0, st, xs, x_stride, ws, os, out_stride, rows, eps);
            else
                hipLaunchKernelGGL((rmsnorm_d128_kernel<4, false>), dim3((rows + 63) / 64), dim3(256), 0, st, xs, x_stride, ws, os, out_stride, rows, eps);
        } else {
            hipLaunchKernelGGL((rmsnorm_d128_kernel<1, false>), dim3((rows + 15) / 16), dim3(256), 0, st, xs, x_stride, ws, os, out_stride, rows, eps);
        }
    } else if (rows >= kRowsPerWaveMin && launch_rmsnorm_rows<false>(xs, x_stride, nullptr, ws, os, out_stride, rows, cols, eps, stream_once, st)) {
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
    if (rows >= kRowsPerWaveMin &&
        launch_rmsnorm_rows<true>(xs, cols, rs, ws, xs, cols, rows, cols, eps, (int64_t)rows * cols * 8 >= kStreamBytes, st))
        return check_launch("fused_add_rmsnorm");
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
    const unsigned col_blocks = (unsigned)((n / 8 + 255) / 256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto xs = static_cast<const uint16_t*>(x);
    auto os = static_cast<uint16_t*>(out);
    // two rows per thread once there are enough of them to keep every compute unit's wave slots filled all the same
    if ((int64_t)rows * col_blocks >= 16384) {
        const unsigned gy = (unsigned)(((rows + 1) / 2) > 65535 ? 65535 : (rows + 1) / 2);
        if ((int64_t)rows * n * 6 >= kStreamBytes)
            hipLaunchKernelGGL((silu_and_mul_kernel<2, true>), dim3(col_blocks, gy), dim3(256), 0, st, xs, x_stride, os, rows, n);
        else
            hipLaunchKernelGGL((silu_and_mul_kernel<2, false>), dim3(col_blocks, gy), dim3(256), 0, st, xs, x_stride, os, rows, n);
    } else {
        hipLaunchKernelGGL((silu_and_mul_kernel<1, false>), dim3(col_blocks, (unsigned)(rows > 65535 ? 65535 : rows)), dim3(256), 0, st,
                           xs, x_stride, os, rows, n);
    }
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
#ifndef ROPE_HPT
#define ROPE_HPT 2
#endif
    constexpr int kHeadsPerRun = ROPE_HPT;
    const int64_t run_items = (int64_t)T * ((Hq + Hkv + kHeadsPerRun - 1) / kHeadsPerRun);
    if (D == 128 && run_items >= 32 * 2048 && run_items < ((int64_t)1 << 31) - 64) {  // many tokens: runs of heads per lane group
        const dim3 rgrid((unsigned)((run_items + 31) / 32));
        const bool stream_once = (int64_t)T * (Hq + Hkv) * D * 4 >= kStreamBytes;
        auto qp = static_cast<uint16_t*>(q);
        auto kp = static_cast<uint16_t*>(k);
#define MSGL_ROPE(R, S) \
    hipLaunchKernelGGL((rope_neox_rows_kernel<R, kHeadsPerRun, S>), rgrid, block, 0, st, positions, qp, q_stride, kp, k_stride, cos_sin_cache, T, Hq, Hkv)
        if (rounding == 0) {
            if (stream_once) MSGL_ROPE(true, true); else MSGL_ROPE(true, false);
        } else {
            if (stream_once) MSGL_ROPE(false, true); else MSGL_ROPE(false, false);
        }
#undef MSGL_ROPE
        return check_launch("rope_neox_rows");
    }
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
    if ((int64_t)T * row_chunks >= 256 * 2048) {  // prefill-size stores
        const int64_t token_groups = ((int64_t)T + 15) / 16;
        const dim3 grid((row_chunks + 63) / 64, (unsigned)(token_groups > 65535 ? 65535 : token_groups));
        // (the source rows are read once: non-temporal loads; the pool rows are read next by this layer's attention, and
        // scattered non-temporal stores measured slower: plain stores)
        hipLaunchKernelGGL((store_kv_rows_kernel<4>), grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                           static_cast<const uint16_t*>(k), k_stride, static_cast<const uint16_t*>(v), v_stride, loc,
                           static_cast<uint16_t*>(k_cache), static_cast<uint16_t*>(v_cache), cache_stride, T, row_chunks);
        return check_launch("store_kv");
    }
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
#ifndef QKN_HPR
#define QKN_HPR 4
#endif
    constexpr int kHeadsPerRow = QKN_HPR;
    const int runs = (Hq + kHeadsPerRow - 1) / kHeadsPerRow + 2 * ((Hkv + kHeadsPerRow - 1) / kHeadsPerRow);
    const int64_t row_items = (int64_t)T * runs;
    if (row_items >= 16 * 2048) {  // enough rows to fill the chip several times over: amortise the per-token loads
        const dim3 rgrid((unsigned)((row_items + 15) / 16));
        const bool stream_once = (int64_t)T * (Hq + 2 * Hkv) * D * 4 >= kStreamBytes;
#define MSGL_QKN(R, S)                                                                                                          \
    hipLaunchKernelGGL((qknorm_rope_store_rows_kernel<R, kHeadsPerRow, S>), rgrid, block, 0, st, qp, qkv_stride, qw, kw, eps, \
                       positions, cos_sin_cache, loc, kc, vc, cache_stride, T, Hq, Hkv)
        if (rounding == 0) {
            if (stream_once) MSGL_QKN(true, true); else MSGL_QKN(true, false);
        } else {
            if (stream_once) MSGL_QKN(false, true); else MSGL_QKN(false, false);
        }
#undef MSGL_QKN
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
