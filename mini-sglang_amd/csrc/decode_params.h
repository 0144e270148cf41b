// Launch parameters and split arithmetic of the decode path, shared by the kernels that write split
// partials (the per-request gather and the shared-prefix kernel) and the ones that plan and merge them.
#pragma once
#include "msgl_common.h"

namespace msgl {

constexpr int kTile = 16;    // tokens per wave iteration
constexpr int kWaves = 8;    // waves per work group
constexpr int kD = 128;      // head dim
constexpr float kNegBig = -1.0e30f;

struct DecodeParams {
    const uint16_t* q;
    int64_t q_stride;
    const uint16_t* kc;
    const uint16_t* vc;
    int64_t cache_stride;
    const int32_t* r2t;
    int64_t r2t_stride;
    const int32_t* req_idx;
    const int32_t* seq_lens;
    uint16_t* out;
    int64_t out_stride;
    float* ws_o;   // [bs, max_splits, Hq, 128] unnormalised partial outputs
    float* ws_ml;  // [bs, max_splits, Hq, 2]   (running max in log2 units, running sum)
    const int32_t* chunk_dev;
    const int32_t* reuse_lens;  // per request: positions below this may be shared with other requests
    const int32_t* plan;        // optional launch order of the (split, request) items, see decode_plan_kernel
    const int32_t* start_lens;  // optional, per request: leading positions another kernel attends over (shared prefix)
    const int32_t* extra_splits;  // optional, per request: partial-result slots that kernel fills, after this one's
    int kv_stream;              // 0: every K/V load is cached; 1: loads at or above reuse_lens are non-temporal
    int chunk;
    int bs, Hq, max_splits;
    float scale_log2e;
};

// Tokens per split: a positive multiple of the tile so that tiles never straddle a page.
__device__ __forceinline__ int split_chunk(const DecodeParams& p) {
    const int c = p.chunk_dev ? *p.chunk_dev : p.chunk;
    return max(kTile, (c + kTile - 1) & ~(kTile - 1));
}

// What request b asks of the gather kernel: positions [start, S) in `nsplit` splits of `chunk` tokens (the last of
// max_splits takes whatever is left), and `extra` more partial results in slots nsplit .. nsplit + extra - 1.
struct SplitInfo {
    int start, S, nsplit, extra;
};
__device__ __forceinline__ SplitInfo split_info(const DecodeParams& p, int b, int chunk) {
    SplitInfo r;
    r.S = p.seq_lens[b];
    r.start = p.start_lens ? p.start_lens[b] : 0;
    r.extra = p.extra_splits ? p.extra_splits[b] : 0;
    r.nsplit = max(0, min((r.S - r.start + chunk - 1) / chunk, p.max_splits));
    return r;
}

// End of a (request, split) work group of either decode kernel: every wave has left its state for its kv head in LDS
// (lds_o [wave][G][128] unnormalised outputs, lds_ml [wave][G][2] running max in log2 units and running sum) and the
// group has passed a barrier.  The WPH waves of kv head h (sub = 0 .. WPH - 1, all of them call this) share the merge,
// four output columns per lane and pass, and write either the final bf16 output (the request has this one split and no
// partials from another kernel) or the split's partial.
template <int G, int WPH>
__device__ __forceinline__ void merge_waves_and_store(const DecodeParams& p, const float* lds_o, const float* lds_ml, int h, int sub, int lane,
                                                      int b, int split, bool final_out) {
#pragma unroll
    for (int idx = (sub * 64 + lane) * 4; idx < G * kD; idx += WPH * 256) {
        const int g = idx / kD, d = idx - g * kD;
        float mw[WPH], lw[WPH];
        float4 ow[WPH];
#pragma unroll
        for (int w = 0; w < WPH; ++w) {  // every read issued before the first is used
            const int ww = h * WPH + w;
            const float2 ml = *reinterpret_cast<const float2*>(lds_ml + (ww * G + g) * 2);
            mw[w] = ml.x, lw[w] = ml.y;
            ow[w] = *reinterpret_cast<const float4*>(lds_o + (ww * G + g) * kD + d);
        }
        float M = kNegBig;
#pragma unroll
        for (int w = 0; w < WPH; ++w) M = fmaxf(M, mw[w]);
        float L = 0.f;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < WPH; ++w) {
            const float f = fast_exp2(mw[w] - M);
            L += f * lw[w];
            o.x += f * ow[w].x, o.y += f * ow[w].y, o.z += f * ow[w].z, o.w += f * ow[w].w;
        }
        const int hq = h * G + g;
        if (final_out) {
            const float inv = 1.0f / L;
            *reinterpret_cast<uint2*>(p.out + (int64_t)b * p.out_stride + hq * kD + d) =
                make_uint2(pack_bf16(o.x * inv, o.y * inv), pack_bf16(o.z * inv, o.w * inv));
        } else {
            const int64_t row = ((int64_t)b * p.max_splits + split) * p.Hq + hq;
            *reinterpret_cast<float4*>(p.ws_o + row * kD + d) = o;
            if (d == 0) *reinterpret_cast<float2*>(p.ws_ml + row * 2) = make_float2(M, L);
        }
    }
}

}  // namespace msgl
