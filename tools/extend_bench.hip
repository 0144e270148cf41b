// Standalone driver for msgl_paged_extend: bs equal-length prompts over a scattered page table, random
// bf16 pool, HIP-event timing.  Used for profiler passes (rocprofv3 --pmc) without a Python process.
//
// Built with -DEXT_STAMPS against a library object built the same way it also prints the cycles the steady-state
// loop of the prefill kernel spends per segment (see scripts/stamp_extend.sh).
//
//   extend_bench [--bs 2] [--sq 8192] [--prefix 0] [--ps 16] [--hq 32] [--hkv 8] [--steps 10] [--warmup 2] [--heads 1|2] [--wide 0|1] [--ring 2|3] [--ksplit 1..8]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../include/msgl_hip.h"

#ifdef EXT_STAMPS
extern "C" void msgl_extend_stamps(unsigned long long* out, int reset);
extern "C" void msgl_extend_wide_stamps(unsigned long long* out);
#endif
#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)

__global__ void fill_bf16(uint16_t* p, size_t n, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint64_t z = i + seed * 0x9E3779B97F4A7C15ull;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        const int s = (int)(z & 0xFFFF) + (int)((z >> 16) & 0xFFFF) + (int)((z >> 32) & 0xFFFF) + (int)((z >> 48) & 0xFFFF);
        const float x = (float)(s - 131070) / 37837.0f;
        p[i] = (uint16_t)(__builtin_bit_cast(uint32_t, x) >> 16);
    }
}

int main(int argc, char** argv) {
    int bs = 2, sq = 8192, prefix = 0, ps = 16, hq = 32, hkv = 8, steps = 10, warmup = 2;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        const int v = atoi(argv[i + 1]);
        if (a == "--bs") bs = v;
        else if (a == "--sq") sq = v;
        else if (a == "--prefix") prefix = v;
        else if (a == "--ps") ps = v;
        else if (a == "--hq") hq = v;
        else if (a == "--hkv") hkv = v;
        else if (a == "--steps") steps = v;
        else if (a == "--warmup") warmup = v;
        // kernel-form overrides (msgl_set_variant): what the library would otherwise choose from the shape
        else if (a == "--heads") msgl_set_variant(MSGL_VARIANT_EXTEND_HEADS, v);
        else if (a == "--wide") msgl_set_variant(MSGL_VARIANT_EXTEND_WIDE, v);
        else if (a == "--ring") msgl_set_variant(MSGL_VARIANT_EXTEND_RING, v);
        else if (a == "--ksplit") msgl_set_variant(MSGL_VARIANT_EXTEND_KEY_SPLIT, v);
        else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
    }
    const int D = 128;
    const int ctx = prefix + sq;
    const int pages_per_req = (ctx + ps - 1) / ps;
    const size_t pool_pages = (size_t)bs * pages_per_req + 8;
    const size_t slots = (pool_pages + 1) * ps;
    const int64_t row = (int64_t)hkv * D;
    const int64_t cache_stride = 2 * row;  // interleaved [slots, 2, Hkv, D] pool

    std::vector<int32_t> order(pool_pages);
    std::iota(order.begin(), order.end(), 1);
    std::mt19937_64 rng(3);
    std::shuffle(order.begin(), order.end(), rng);
    const int64_t max_ctx = (int64_t)pages_per_req * ps;
    std::vector<int32_t> r2t((size_t)bs * max_ctx, 0);
    size_t cur = 0;
    for (int b = 0; b < bs; ++b)
        for (int pg = 0; pg < pages_per_req; ++pg) {
            const int32_t page = order[cur++];
            for (int o = 0; o < ps; ++o) r2t[(size_t)b * max_ctx + (size_t)pg * ps + o] = page * ps + o;
        }
    std::vector<int32_t> cu(bs + 1), pre(bs, prefix), idx(bs);
    for (int b = 0; b <= bs; ++b) cu[b] = b * sq;
    std::iota(idx.begin(), idx.end(), 0);

    int32_t *d_r2t, *d_cu, *d_pre, *d_idx;
    CK(hipMalloc(&d_r2t, r2t.size() * 4));
    CK(hipMalloc(&d_cu, (bs + 1) * 4));
    CK(hipMalloc(&d_pre, bs * 4));
    CK(hipMalloc(&d_idx, bs * 4));
    CK(hipMemcpy(d_r2t, r2t.data(), r2t.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_cu, cu.data(), (bs + 1) * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_pre, pre.data(), bs * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx, idx.data(), bs * 4, hipMemcpyHostToDevice));

    uint16_t *kv, *q, *out;
    const size_t kv_elems = slots * row * 2;
    const size_t q_elems = (size_t)bs * sq * hq * D;
    CK(hipMalloc(&kv, kv_elems * 2));
    CK(hipMalloc(&q, q_elems * 2));
    CK(hipMalloc(&out, q_elems * 2));
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, kv, kv_elems, (uint64_t)1);
    hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, q, q_elems, (uint64_t)2);
    CK(hipDeviceSynchronize());
    // key-split workspace, as a caller hands it over when the launch has fewer work items than compute units
    void* d_ws = nullptr;
    size_t ws_bytes = 0;
    if ((int64_t)bs * ((sq + 127) / 128) * std::max(1, hq / 2) < 256 && prefix + sq >= 2048) {
        ws_bytes = msgl_paged_extend_workspace_bytes((int64_t)bs * sq, hq, D);
        CK(hipMalloc(&d_ws, ws_bytes));
    }

    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto step = [&]() {
        const int rc = msgl_paged_extend(q, (int64_t)hq * D, kv, kv + row, cache_stride, d_r2t, max_ctx, d_idx, d_cu, d_pre, out,
                                         (int64_t)hq * D, bs, sq, /*max_seqlen_k=*/prefix + sq, /*plan=*/nullptr, hq, hkv, D, ps,
                                         0.08838834764831845f, d_ws, ws_bytes, (int64_t)bs * sq, st);
        if (rc != 0) {
            fprintf(stderr, "msgl_paged_extend: %d %s\n", rc, msgl_last_error());
            exit(1);
        }
    };
    for (int i = 0; i < warmup; ++i) step();
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < steps; ++i) step();
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double flops = 4.0 * hq * D * bs * (double)sq * (prefix + (sq + 1) / 2.0);
    printf("{\"bs\": %d, \"sq\": %d, \"prefix\": %d, \"ps\": %d, \"hq\": %d, \"hkv\": %d, \"ms_per_step\": %.4f, \"TFLOPs\": %.1f, "
           "\"flops_per_step\": %.0f}\n",
           bs, sq, prefix, ps, hq, hkv, ms / steps, flops / (ms / steps * 1e-3) / 1e12, flops);
#ifdef EXT_STAMPS
    {
        unsigned long long st_[16];
        msgl_extend_stamps(st_, 0);
        if (st_[15] == 0) msgl_extend_wide_stamps(st_);  // the launch took the one-wave-per-SIMD kernel
        const double it = (double)st_[15];
        printf("steady iterations sampled: %.0f; cycles per iteration by segment:\n", it);
        const bool wide_ = st_[7] == 0 && st_[8] == 0;  // the four-wave kernel stamps the edges of its four blocks
        const char* names8[9] = {"loop edge", "barrier", "rescale+QK(t+1)+staging", "PV0+max", "PV1+exp12", "PV2+exp12", "PV3+exp8+pack",
                                 "-", "-"};
        const char* names4[9] = {"loop edge", "barrier", "PV k-steps 0,1", "QK second half", "PV k-steps 2,3", "QK first half", "tail of iteration",
                                 "-", "-"};
        const char** names = wide_ ? names4 : names8;
        double tot = 0;
        for (int i = 0; i < 9; ++i) tot += st_[i] / it;
        for (int i = 0; i < 9; ++i) printf("  %-22s %8.1f\n", names[i], st_[i] / it);
        printf("  %-22s %8.1f\n", "total", tot);
        if (st_[11] > 0)
            printf("  whole loop: %.1f cycles per iteration, clock %.3f GHz\n", st_[10] / it, (double)st_[10] / (double)st_[11] * 0.1);
    }
#endif
    return 0;
}
