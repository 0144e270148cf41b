// Standalone driver for msgl_paged_decode: builds a scattered page table and random bf16 pools with
// hipMalloc memory and times the kernel with HIP events.  Used for profiler passes (rocprofv3 --pmc)
// and quick parameter sweeps without a Python process.
//
//   decode_bench [--bs 256] [--ctx 8192] [--ps 16] [--hq 32] [--hkv 8] [--layers 4] [--steps 20]
//                [--warmup 5] [--chunk 4096] [--splits 8] [--interleaved 1]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <random>
#include <string>
#include <vector>

#include "../include/msgl_hip.h"
#ifdef DEC_STAMPS
extern "C" void msgl_decode_stamps(unsigned long long* out, int groups);
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
    int bs = 256, ctx = 8192, ps = 16, hq = 32, hkv = 8, layers = 4, steps = 20, warmup = 5, chunk = 4096, splits = 8,
        interleaved = 1, kv_stream = 1;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        const int v = atoi(argv[i + 1]);
        if (a == "--bs") bs = v;
        else if (a == "--ctx") ctx = v;
        else if (a == "--ps") ps = v;
        else if (a == "--hq") hq = v;
        else if (a == "--hkv") hkv = v;
        else if (a == "--layers") layers = v;
        else if (a == "--steps") steps = v;
        else if (a == "--warmup") warmup = v;
        else if (a == "--chunk") chunk = v;
        else if (a == "--splits") splits = v;
        else if (a == "--interleaved") interleaved = v;
        else if (a == "--stream") kv_stream = v;
        // kernel-form overrides (msgl_set_variant): what the library would otherwise choose from the shape
        else if (a == "--mfma") msgl_set_variant(MSGL_VARIANT_DECODE_MFMA, v);
        else if (a == "--waves") msgl_set_variant(MSGL_VARIANT_DECODE_MFMA_WAVES, v);
        else if (a == "--depth") msgl_set_variant(MSGL_VARIANT_DECODE_MFMA_DEPTH, v);
        else { fprintf(stderr, "unknown flag %s\n", a.c_str()); return 2; }
    }
    const int D = 128;
    const int pages_per_req = (ctx + ps - 1) / ps;
    const size_t pool_pages = (size_t)bs * pages_per_req + 64;
    const size_t slots = (pool_pages + 1) * ps;
    const int64_t row = (int64_t)hkv * D;
    const int64_t cache_stride = interleaved ? 2 * row : row;

    // scattered page table: page ids 1..pool_pages in a fixed pseudo-random order
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
    std::vector<int32_t> seq(bs, ctx), idx(bs);
    std::iota(idx.begin(), idx.end(), 0);

    int32_t *d_r2t, *d_seq, *d_idx;
    CK(hipMalloc(&d_r2t, r2t.size() * 4));
    CK(hipMalloc(&d_seq, bs * 4));
    CK(hipMalloc(&d_idx, bs * 4));
    CK(hipMemcpy(d_r2t, r2t.data(), r2t.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_seq, seq.data(), bs * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_idx, idx.data(), bs * 4, hipMemcpyHostToDevice));

    std::vector<uint16_t*> kbuf(layers), vbuf(layers), qbuf(layers);
    const size_t kv_elems = slots * row;
    for (int l = 0; l < layers; ++l) {
        if (interleaved) {
            uint16_t* kv;
            CK(hipMalloc(&kv, kv_elems * 2 * 2));
            hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, kv, kv_elems * 2, (uint64_t)(l + 1));
            kbuf[l] = kv;
            vbuf[l] = kv + row;
        } else {
            CK(hipMalloc(&kbuf[l], kv_elems * 2));
            CK(hipMalloc(&vbuf[l], kv_elems * 2));
            hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, kbuf[l], kv_elems, (uint64_t)(l + 1));
            hipLaunchKernelGGL(fill_bf16, dim3(4096), dim3(256), 0, 0, vbuf[l], kv_elems, (uint64_t)(l + 101));
        }
        CK(hipMalloc(&qbuf[l], (size_t)bs * hq * D * 2));
        hipLaunchKernelGGL(fill_bf16, dim3(256), dim3(256), 0, 0, qbuf[l], (size_t)bs * hq * D, (uint64_t)(l + 201));
    }
    uint16_t* d_out;
    CK(hipMalloc(&d_out, (size_t)bs * hq * D * 2));
    const size_t ws_bytes = std::max<size_t>(msgl_paged_decode_workspace_bytes(bs, hq, D, splits), 16);
    void* d_ws;
    CK(hipMalloc(&d_ws, ws_bytes));
    CK(hipDeviceSynchronize());

    hipStream_t st;
    CK(hipStreamCreate(&st));
    auto step = [&](int i) {
        const int l = i % layers;
        const int rc = msgl_paged_decode(qbuf[l], (int64_t)hq * D, kbuf[l], vbuf[l], cache_stride, d_r2t, max_ctx, d_idx, d_seq,
                                         d_out, (int64_t)hq * D, d_ws, ws_bytes, bs, hq, hkv, D, ps, 0.08838834764831845f,
                                         splits, chunk, nullptr, kv_stream, nullptr, /*plan=*/nullptr, /*start_lens=*/nullptr, /*extra_splits=*/nullptr, st);
        if (rc != 0) {
            fprintf(stderr, "msgl_paged_decode: %d %s\n", rc, msgl_last_error());
            exit(1);
        }
    };
    for (int i = 0; i < warmup; ++i) step(i);
    CK(hipStreamSynchronize(st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < steps; ++i) step(i);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = 2.0 * hkv * D * 2 * (double)ctx * bs;
    printf("{\"bs\": %d, \"ctx\": %d, \"ps\": %d, \"hq\": %d, \"hkv\": %d, \"chunk\": %d, \"splits\": %d, \"interleaved\": %d, "
           "\"ms_per_step\": %.4f, \"GBps\": %.1f, \"bytes_per_step\": %.0f}\n",
           bs, ctx, ps, hq, hkv, chunk, splits, interleaved, ms / steps, bytes / (ms / steps * 1e-3) / 1e9, bytes);
#ifdef DEC_STAMPS
    {   // where one launch's time goes: the stamps wave 0 of every work group left (100 MHz counter), relative to the
        // earliest work group's entry
        step(0);
        CK(hipStreamSynchronize(st));
        const int groups = std::min(bs * splits, 8192);
        std::vector<unsigned long long> raw((size_t)groups * 6);
        msgl_decode_stamps(raw.data(), groups);
        unsigned long long t0 = ~0ull;
        for (int g = 0; g < groups; ++g)
            if (raw[(size_t)g * 6]) t0 = std::min(t0, raw[(size_t)g * 6]);
        const char* what[6] = {"entry", "page-table entries at hand", "first K tile landed", "tile loop done", "waves met", "partials stored"};
        for (int i = 0; i < 6; ++i) {
            std::vector<double> v;
            for (int g = 0; g < groups; ++g)
                if (raw[(size_t)g * 6] && raw[(size_t)g * 6 + i] >= t0) v.push_back((double)(raw[(size_t)g * 6 + i] - t0) * 0.01);
            if (v.empty()) continue;
            std::sort(v.begin(), v.end());
            printf("  stamp %d %-28s min %6.2f  median %6.2f  p90 %6.2f  max %6.2f us  (%zu groups)\n", i, what[i], v.front(), v[v.size() / 2],
                   v[v.size() * 9 / 10], v.back(), v.size());
        }
    }
#endif
    return 0;
}
