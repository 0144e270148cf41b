// Micro-experiment: how many cycles one wave per SIMD spends per v_mfma_f32_32x32x16_bf16 with K independent vector
// instructions between consecutive matrix instructions, by where the accumulator lives (AGPR through inline asm, VGPR
// through the builtin) and by the kind of filler.  Prints cycles per matrix instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

template <int MODE, int K, int FILL>
__global__ __launch_bounds__(256, 1) void kern(unsigned long long* out, const float* in, float* sink) {
    f32x16_t acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = in[threadIdx.x + r + i];
    u32x4_t a, b;
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(unsigned, in[threadIdx.x + 100 + i]), b[i] = __builtin_bit_cast(unsigned, in[threadIdx.x + 200 + i]);
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = in[threadIdx.x + 300 + i];
    const float c1 = in[5], c2 = in[6];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(b));
            else acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[g], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
                else if (FILL == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k]));
                else if (FILL == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
            }
            if (FILL == 3) {  // the prefill loop's mix per matrix instruction: fma, exp, add, and a pack every other time; K = variant
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[0]) : "v"(c1), "v"(c2));
                asm volatile("v_exp_f32 %0, %0" : "+v"(f[1]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[2]) : "v"(c1));
                if (g & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[3]) : "v"(c1));
                if (K >= 2) asm volatile("s_waitcnt lgkmcnt(0)");
                if (K >= 4 && (g & 1)) asm volatile("v_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(f[4]) : "v"(c1));
                if (K >= 6) asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"((threadIdx.x & 63) * 16) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 8; ++i) {
        if (MODE == 0) asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int r = 0; r < 16; ++r) s += acc[i][r];
        s += f[i];
    }
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE, int K, int FILL>
void run(const char* name, unsigned long long* d_out, float* d_in, float* d_sink) {
    hipLaunchKernelGGL((kern<MODE, K, FILL>), dim3(256), dim3(256), 0, 0, d_out, d_in, d_sink);
    hipLaunchKernelGGL((kern<MODE, K, FILL>), dim3(256), dim3(256), 0, 0, d_out, d_in, d_sink);
    hipDeviceSynchronize();
    unsigned long long t;
    hipMemcpy(&t, d_out, 8, hipMemcpyDeviceToHost);
    printf("%-28s K=%d: %.1f cycles per MFMA\n", name, K, (double)t / 8000.0);
}

int main() {
    unsigned long long* d_out;
    float *d_in, *d_sink;
    hipMalloc(&d_out, 64);
    hipMalloc(&d_in, 4096 * 4);
    hipMalloc(&d_sink, 256 * 256 * 4);
    float h[4096];
    for (int i = 0; i < 4096; ++i) h[i] = 0.001f * (i % 97) - 0.03f;
    hipMemcpy(d_in, h, sizeof(h), hipMemcpyHostToDevice);
#define RUNK(M, F, NAME) run<M, 0, F>(NAME, d_out, d_in, d_sink); run<M, 2, F>(NAME, d_out, d_in, d_sink); run<M, 4, F>(NAME, d_out, d_in, d_sink); run<M, 6, F>(NAME, d_out, d_in, d_sink); run<M, 8, F>(NAME, d_out, d_in, d_sink);
    RUNK(0, 0, "agpr acc (asm), v_fma")
    RUNK(1, 0, "builtin acc, v_fma")
    RUNK(0, 1, "agpr acc (asm), v_exp")
    RUNK(1, 1, "builtin acc, v_exp")
    RUNK(0, 2, "agpr acc (asm), v_cvt_pk")
    RUNK(0, 3, "agpr, mix (+wait,+addr,+lds)")
    RUNK(1, 3, "builtin, mix (+wait,+addr,+lds)")
    return 0;
}
