// Micro-experiment: how many cycles one wave per SIMD spends per v_mfma_f32_32x32x16_bf16 with K independent vector
// instructions between consecutive matrix instructions, by where the accumulator lives (AGPR through inline asm, VGPR
// through the builtin) and by the kind of filler.  Prints cycles per matrix instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

// FILL 11: the prefill loop's whole per-matrix-instruction mix with LDS reads that nothing waits for: fma, exp, add, a
// pack every other time, an integer op every other time, K / 2 ds_read_b128 (K = 2: one per matrix instruction, 3: three
// in two) -- run with one wave per SIMD (256 threads) and with two (512 threads: NT).
template <int MODE, int K, int FILL, int NT = 256>
__global__ __launch_bounds__(NT, 1) void kern(unsigned long long* out, const float* in, float* sink) {
    f32x16_t acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = in[threadIdx.x + r + i];
    u32x4_t a, b;
    for (int i = 0; i < 4; ++i) a[i] = __builtin_bit_cast(unsigned, in[threadIdx.x + 100 + i]), b[i] = __builtin_bit_cast(unsigned, in[threadIdx.x + 200 + i]);
    float f[8];
    for (int i = 0; i < 8; ++i) f[i] = in[threadIdx.x + 300 + i];
    const float c1 = in[5], c2 = in[6];
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    f32x2_t pk[8], pc1 = {c1, c2}, pc2 = {c2, c1};
    for (int i = 0; i < 8; ++i) pk[i] = f32x2_t{f[i], f[(i + 1) & 7]};
    u32x4_t ld[4] = {a, b, a, b};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < 1000; ++it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            // MODE 2: the waves of a SIMD in opposite roles: waves 0-3 issue matrix instructions only, waves 4-7 only the fillers
            if (MODE == 2) {
                if (threadIdx.x < 256) {
                    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(b));
                    __builtin_amdgcn_sched_barrier(0);
                    continue;
                }
            } else if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[g]) : "v"(a), "v"(b));
            else acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), acc[g], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
                else if (FILL == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f[k]));
                else if (FILL == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
                else if (FILL == 4) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pk[k]) : "v"(pc1), "v"(pc2));
                else if (FILL == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[k]) : "v"(pc1));
                else if (FILL == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
                else if (FILL == 7) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[k & 3]) : "v"((threadIdx.x & 63) * 16 + k * 1024) : "memory");
                else if (FILL == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[k]) : "v"(pc1));
                else if (FILL == 9) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
                else if (FILL == 10) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[k]) : "v"(c1));
                else if (FILL == 12) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(f[k]) : "v"(c1), "v"(c2));
                else if (FILL == 13) asm volatile("v_dot2_f32_bf16 %0, %1, %2, %0" : "+v"(f[k]) : "v"(c1), "v"(c2));
                else if (FILL == 14) asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(pk[k]) : "v"((threadIdx.x & 63) * 8 + k * 512) : "memory");
            }
            if (FILL == 11) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[0]) : "v"(c1), "v"(c2));
                asm volatile("v_exp_f32 %0, %0" : "+v"(f[1]));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[2]) : "v"(c1));
                if (g & 1) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(f[3]) : "v"(c1));
                if (g & 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(f[4]) : "v"(c1));
                if (K >= 2) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[g & 3]) : "v"((threadIdx.x & 63) * 16 + g * 1024) : "memory");
                if (K >= 3 && (g & 1)) asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(ld[(g + 1) & 3]) : "v"((threadIdx.x & 63) * 16 + g * 1024) : "memory");
                if (K >= 2 && (g & 3) == 3) asm volatile("s_waitcnt lgkmcnt(2)");
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
        s += f[i] + pk[i][0] + pk[i][1] + __builtin_bit_cast(float, ld[i & 3][0]);
    }
    sink[blockIdx.x * 256 + (threadIdx.x & 255)] = s;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) out[threadIdx.x >> 6] = t1 - t0, out[8 + (threadIdx.x >> 6)] = t0, out[16 + (threadIdx.x >> 6)] = t1;
}

template <int MODE, int K, int FILL, int NT = 256>
void run(const char* name, unsigned long long* d_out, float* d_in, float* d_sink) {
    hipLaunchKernelGGL((kern<MODE, K, FILL, NT>), dim3(256), dim3(NT), 0, 0, d_out, d_in, d_sink);
    hipLaunchKernelGGL((kern<MODE, K, FILL, NT>), dim3(256), dim3(NT), 0, 0, d_out, d_in, d_sink);
    hipDeviceSynchronize();
    unsigned long long t[24];
    hipMemcpy(t, d_out, sizeof(t), hipMemcpyDeviceToHost);
    // cycles per matrix instruction of a SIMD: a SIMD runs NT / 256 waves, each issuing 8000; span = first start to last end
    unsigned long long lo = ~0ull, hi = 0;
    for (int w = 0; w < NT / 64; ++w) lo = t[8 + w] < lo ? t[8 + w] : lo, hi = t[16 + w] > hi ? t[16 + w] : hi;
    printf("%-28s K=%d waves/SIMD=%d: %.1f cycles per MFMA per SIMD (span); own time of waves 0 / last: %.1f / %.1f per own MFMA\n", name, K, NT / 256,
           (double)(hi - lo) / 8000.0 / (NT / 256), (double)t[0] / 8000.0, (double)t[NT / 64 - 1] / 8000.0);
}

int main() {
    unsigned long long* d_out;
    float *d_in, *d_sink;
    hipMalloc(&d_out, 256);
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
    run<0, 0, 11>("agpr, loop mix, no lds", d_out, d_in, d_sink); run<0, 2, 11>("agpr, loop mix, 1 lds", d_out, d_in, d_sink); run<0, 3, 11>("agpr, loop mix, 1.5 lds", d_out, d_in, d_sink);
    run<0, 0, 11, 512>("agpr, loop mix, no lds", d_out, d_in, d_sink); run<0, 2, 11, 512>("agpr, loop mix, 1 lds", d_out, d_in, d_sink); run<0, 3, 11, 512>("agpr, loop mix, 1.5 lds", d_out, d_in, d_sink);
    run<2, 8, 0, 512>("roles: mfma | 8 v_fma", d_out, d_in, d_sink); run<2, 4, 1, 512>("roles: mfma | 4 v_exp", d_out, d_in, d_sink);
    run<2, 0, 11, 512>("roles: mfma | loop mix", d_out, d_in, d_sink); run<2, 3, 11, 512>("roles: mfma | mix+1.5 lds", d_out, d_in, d_sink);
    run<2, 8, 7, 512>("roles: mfma | 8 ds_read", d_out, d_in, d_sink);
    run<0, 0, 0, 512>("agpr, nothing else", d_out, d_in, d_sink); run<0, 4, 1, 512>("agpr, 4 v_exp", d_out, d_in, d_sink); run<0, 8, 0, 512>("agpr, 8 v_fma", d_out, d_in, d_sink);
    RUNK(0, 12, "agpr acc (asm), v_dot2c_f32_bf16")
    RUNK(0, 13, "agpr acc (asm), v_dot2_f32_bf16")
    RUNK(0, 14, "agpr acc (asm), ds_read_b64_tr")
    RUNK(0, 4, "agpr acc (asm), v_pk_fma_f32")
    RUNK(0, 5, "agpr acc (asm), v_pk_mul_f32")
    RUNK(0, 8, "agpr acc (asm), v_pk_add_f32")
    RUNK(0, 6, "agpr acc (asm), v_add_f32")
    RUNK(0, 10, "agpr acc (asm), v_mul_f32")
    RUNK(0, 9, "agpr acc (asm), v_max3_f32")
    RUNK(0, 7, "agpr acc (asm), ds_read_b128")
    RUNK(0, 3, "agpr, mix (+wait,+addr,+lds)")
    RUNK(1, 3, "builtin, mix (+wait,+addr,+lds)")
    return 0;
}
