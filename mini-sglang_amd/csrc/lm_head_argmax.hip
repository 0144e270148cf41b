// Greedy token straight from the final hidden state: out[b] = argmax_v bf16(sum_k x[b, k] * w[v, k]) over this rank's
// vocabulary rows, without the [bs, vocab] logits ever reaching memory.
//
// Decode regime (bs <= a few hundred): the product is bound by the one pass over the head's weights (1.24 GB for a
// 151 936 x 4096 head), so the kernel is a weight stream with the matrix cores beside it.
//   work group = 4 waves = 128 vocabulary rows x up to 128 batch columns; wave = 32 rows, all columns;
//   C^T tile = W . X^T with v_mfma_f32_32x32x16_bf16, both operands through LDS images filled by DMA (see the kernel).
//   Epilogue: accumulators rounded to bf16 (what F.linear would have stored), best (value, row) per column in the lane,
//   across the lane pair, across the waves, one partial per (row tile, column) in the caller's workspace; a second small
//   kernel folds the row tiles.  Order: torch.argmax's (NaN largest, ties to the lower index).
#include <type_traits>
#include <utility>

#include "msgl_common.h"

namespace msgl {

typedef __attribute__((ext_vector_type(8))) __bf16 lbf16x8_t;
typedef __attribute__((ext_vector_type(16))) float lf32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int lhu32x4_t;

constexpr int kLhRows = 128;      // vocabulary rows per work group

__device__ __forceinline__ bool lh_better(float v, int i, float bv, int bi) {
    const bool vn = v != v, bn = bv != bv;
    if (vn != bn) return vn;
    if (vn) return i < bi;
    return v > bv || (v == bv && i < bi);
}

template <int Lo, int... I, typename F>
__device__ __forceinline__ void static_for_lh_impl(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, Lo + I>{}), ...);
}
template <int Lo, int Hi, typename F>
__device__ __forceinline__ void static_for_lh(F&& f) {
    static_for_lh_impl<Lo>(f, std::make_integer_sequence<int, Hi - Lo>{});
}

struct LmHeadParams {
    const uint16_t* x;
    int64_t x_stride;
    const uint16_t* w;
    int64_t w_stride;
    float* ws_val;   // [row tiles][bs]
    int32_t* ws_idx;
    int bs;
    int vocab;
    int hidden;
};

// One LDS-DMA load: 64 lanes x 16 bytes from `base` + per-lane byte offset `off` land lane-linearly in the 1 KiB at LDS
// byte address `dst` (M0).  `base` and `dst` may have just been written by scalar arithmetic the assembler statement
// cannot see: s_nop 4 covers the wait states a vector-memory read of a fresh scalar needs.  M0 is saved and restored.
template <bool NT>
__device__ __forceinline__ void lh_dma_1k(const void* base, uint32_t off, uint32_t dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(off), "s"(dst)
                     : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_nop 4\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "s"(base), "v"(off), "s"(dst)
                     : "memory");
}
template <int N>
__device__ __forceinline__ void lh_wait_loads() {  // at most N of the wave's loads still outstanding (they land in issue order)
    static_assert(N >= 0 && N < 64, "vmcnt");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Work group = 4 waves = 128 weight rows x 32 NB batch columns, k in chunks of CK elements (a 256- or 512-byte piece of
// a row).  Everything reaches LDS by DMA in whole row pieces (16 or 32 lanes a row, 4 or 2 rows an instruction:
// coalesced, which a load straight into the matrix instruction's operand layout - neighbouring lanes on neighbouring
// rows - is not; the longer piece also opens half as many DRAM pages per byte):
//   weights   wave-private ring of D slots of 32 row pieces, read once, non-temporal;
//   X^T       one image of 32 NB columns per slot, shared by the four waves, each bringing a quarter of it (from L2).
// A 16-byte piece q of row r sits at position q ^ (r & 15) of its row (the swizzle is applied to the source address):
// the ds_read_b128 fragment reads of 16 different rows then fall on 16 different bank groups.
// Iteration c: wait for chunk c's own loads, barrier (X^T of chunk c complete, everyone done with chunk c - 1), issue
// chunk c + D - 1 into the slot that frees, multiply chunk c.  D - 1 chunks per wave are in flight while it multiplies.
template <int NB, int D, int CK>
__global__ __launch_bounds__(256, 1) void lm_head_argmax_kernel(LmHeadParams p) {
    static_assert(CK == 128 || CK == 256, "a row piece is 256 or 512 bytes");
    constexpr int COLS = 32 * NB;
    constexpr int RB = CK * 2;                 // bytes of a row piece
    constexpr int LPR = RB / 16;               // lanes that bring one row piece
    constexpr int RPI = 64 / LPR;              // rows per load instruction
    constexpr int NWI = 32 / RPI;              // weight load instructions per wave and chunk
    constexpr int kXImage = COLS * RB, kWImage = 32 * RB, kSlot = kXImage + 4 * kWImage;
    constexpr int NXI = COLS / 4 / RPI;        // X^T load instructions per wave and chunk (a quarter of the columns)
    constexpr int PER_CHUNK = NWI + NXI;       // loads a wave issues per chunk
    extern __shared__ __attribute__((aligned(1024))) char lds[];  // D * kSlot; the epilogue reuses the start
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    const int gr = lane / LPR, q = lane % LPR;   // row within a load instruction, 16-byte piece within the row
    const int row0 = blockIdx.x * kLhRows + wave * 32;
    const int col0 = blockIdx.y * COLS;
    const int nchunks = p.hidden / CK;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)lds;

    // sources: rows past the vocabulary and columns past the batch re-read the last valid one (left out in the epilogue)
    const int base_row = min(row0, p.vocab - 1);
    const uint16_t* wbase = p.w + (int64_t)base_row * p.w_stride;
    uint32_t woff[NWI], xoff[NXI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int r = RPI * i + gr;
        woff[i] = (uint32_t)((min(row0 + r, p.vocab - 1) - base_row) * p.w_stride * 2 + ((q ^ (r & 15)) << 4));
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
        const int r = wave * (COLS / 4) + RPI * i + gr;  // column of the image
        xoff[i] = (uint32_t)(min(col0 + r, p.bs - 1) * p.x_stride * 2 + ((q ^ (r & 15)) << 4));
    }
    auto issue_chunk = [&](int c, int slot) __attribute__((always_inline)) {
        c = min(c, nchunks - 1);  // (past the end: a re-read of the last chunk that nothing consumes)
        const uint32_t xdst = lds0 + slot * kSlot + wave * (COLS / 4) * RB;
        const uint32_t wdst = lds0 + slot * kSlot + kXImage + wave * kWImage;
        const uint16_t* xb = p.x + c * CK;
        const uint16_t* wb = wbase + c * CK;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the wave's fragment reads of the slot's last tenant
#pragma unroll
        for (int i = 0; i < NXI; ++i) lh_dma_1k<false>(xb, xoff[i], xdst + i * 1024);
#pragma unroll
        for (int i = 0; i < NWI; ++i) lh_dma_1k<true>(wb, woff[i], wdst + i * 1024);
    };

    lf32x16_t acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    // fragment reads: row n of an image, pieces 2 j + h (the low four bits of a piece's position are swizzled by the row)
    int fpos[CK / 16];
#pragma unroll
    for (int j = 0; j < CK / 16; ++j) fpos[j] = n * RB + (((2 * j + h) ^ (n & 15)) << 4);

#pragma unroll
    for (int s = 0; s < D - 1; ++s) issue_chunk(s, s);
    int slot = 0;
    for (int c = 0; c < nchunks; ++c) {
        lh_wait_loads<(D - 2) * PER_CHUNK>();
        asm volatile("s_barrier" ::: "memory");
        issue_chunk(c + D - 1, slot == 0 ? D - 1 : slot - 1);
        const char* ximg = lds + slot * kSlot;
        const char* wimg = ximg + kXImage + wave * kWImage;
#pragma unroll
        for (int j = 0; j < CK / 16; ++j) {
            const lhu32x4_t a = *reinterpret_cast<const lhu32x4_t*>(wimg + fpos[j]);
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const lhu32x4_t bf = *reinterpret_cast<const lhu32x4_t*>(ximg + b * 32 * RB + fpos[j]);
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(lbf16x8_t, a), __builtin_bit_cast(lbf16x8_t, bf), acc[b], 0, 0, 0);
            }
        }
        slot = slot == D - 1 ? 0 : slot + 1;
    }
    lh_wait_loads<0>();  // the surplus loads of the last iterations still target the ring
    __syncthreads();

    // ---- epilogue: per column, the best of this wave's 32 rows, then of the work group's 128
    float* ex_val = reinterpret_cast<float*>(lds);  // [4][COLS]
    int* ex_idx = reinterpret_cast<int*>(lds) + 4 * COLS;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float best = -INFINITY;
        int besti = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * h;  // rows rise with r: the first maximum is the lowest row
            const float v = round_bf16(acc[b][r]);
            if (row < p.vocab && lh_better(v, row, best, besti)) best = v, besti = row;
        }
        const float ov = __shfl_xor(best, 32);
        const int oi = __shfl_xor(besti, 32);
        if (lh_better(ov, oi, best, besti)) best = ov, besti = oi;
        if (h == 0) ex_val[wave * COLS + 32 * b + n] = best, ex_idx[wave * COLS + 32 * b + n] = besti;
    }
    __syncthreads();
    if (tid < COLS && col0 + tid < p.bs) {
        float best = ex_val[tid];
        int besti = ex_idx[tid];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ov = ex_val[w * COLS + tid];
            const int oi = ex_idx[w * COLS + tid];
            if (lh_better(ov, oi, best, besti)) best = ov, besti = oi;
        }
        p.ws_val[(int64_t)blockIdx.x * p.bs + col0 + tid] = best;
        p.ws_idx[(int64_t)blockIdx.x * p.bs + col0 + tid] = besti;
    }
}

// One work group per batch column folds the row tiles' partials.
__global__ __launch_bounds__(256) void lm_head_argmax_finish_kernel(const float* ws_val, const int32_t* ws_idx, int tiles, int bs, int32_t* out,
                                                                   float* out_val) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.x;
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int t = threadIdx.x; t < tiles; t += 256) {
        const float v = ws_val[(int64_t)t * bs + b];
        const int i = ws_idx[(int64_t)t * bs + b];
        if (lh_better(v, i, best, besti)) best = v, besti = i;
    }
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(besti, off);
        if (lh_better(ov, oi, best, besti)) best = ov, besti = oi;
    }
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = best, si[threadIdx.x >> 6] = besti;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (lh_better(sv[w], si[w], best, besti)) best = sv[w], besti = si[w];
        out[b] = besti == 0x7fffffff ? 0 : besti;
        if (out_val != nullptr) out_val[b] = best;
    }
}

template <int NB, int D, int CK>
static int launch_lm_head(const LmHeadParams& p, int tiles, hipStream_t st) {
    constexpr int kLds = D * (32 * NB * CK * 2 + 4 * 32 * CK * 2);
    static_assert(kLds <= 160 * 1024, "LDS of a compute unit");
    static PerDeviceOnce once;
    if (!once.ensure([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(lm_head_argmax_kernel<NB, D, CK>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess;
        })) {
        set_error("msgl_lm_head_argmax: cannot reserve %d bytes of LDS", kLds);
        return MSGL_E_LAUNCH;
    }
    const dim3 grid((unsigned)tiles, (unsigned)((p.bs + 32 * NB - 1) / (32 * NB)));
    hipLaunchKernelGGL((lm_head_argmax_kernel<NB, D, CK>), grid, dim3(256), kLds, st, p);
    return check_launch("lm_head_argmax_kernel");
}

}  // namespace msgl

extern "C" size_t msgl_lm_head_argmax_workspace_bytes(int bs, int vocab) {
    if (bs <= 0 || vocab <= 0) return 0;
    return (size_t)((vocab + msgl::kLhRows - 1) / msgl::kLhRows) * (size_t)bs * 8;
}

extern "C" int msgl_lm_head_argmax(const void* x, int64_t x_stride, const void* w, int64_t w_stride, int32_t* out, float* out_val, int bs, int vocab,
                                   int hidden, void* workspace, size_t workspace_bytes, void* stream) {
    using namespace msgl;
    MSGL_REQUIRE(x && w && out && workspace, "msgl_lm_head_argmax: null pointer");
    MSGL_REQUIRE(bs > 0 && vocab > 0, "msgl_lm_head_argmax: bs %d and vocab %d must be positive", bs, vocab);
    MSGL_REQUIRE(hidden >= 256 && hidden % 256 == 0, "msgl_lm_head_argmax: hidden %d must be a positive multiple of 256", hidden);
    MSGL_REQUIRE(x_stride >= hidden && w_stride >= hidden && x_stride % 8 == 0 && w_stride % 8 == 0 && aligned16(x) && aligned16(w),
                 "msgl_lm_head_argmax: rows must be 16-byte aligned and at least hidden elements apart");
    // (the DMA loads address a lane's piece as a 32-bit byte offset from the tensor's base)
    MSGL_REQUIRE((int64_t)bs * x_stride * 2 < (int64_t)1 << 32 && (int64_t)32 * w_stride * 2 < (int64_t)1 << 32,
                 "msgl_lm_head_argmax: hidden states of %lld bytes or more are not supported", (long long)((int64_t)1 << 32));
    const size_t need = msgl_lm_head_argmax_workspace_bytes(bs, vocab);
    if (workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 3u)) {
        set_error("msgl_lm_head_argmax: workspace of %zu bytes, %zu needed (4-byte aligned)", workspace_bytes, need);
        return MSGL_E_WORKSPACE;
    }
    const int tiles = (vocab + kLhRows - 1) / kLhRows;
    LmHeadParams p;
    p.x = static_cast<const uint16_t*>(x);
    p.x_stride = x_stride;
    p.w = static_cast<const uint16_t*>(w);
    p.w_stride = w_stride;
    p.ws_val = static_cast<float*>(workspace);
    p.ws_idx = reinterpret_cast<int32_t*>(p.ws_val + (size_t)tiles * bs);
    p.bs = bs;
    p.vocab = vocab;
    p.hidden = hidden;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc;
    // (up to 32 columns the LDS holds two slots of 512-byte row pieces: 5.26 against 4.8 TB/s with three slots of 256-byte
    // ones on a 151 936 x 4096 head; short rows and launches of less than two row blocks per compute unit are ahead with
    // the shorter pieces, whose pipeline fills sooner)
    if (bs <= 32 && hidden >= 2048 && tiles >= 2 * device_cus()) rc = launch_lm_head<1, 2, 256>(p, tiles, st);
    else if (bs <= 32) rc = launch_lm_head<1, 3, 128>(p, tiles, st);
    else if (bs <= 64) rc = launch_lm_head<2, 3, 128>(p, tiles, st);
    else rc = launch_lm_head<4, 2, 128>(p, tiles, st);
    if (rc != MSGL_OK) return rc;
    hipLaunchKernelGGL(lm_head_argmax_finish_kernel, dim3((unsigned)bs), dim3(256), 0, st, p.ws_val, p.ws_idx, tiles, bs, out, out_val);
    return check_launch("lm_head_argmax_finish_kernel");
}
