// Shared by the prefill (extend) attention kernels: launch parameters, LDS tile geometry, staging of a 64-key K/V tile
// through registers, and small single-instruction helpers.
#pragma once
#include "msgl_common.h"
#include "decode_params.h"

namespace msgl {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;

constexpr int kBM = 128;          // query rows per work group (per head)
constexpr int kBN = 64;           // keys per LDS tile
constexpr int kHD = 128;          // head dim
constexpr int kKRow = 256;        // bytes per K row in LDS (swizzled 16-byte chunks)
constexpr int kVRow = 320;        // bytes per V row in LDS (padded: conflict-free transposed reads)
constexpr int kKRowPadded = 272;  // K rows of the four-wave kernel: 16 bytes of padding instead of a swizzle
constexpr int kKBytes = kBN * kKRow;
constexpr int kTileBytes = kBN * kKRow + kBN * kVRow;

constexpr float kNegBigE = -1.0e30f;
constexpr int kShortKv = 1024;    // longest request (prefix + new tokens) that still takes the two-slot kernel
constexpr int kWideMinKv = 1536;  // longest request for which the eight-wave kernel is preferred to the four-wave one

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
    const int32_t* plan;  // optional: (request, query tile) items in launch order, see extend_plan_kernel
    // Key split: ksplit > 1 cuts every work item's key tiles into ksplit equal runs, one work group each, so that a
    // launch of few items with long key ranges still covers the chip; the groups leave partial results (unnormalised
    // O, running max, sum: [token][head][split]) in ws_o / ws_ml and extend_merge_kernel joins them.
    int ksplit;
    float* ws_o;
    float* ws_ml;
};

// Shared-prefix mode (GROUPED): the "query rows" of a work group are the decode queries of the requests that share
// one radix prefix, (member, head of the GQA group) pairs of one kv head, and the keys are a slice of that prefix,
// visible to every row.  The result is not an output row but a split partial (unnormalised O, running max, sum)
// in the decode workspace, in the slots behind the request's own splits, for the decode merge kernel.
struct GroupedParams {
    const int32_t* first;   // per group: index of its first member (in the batch, or in `members` when that is given)
    const int32_t* count;   // per group: number of member requests
    const int32_t* members; // optional: batch indices of the members, group after group; NULL = members are consecutive
    const int32_t* prefix;  // per group: shared prefix length in tokens
    int key_chunk;          // keys per work group (a multiple of 64)
    int key_splits;         // ceil(longest prefix / key_chunk)
    int row_tiles;          // ceil(most members * heads per kv head / 128)
    // Captured launches: the grid cannot follow the group structure, so the host lists the work items
    // (group << 20 | row tile << 14 | key slice; -1 = none) and the slice length in device memory.
    const int32_t* items;
    const int32_t* key_chunk_dev;
    DecodeParams dec;       // seq_lens, start_lens, split size, max_splits, Hq, workspace of the decode launch
};

__device__ __forceinline__ bf16x8_t as_bf16x8(const uint4& v) { return __builtin_bit_cast(bf16x8_t, v); }

// v_max3_f32 on raw matrix-core outputs (fmaxf would first canonicalise each operand with a v_max x, x)
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
// max of m and eight more values: four chained v_max3_f32 in one statement (no padding between them)
__device__ __forceinline__ float max9_raw(float m, float a, float b, float c, float d, float e, float f, float g, float h) {

    asm("v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %0, %0, %3, %4\n\tv_max3_f32 %0, %0, %5, %6\n\tv_max3_f32 %0, %0, %7, %8"
        : "+v"(m)
        : "v"(a), "v"(b), "v"(c), "v"(d), "v"(e), "v"(f), "v"(g), "v"(h));
    return m;
}

// max over the two lanes that share a query column (lane, lane ^ 32): one v_permlane32_swap
__device__ __forceinline__ float max_with_lane_xor32(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // (the two results go through scalars: a bit cast applied straight to an element of the result vector reads
    // element 0 whichever element is named)
    const unsigned lo = r[0], hi = r[1];
    float m;  // raw v_max_f32: the values are maxima of matrix-core outputs, no canonicalising v_max x, x needed
    asm("v_max_f32 %0, %1, %2" : "=v"(m) : "v"(lo), "v"(hi));
    return m;
}

// Staging of one 64-key tile through registers: thread owns 16-byte chunk c16 of KPT keys
// skey + i * KSTEP.  Positions past the last visible key re-read that key (always a valid
// slot), so the loads are unconditional; the causal mask discards those columns.
template <int KPT>
struct StageRegs {
    u32x4_t k[KPT], v[KPT];
};
template <int KPT>
struct StageSlots {
    uint32_t s[KPT];
};

template <int KPT, int KSTEP>
__device__ __forceinline__ StageSlots<KPT> stage_slots(const int32_t* r2t_row, int kt, int skey, int kv_len) {
    const int base = kt * kBN + skey;
    StageSlots<KPT> r;
#pragma unroll
    for (int i = 0; i < KPT; ++i) r.s[i] = (uint32_t)r2t_row[min(base + i * KSTEP, kv_len - 1)];
    return r;
}

template <int KPT>
__device__ __forceinline__ StageRegs<KPT> stage_load(const uint16_t* kcol, const uint16_t* vcol, uint32_t cache_stride,
                                                     const StageSlots<KPT>& sl) {
    StageRegs<KPT> r;
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint64_t o = (uint64_t)sl.s[i] * cache_stride;
        r.k[i] = *reinterpret_cast<const u32x4_t*>(kcol + o);
        r.v[i] = *reinterpret_cast<const u32x4_t*>(vcol + o);
    }
    return r;
}

// The K image of a tile comes in two forms: rows of kKRow bytes with the 16-byte chunks XOR-swizzled by the row
// (KROW == kKRow), or rows padded to kKRowPadded bytes with the chunks in place (both conflict-free for the fragment
// reads; with padded rows the eight k-steps of a lane's fragments are one base address plus immediates).
template <int KROW>
__device__ __forceinline__ int k_chunk_offset(int key, int c16) {
    return key * KROW + ((KROW == kKRow ? (c16 ^ (key & 15)) : c16) << 4);
}

template <int KPT, int KSTEP, int KROW = kKRow>
__device__ __forceinline__ void stage_write(char* tile, int skey, int c16, const StageRegs<KPT>& r) {
    // keys skey + i * KSTEP: KSTEP is a multiple of 16, so (key & 15) == (skey & 15) for every i
    char* kd = tile + k_chunk_offset<KROW>(skey, c16);
    char* vd = tile + kBN * KROW + skey * kVRow + (c16 << 4);
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        *reinterpret_cast<u32x4_t*>(kd + i * KSTEP * KROW) = r.k[i];
        *reinterpret_cast<u32x4_t*>(vd + i * KSTEP * kVRow) = r.v[i];
    }
}

// one key of a thread's share at a time, for callers that spread the staging over their matrix work
template <int KSTEP, int KROW = kKRow, int KPT>
__device__ __forceinline__ void stage_write_one(char* tile, int skey, int c16, const StageRegs<KPT>& r, int i) {
    *reinterpret_cast<u32x4_t*>(tile + k_chunk_offset<KROW>(skey, c16) + i * KSTEP * KROW) = r.k[i];
    *reinterpret_cast<u32x4_t*>(tile + kBN * KROW + skey * kVRow + (c16 << 4) + i * KSTEP * kVRow) = r.v[i];
}
template <int KPT>
__device__ __forceinline__ void stage_load_one(const uint16_t* kcol, const uint16_t* vcol, uint32_t cache_stride, const StageSlots<KPT>& sl,
                                               StageRegs<KPT>& r, int i) {
    const uint64_t o = (uint64_t)sl.s[i] * cache_stride;
    r.k[i] = *reinterpret_cast<const u32x4_t*>(kcol + o);
    r.v[i] = *reinterpret_cast<const u32x4_t*>(vcol + o);
}

// Diagnostic build only (-DEXT_STAMPS, see tools/extend_bench.hip): cycle stamps around the segments of the steady-state
// loop, summed over two waves of every work group.  Each stamp drains the LDS queue, so the segment times are upper
// bounds; no stamp exists in the library build.  -DEXT_STAMPS=2 stamps the loop as a whole only (cycles and the 100 MHz
// real-time counter: cycles per iteration undisturbed, and the clock the chip holds in the loop).
#if defined(EXT_STAMPS) && EXT_STAMPS != 2
#define STAMP(i)                                                             \
    do {                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                   \
        unsigned long long t_;                                               \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        acc_[i] += t_ - last_;                                               \
        last_ = t_;                                                          \
        __builtin_amdgcn_sched_barrier(0);                                   \
    } while (0)
#else
#define STAMP(i)
#endif
}  // namespace msgl
