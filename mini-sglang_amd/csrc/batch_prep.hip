// Device-side preparation of a decode step: what BatchInfo.prepare_for_decode does on the host
// (mini_sglang/managers/batch_info.py:247-302) with the page allocator's take-from-the-front order
// (mini_sglang/mem_cache/token2kv.py:186-196), as one small kernel, so that consecutive decode steps can
// be chained on the GPU (and captured in one hipGraph) without a host round trip.
//
// One work group of 1024 threads walks the batch in chunks.  Pass 1 counts the requests that start a new
// page; if the free list cannot serve them all, or a request would outgrow its page-table row, the step is
// refused as a whole: status != 0, lengths, table and free list untouched, and every out_cache_loc entry
// pointed at the dummy slot 0, so that the forward pass that follows in a captured step overwrites no
// request's K/V.  A refusal is sticky (every later step is refused too) until the host clears status.
// Pass 2 assigns pages in batch order with a block-wide exclusive scan, so request i gets exactly the page
// the host allocator would have handed it.
#include "msgl_common.h"

namespace msgl {

struct PrepParams {
    const int32_t* req_idx;
    int32_t* seq_lens;
    int32_t* last_slots;
    const int32_t* free_pages;
    int32_t free_capacity;
    int32_t* free_range;  // [0] = first page not handed out yet (advanced here), [1] = end of the valid entries
    int32_t* r2t;
    int64_t r2t_stride;
    int32_t* r2p;  // optional page table: [row][position / page_size] = page id, written when a request takes a page
    int64_t r2p_stride;
    int32_t max_ctx;
    int32_t* positions;
    int32_t* out_loc;
    const int32_t* next_tokens;
    int32_t* input_ids;
    int32_t* status;
    int bs;
    int ps;
};

constexpr int kPrepThreads = 1024;

// exclusive prefix of `flag` over the block and the block total; all threads call it
__device__ __forceinline__ int block_scan_flags(bool flag, int* lds_wave, int& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(flag);
    const int in_wave = __popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();  // lds_wave may still be read from the previous call
    if (lane == 0) lds_wave[wave] = __popcll(mask);
    __syncthreads();
    int before = 0, sum = 0;
#pragma unroll
    for (int w = 0; w < kPrepThreads / 64; ++w) {
        const int c = lds_wave[w];
        before += (w < wave) ? c : 0;
        sum += c;
    }
    total = sum;
    return before + in_wave;
}

__global__ __launch_bounds__(kPrepThreads) void decode_prepare_kernel(PrepParams p) {
    __shared__ int lds_wave[kPrepThreads / 64];
    const int tid = threadIdx.x;
    const int head = p.free_range[0];
    const int end = min(p.free_range[1], p.free_capacity);

    // pass 1: pages needed, longest row after the step
    int need_total = 0;
    bool too_long = false;
    for (int c0 = 0; c0 < p.bs; c0 += kPrepThreads) {
        const int i = c0 + tid;
        const int old = i < p.bs ? p.seq_lens[i] : 0;
        const bool need = i < p.bs && (p.ps == 1 || old % p.ps == 0);
        too_long |= (i < p.bs && old >= p.max_ctx);
        int chunk_total;
        block_scan_flags(need, lds_wave, chunk_total);
        need_total += chunk_total;
    }
    const int earlier = *p.status;
    const int refuse = earlier ? earlier : (__syncthreads_or(too_long) ? 2 : (need_total > end - head ? 1 : 0));
    if (refuse) {
        for (int i = tid; i < p.bs; i += kPrepThreads) p.out_loc[i] = 0;
        __syncthreads();  // every thread has read the status word
        if (tid == 0) *p.status = refuse;
        return;
    }

    // pass 2: assign
    int running = 0;
    for (int c0 = 0; c0 < p.bs; c0 += kPrepThreads) {
        const int i = c0 + tid;
        const bool valid = i < p.bs;
        const int old = valid ? p.seq_lens[i] : 0;
        const bool need = valid && (p.ps == 1 || old % p.ps == 0);
        int chunk_total;
        const int rank = block_scan_flags(need, lds_wave, chunk_total);
        if (valid) {
            const int slot = need ? p.free_pages[head + running + rank] * p.ps : p.last_slots[i] + 1;
            p.positions[i] = old;
            p.seq_lens[i] = old + 1;
            p.out_loc[i] = slot;
            p.last_slots[i] = slot;
            p.r2t[(int64_t)p.req_idx[i] * p.r2t_stride + old] = slot;
            if (p.r2p && need && p.ps > 1) p.r2p[(int64_t)p.req_idx[i] * p.r2p_stride + old / p.ps] = slot / p.ps;
            if (p.input_ids) p.input_ids[i] = p.next_tokens[i];
        }
        running += chunk_total;
    }
    if (tid == 0) p.free_range[0] = head + running;
}

}  // namespace msgl

using namespace msgl;

extern "C" int msgl_prepare_decode_paged(const int32_t* req_pool_indices, int32_t* seq_lens, int32_t* last_slots,
                                         const int32_t* free_pages, int32_t free_capacity, int32_t* free_range, int32_t* req_to_token,
                                         int64_t r2t_stride, int32_t* req_to_page, int64_t r2p_stride, int32_t max_ctx,
                                         int32_t* positions, int32_t* out_cache_loc, const int32_t* next_tokens, int32_t* input_ids,
                                         int32_t* status, int bs, int page_size, void* stream) {
    if (bs == 0) return MSGL_OK;
    MSGL_REQUIRE(bs > 0 && page_size > 0 && free_capacity >= 0 && max_ctx > 0,
                 "msgl_prepare_decode: bs=%d page_size=%d free_capacity=%d max_ctx=%d", bs, page_size, free_capacity, max_ctx);
    MSGL_REQUIRE(req_pool_indices && seq_lens && last_slots && free_pages && free_range && req_to_token && positions &&
                     out_cache_loc && status,
                 "msgl_prepare_decode: null pointer argument");
    MSGL_REQUIRE((next_tokens == nullptr) == (input_ids == nullptr), "msgl_prepare_decode: next_tokens and input_ids go together");
    MSGL_REQUIRE(r2t_stride >= max_ctx, "msgl_prepare_decode: r2t_stride %lld < max_ctx %d", (long long)r2t_stride, max_ctx);
    MSGL_REQUIRE(req_to_page == nullptr || r2p_stride >= (max_ctx + page_size - 1) / page_size,
                 "msgl_prepare_decode: r2p_stride %lld < pages of a row", (long long)r2p_stride);
    PrepParams p;
    p.req_idx = req_pool_indices;
    p.seq_lens = seq_lens;
    p.last_slots = last_slots;
    p.free_pages = free_pages;
    p.free_capacity = free_capacity;
    p.free_range = free_range;
    p.r2t = req_to_token;
    p.r2t_stride = r2t_stride;
    p.r2p = req_to_page;
    p.r2p_stride = r2p_stride;
    p.max_ctx = max_ctx;
    p.positions = positions;
    p.out_loc = out_cache_loc;
    p.next_tokens = next_tokens;
    p.input_ids = input_ids;
    p.status = status;
    p.bs = bs;
    p.ps = page_size;
    hipLaunchKernelGGL(decode_prepare_kernel, dim3(1), dim3(kPrepThreads), 0, static_cast<hipStream_t>(stream), p);
    return check_launch("decode_prepare_kernel");
}

extern "C" int msgl_prepare_decode(const int32_t* req_pool_indices, int32_t* seq_lens, int32_t* last_slots,
                                   const int32_t* free_pages, int32_t free_capacity, int32_t* free_range, int32_t* req_to_token,
                                   int64_t r2t_stride, int32_t max_ctx, int32_t* positions, int32_t* out_cache_loc,
                                   const int32_t* next_tokens, int32_t* input_ids, int32_t* status, int bs, int page_size,
                                   void* stream) {
    return msgl_prepare_decode_paged(req_pool_indices, seq_lens, last_slots, free_pages, free_capacity, free_range, req_to_token,
                                     r2t_stride, nullptr, 0, max_ctx, positions, out_cache_loc, next_tokens, input_ids, status, bs,
                                     page_size, stream);
}
