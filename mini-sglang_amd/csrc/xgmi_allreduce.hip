// Sum all-reduce of bf16 activations across the GPUs of one node, for tensor-parallel decode
// steps: every rank maps its peers' staging areas (IPC) and reads them directly over xGMI.
// The links are point to point, so a rank pulls from its 7 peers over 7 links at once; there
// is no ring and no host involvement, and one kernel per call makes the call capturable.
//
// Memory per rank:
//   staging  [2 buffers][2 stages][max_bytes]   plain device memory, written only by its owner,
//            read by everybody; buffer = call parity (see below), stage 0 = copy of the input,
//            stage 1 = this rank's reduced slice (two-step form);
//   flags    [2 sets][kArMaxBlocks][8 sources]  uncached device memory, written by the peers;
//            word (set, block, src) = number of the last call for which block `block` of rank
//            `src` has passed that synchronisation point.
// Work groups synchronise pairwise: group b of every rank touches the same element set, so
// it only has to wait for group b of its peers (no grid-wide step).  A group publishes its
// writes with a system-scope release store into each peer's flag word and acquires the
// peers' words before reading their memory.
// Two buffers suffice without a trailing synchronisation: a rank overwrites buffer (n & 1)
// in call n+2, which it can only start after call n+1, whose wait needed every peer's
// arrival in call n+1 — sent after that peer had finished call n and all its reads.
#include <string.h>

#include "../../include/msgl_comm.h"
#include "msgl_common.h"

namespace msgl {

constexpr int kArThreads = 512;
constexpr int kArMaxBlocks = 128;
constexpr long long kArTimeoutTicks = 300000000LL;  // wall_clock64 runs at 100 MHz: 3 s

struct ArParams {
    const uint4* peer_data[MSGL_AR_MAX_WORLD];  // staging areas in rank order (own one included)
    uint32_t* peer_flags[MSGL_AR_MAX_WORLD];
    uint32_t* epoch;   // [kArMaxBlocks]: calls completed by group b of this rank
    uint32_t* error;   // set to 1 when a wait timed out
    const uint4* in;
    uint4* out;
    int64_t nvec;       // 16-byte vectors in the message
    int64_t chunk_vec;  // two-step form: vectors per rank slice
    int64_t area_vec;   // vectors per (buffer, stage) area
    int rank, world;
};

// Tell every peer that this group's writes up to here are visible, then wait until every
// peer's group has said the same.  False after a timeout (the caller skips its reads).
__device__ __forceinline__ bool peers_arrive(const ArParams& p, int set, uint32_t e, int* s_ok) {
    if (threadIdx.x == 0) *s_ok = 1;
    __syncthreads();  // every wave's stores are issued and counted before the release below
    const int t = threadIdx.x;
    if (t < p.world && t != p.rank) {
        const int slot = (set * kArMaxBlocks + blockIdx.x) * MSGL_AR_MAX_WORLD;
        __scoped_atomic_store_n(p.peer_flags[t] + slot + p.rank, e, __ATOMIC_RELEASE, __MEMORY_SCOPE_SYSTEM);
        const uint32_t* mine = p.peer_flags[p.rank] + slot + t;
        const long long t0 = wall_clock64();
        while ((int32_t)(__scoped_atomic_load_n(mine, __ATOMIC_ACQUIRE, __MEMORY_SCOPE_SYSTEM) - e) < 0) {
            if (wall_clock64() - t0 > kArTimeoutTicks) {
                *s_ok = 0;
                atomicExch(p.error, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    __syncthreads();
    return *s_ok != 0;
}

__device__ __forceinline__ void add8(float (&acc)[8], const uint4& v) {
    acc[0] += bf16_lo(v.x), acc[1] += bf16_hi(v.x);
    acc[2] += bf16_lo(v.y), acc[3] += bf16_hi(v.y);
    acc[4] += bf16_lo(v.z), acc[5] += bf16_hi(v.z);
    acc[6] += bf16_lo(v.w), acc[7] += bf16_hi(v.w);
}

// Sum of vector `idx` of every rank's area at offset `off`, ranks in ascending order (the same
// order on every rank: identical bits everywhere).  All loads are issued before the first add.
template <int W>
__device__ __forceinline__ uint4 sum_ranks(const ArParams& p, int64_t off, int64_t idx) {
    constexpr int N = W ? W : MSGL_AR_MAX_WORLD;
    const int world = W ? W : p.world;
    uint4 v[N];
#pragma unroll
    for (int r = 0; r < N; ++r)
        if (r < world) v[r] = p.peer_data[r][off + idx];
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < N; ++r)
        if (r < world) add8(acc, v[r]);
    return make_uint4(pack_bf16(acc[0], acc[1]), pack_bf16(acc[2], acc[3]), pack_bf16(acc[4], acc[5]), pack_bf16(acc[6], acc[7]));
}

__device__ __forceinline__ uint32_t begin_call(const ArParams& p, uint32_t* s_epoch) {
    if (threadIdx.x == 0) *s_epoch = p.epoch[blockIdx.x] + 1;
    __syncthreads();
    return *s_epoch;
}

// One step: copy the input into the staging area, synchronise, sum everybody's copy.
template <int W>
__global__ __launch_bounds__(kArThreads) void ar_one_shot_kernel(ArParams p) {
    __shared__ uint32_t s_epoch;
    __shared__ int s_ok;
    const uint32_t e = begin_call(p, &s_epoch);
    const int64_t stage0 = (int64_t)(e & 1) * 2 * p.area_vec;
    uint4* mine = const_cast<uint4*>(p.peer_data[p.rank]) + stage0;
    const int64_t stride = (int64_t)gridDim.x * kArThreads;
    const int64_t first = (int64_t)blockIdx.x * kArThreads + threadIdx.x;
    for (int64_t i = first; i < p.nvec; i += stride) mine[i] = p.in[i];
    if (peers_arrive(p, 0, e, &s_ok))
        for (int64_t i = first; i < p.nvec; i += stride) p.out[i] = sum_ranks<W>(p, stage0, i);
    if (threadIdx.x == 0) p.epoch[blockIdx.x] = e;
}

// Two steps: rank r sums slice r of everybody's input, then everybody collects the slices.
// Group b works on the same within-slice index set in every phase and on every rank.
template <int W>
__global__ __launch_bounds__(kArThreads) void ar_two_shot_kernel(ArParams p) {
    __shared__ uint32_t s_epoch;
    __shared__ int s_ok;
    const int world = W ? W : p.world;
    const uint32_t e = begin_call(p, &s_epoch);
    const int64_t stage0 = (int64_t)(e & 1) * 2 * p.area_vec;
    const int64_t stage1 = stage0 + p.area_vec;
    uint4* mine = const_cast<uint4*>(p.peer_data[p.rank]);
    const int64_t stride = (int64_t)gridDim.x * kArThreads;
    const int64_t first = (int64_t)blockIdx.x * kArThreads + threadIdx.x;
    const int64_t cv = p.chunk_vec;
    for (int64_t j = first; j < cv; j += stride)
        for (int c = 0; c < world; ++c) {
            const int64_t i = c * cv + j;
            if (i < p.nvec) mine[stage0 + i] = p.in[i];
        }
    bool ok = peers_arrive(p, 0, e, &s_ok);
    if (ok)
        for (int64_t j = first; j < cv; j += stride) {
            const int64_t i = p.rank * cv + j;
            if (i < p.nvec) mine[stage1 + j] = sum_ranks<W>(p, stage0, i);
        }
    ok = peers_arrive(p, 1, e, &s_ok) && ok;
    if (ok)
        for (int64_t j = first; j < cv; j += stride) {
            constexpr int N = W ? W : MSGL_AR_MAX_WORLD;
            uint4 v[N];
#pragma unroll
            for (int c = 0; c < N; ++c)
                if (c < world && c * cv + j < p.nvec) v[c] = p.peer_data[c][stage1 + j];
#pragma unroll
            for (int c = 0; c < N; ++c)
                if (c < world && c * cv + j < p.nvec) p.out[c * cv + j] = v[c];
        }
    if (threadIdx.x == 0) p.epoch[blockIdx.x] = e;
}

template <int W>
static void launch_ar(bool one_shot, int blocks, hipStream_t st, const ArParams& p) {
    if (one_shot) hipLaunchKernelGGL(ar_one_shot_kernel<W>, dim3(blocks), dim3(kArThreads), 0, st, p);
    else hipLaunchKernelGGL(ar_two_shot_kernel<W>, dim3(blocks), dim3(kArThreads), 0, st, p);
}

}  // namespace msgl

using namespace msgl;

struct msgl_ar {
    int rank, world;
    size_t max_bytes;
    void* data;
    uint32_t* flags;
    uint32_t* state;  // [kArMaxBlocks] epochs, then the error word
    void* peer_data[MSGL_AR_MAX_WORLD];
    void* peer_flags[MSGL_AR_MAX_WORLD];
    bool opened;
};

static constexpr size_t kFlagBytes = 2 * (size_t)kArMaxBlocks * MSGL_AR_MAX_WORLD * sizeof(uint32_t);

#define MSGL_HIP_OK(call, what)                                           \
    do {                                                                  \
        const hipError_t err_ = (call);                                   \
        if (err_ != hipSuccess) {                                         \
            set_error("%s: %s", what, hipGetErrorString(err_));          \
            return MSGL_E_LAUNCH;                                         \
        }                                                                 \
    } while (0)

extern "C" int msgl_ar_create(int rank, int world, size_t max_bytes, msgl_ar_t** out) {
    MSGL_REQUIRE(out, "msgl_ar_create: null output");
    *out = nullptr;
    MSGL_REQUIRE(world >= 2 && world <= MSGL_AR_MAX_WORLD && rank >= 0 && rank < world, "msgl_ar_create: rank %d of %d", rank, world);
    MSGL_REQUIRE(max_bytes > 0 && max_bytes % (16 * (size_t)world) == 0, "msgl_ar_create: max_bytes %zu must be a positive multiple of %d",
                 max_bytes, 16 * world);
    msgl_ar* c = new msgl_ar();
    c->rank = rank, c->world = world, c->max_bytes = max_bytes, c->opened = false;
    hipError_t err = hipMalloc(&c->data, 4 * max_bytes);
    if (err == hipSuccess) err = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->flags), kFlagBytes, hipDeviceMallocUncached);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&c->state), (kArMaxBlocks + 1) * sizeof(uint32_t));
    if (err == hipSuccess) err = hipMemset(c->flags, 0, kFlagBytes);
    if (err == hipSuccess) err = hipMemset(c->state, 0, (kArMaxBlocks + 1) * sizeof(uint32_t));
    if (err == hipSuccess) err = hipDeviceSynchronize();
    if (err != hipSuccess) {
        set_error("msgl_ar_create: %s", hipGetErrorString(err));
        if (c->data) (void)hipFree(c->data);
        if (c->flags) (void)hipFree(c->flags);
        if (c->state) (void)hipFree(c->state);
        delete c;
        return MSGL_E_LAUNCH;
    }
    for (int r = 0; r < MSGL_AR_MAX_WORLD; ++r) c->peer_data[r] = nullptr, c->peer_flags[r] = nullptr;
    c->peer_data[rank] = c->data;
    c->peer_flags[rank] = c->flags;
    *out = c;
    return MSGL_OK;
}

extern "C" int msgl_ar_export(msgl_ar_t* c, void* handles_out) {
    MSGL_REQUIRE(c && handles_out, "msgl_ar_export: null argument");
    static_assert(2 * sizeof(hipIpcMemHandle_t) == MSGL_AR_EXPORT_BYTES, "export size");
    hipIpcMemHandle_t h[2];
    MSGL_HIP_OK(hipIpcGetMemHandle(&h[0], c->data), "msgl_ar_export (staging)");
    MSGL_HIP_OK(hipIpcGetMemHandle(&h[1], c->flags), "msgl_ar_export (flags)");
    memcpy(handles_out, h, sizeof(h));
    return MSGL_OK;
}

extern "C" int msgl_ar_open(msgl_ar_t* c, const void* all_handles) {
    MSGL_REQUIRE(c && all_handles, "msgl_ar_open: null argument");
    MSGL_REQUIRE(!c->opened, "msgl_ar_open: already open");
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        hipIpcMemHandle_t h[2];
        memcpy(h, static_cast<const char*>(all_handles) + (size_t)r * MSGL_AR_EXPORT_BYTES, sizeof(h));
        MSGL_HIP_OK(hipIpcOpenMemHandle(&c->peer_data[r], h[0], hipIpcMemLazyEnablePeerAccess), "msgl_ar_open (staging)");
        MSGL_HIP_OK(hipIpcOpenMemHandle(&c->peer_flags[r], h[1], hipIpcMemLazyEnablePeerAccess), "msgl_ar_open (flags)");
    }
    c->opened = true;
    return MSGL_OK;
}

extern "C" int msgl_ar_allreduce_bf16(msgl_ar_t* c, const void* in, void* out, int64_t numel, size_t one_shot_bytes, void* stream) {
    MSGL_REQUIRE(c && c->opened, "msgl_ar_allreduce_bf16: communicator not open");
    if (numel == 0) return MSGL_OK;
    MSGL_REQUIRE(in && out && aligned16(in) && aligned16(out), "msgl_ar_allreduce_bf16: in / out must be 16-byte aligned device pointers");
    MSGL_REQUIRE(numel > 0 && numel % 8 == 0, "msgl_ar_allreduce_bf16: numel %lld must be a multiple of 8", (long long)numel);
    MSGL_REQUIRE((size_t)numel * 2 <= c->max_bytes, "msgl_ar_allreduce_bf16: %lld bytes exceed the communicator's %zu", (long long)numel * 2,
                 c->max_bytes);
    ArParams p;
    for (int r = 0; r < MSGL_AR_MAX_WORLD; ++r) {
        p.peer_data[r] = static_cast<const uint4*>(c->peer_data[r]);
        p.peer_flags[r] = static_cast<uint32_t*>(c->peer_flags[r]);
    }
    p.epoch = c->state;
    p.error = c->state + kArMaxBlocks;
    p.in = static_cast<const uint4*>(in);
    p.out = static_cast<uint4*>(out);
    p.nvec = numel / 8;
    p.area_vec = (int64_t)(c->max_bytes / 16);
    p.rank = c->rank, p.world = c->world;
    p.chunk_vec = (p.nvec + c->world - 1) / c->world;
    // two ranks: the one-step form moves the same bytes per link with one synchronisation less
    if (one_shot_bytes == 0) one_shot_bytes = c->world == 2 ? c->max_bytes : (size_t)256 << 10;
    const bool one_shot = (size_t)numel * 2 <= one_shot_bytes;
    const int64_t per_group = one_shot ? p.nvec : p.chunk_vec;
    const int64_t want = (per_group + kArThreads - 1) / kArThreads;
    const int cap = one_shot ? 32 : 64;
    const int blocks = (int)(want < 1 ? 1 : (want > cap ? cap : want));
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (c->world) {
        case 2: launch_ar<2>(one_shot, blocks, st, p); break;
        case 4: launch_ar<4>(one_shot, blocks, st, p); break;
        case 8: launch_ar<8>(one_shot, blocks, st, p); break;
        default: launch_ar<0>(one_shot, blocks, st, p); break;
    }
    return check_launch(one_shot ? "ar_one_shot_kernel" : "ar_two_shot_kernel");
}

extern "C" int msgl_ar_check(msgl_ar_t* c) {
    MSGL_REQUIRE(c, "msgl_ar_check: null communicator");
    MSGL_HIP_OK(hipDeviceSynchronize(), "msgl_ar_check");
    uint32_t err = 0;
    MSGL_HIP_OK(hipMemcpy(&err, c->state + kArMaxBlocks, sizeof(err), hipMemcpyDeviceToHost), "msgl_ar_check");
    if (err != 0) {
        set_error("msgl_ar: a kernel of rank %d gave up waiting for a peer", c->rank);
        return MSGL_E_LAUNCH;
    }
    return MSGL_OK;
}

extern "C" int msgl_ar_destroy(msgl_ar_t* c) {
    if (!c) return MSGL_OK;
    (void)hipDeviceSynchronize();
    for (int r = 0; r < c->world; ++r) {
        if (r == c->rank) continue;
        if (c->peer_data[r]) (void)hipIpcCloseMemHandle(c->peer_data[r]);
        if (c->peer_flags[r]) (void)hipIpcCloseMemHandle(c->peer_flags[r]);
    }
    (void)hipFree(c->data);
    (void)hipFree(c->flags);
    (void)hipFree(c->state);
    delete c;
    return MSGL_OK;
}
