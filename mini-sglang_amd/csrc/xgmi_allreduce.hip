// Sum all-reduce of bf16 activations across the GPUs of one node, for tensor-parallel decode
// steps: every rank maps its peers' staging areas (IPC) and reads them directly over xGMI.
// The links are point to point, so a rank pulls from its 7 peers over 7 links at once; there
// is no ring and no host involvement, and one kernel per call makes the call capturable.
//
// Memory per rank:
//   staging  [2 buffers][2 stages][max_bytes]   plain device memory, written only by its owner,
//            read by everybody; buffer = parity of the call number (see below), stage 0 = copy
//            of the input, stage 1 = this rank's reduced slice (two-step form);
//   flags    [2 sets][kArMaxBlocks][8 sources]  uncached device memory, written by the peers;
//            word (set, block, src) = number of the last call in which block `block` of rank
//            `src` has passed that synchronisation point;
//   state    [0] calls completed on this communicator, [1] work groups that have finished the
//            call in progress, [2] error word.
// The call number is ONE counter per communicator: every work group of a call reads it when it
// starts and the last one to finish advances it, so flag values and buffer parity are the same
// for all work groups whatever the grid sizes of earlier calls were.
// Work groups synchronise pairwise: group b of every rank touches the same element set, so
// it only has to wait for group b of its peers (no grid-wide step).  A group publishes its
// writes with a system-scope release store into each peer's flag word; the reader polls its own
// words and then takes a system-scope acquire before it reads the peers' memory.
// Two buffers suffice without a trailing synchronisation.  Calls of one communicator are ordered
// on their stream, so rank A starts call n+1 only after ALL its groups have finished call n.
// Group 0 exists in every call; in call n it waited for group 0 of every peer B to arrive in
// call n, and B's group 0 started call n only after all of B's groups had finished call n-1,
// reads included.  So when A overwrites buffer (n+1) & 1 in call n+1, nobody still reads what
// call n-1 left there — for any grid sizes and either form in calls n-1, n, n+1.
//
// A call may also be issued step by step (msgl_ar_allreduce_bf16_step): the same device code,
// cut at its synchronisation points into launches of their own, for set-ups in which the ranks'
// kernels cannot be relied upon to run at the same time (several ranks on ONE device: a kernel
// that spins there may keep the peer's kernel from ever starting).  The caller crosses each
// synchronisation point on the host; the waits then find the flags already raised.
#include <string.h>

#include "../../include/msgl_comm.h"
#include "msgl_common.h"

namespace msgl {

constexpr int kArThreads = 512;
constexpr int kArMaxBlocks = 128;
constexpr long long kArTicksPerMs = 100000LL;  // wall_clock64 runs at 100 MHz
constexpr int kArDefaultTimeoutMs = 1000;

// steps of a call; a fused launch runs all of its form's steps
constexpr int kStepPublish = 1;  // copy the input into the staging area, tell the peers
constexpr int kStepReduce = 2;   // wait for their copies; one step: sum them all; two steps: sum this rank's slice, tell the peers
constexpr int kStepCollect = 4;  // two steps only: wait for the peers' slices, collect them

struct ArParams {
    const uint4* peer_data[MSGL_AR_MAX_WORLD];  // staging areas in rank order (own one included)
    uint32_t* peer_flags[MSGL_AR_MAX_WORLD];
    uint32_t* state;   // [0] calls completed, [1] groups done with the current call, [2] error (1 = a wait timed out)
    const uint4* in;
    uint4* out;
    int64_t nvec;       // 16-byte vectors in the message
    int64_t chunk_vec;  // two-step form: vectors per rank slice
    int64_t area_vec;   // vectors per (buffer, stage) area
    long long timeout_ticks;
    int rank, world;
    int steps;          // kStep* bits this launch runs
    int last_step;      // the kStep* bit that completes a call of this form
};

// Tell every peer that this group's writes up to here are visible.
__device__ __forceinline__ void signal_peers(const ArParams& p, int set, uint32_t e) {
    __syncthreads();  // every wave's stores are issued and counted before the release below
    const int t = threadIdx.x;
    if (t < p.world && t != p.rank) {
        const int slot = (set * kArMaxBlocks + blockIdx.x) * MSGL_AR_MAX_WORLD;
        // release fence, then a wait the compiler cannot drop (it omits the one after the write-back when it has
        // just waited for the wave's own stores, and the flag could overtake the write-back), then the flag
        __scoped_atomic_thread_fence(__ATOMIC_RELEASE, __MEMORY_SCOPE_SYSTEM);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __scoped_atomic_store_n(p.peer_flags[t] + slot + p.rank, e, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM);
    }
}

// Wait until every peer's group has signalled call e at this point.  False after a timeout (the
// caller skips its reads; the error word is set).
__device__ __forceinline__ bool wait_peers(const ArParams& p, int set, uint32_t e, int* s_ok) {
    __syncthreads();  // nobody is still reading the verdict of an earlier wait
    if (threadIdx.x == 0) *s_ok = 1;
    __syncthreads();
    const int t = threadIdx.x;
    if (t < p.world && t != p.rank) {
        const uint32_t* mine = p.peer_flags[p.rank] + (set * kArMaxBlocks + blockIdx.x) * MSGL_AR_MAX_WORLD + t;
        const long long t0 = wall_clock64();
        while ((int32_t)(__scoped_atomic_load_n(mine, __ATOMIC_RELAXED, __MEMORY_SCOPE_SYSTEM) - e) < 0) {
            if (wall_clock64() - t0 > p.timeout_ticks) {
                *s_ok = 0;
                atomicExch(p.state + 2, 1u);
                break;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        __scoped_atomic_thread_fence(__ATOMIC_ACQUIRE, __MEMORY_SCOPE_SYSTEM);  // once, after the polling
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate has completed before the barrier lets the others read
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

// Number of the call this launch belongs to: one more than the calls completed.
__device__ __forceinline__ uint32_t begin_call(const ArParams& p, uint32_t* s_epoch) {
    if (threadIdx.x == 0) *s_epoch = __hip_atomic_load(p.state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
    __syncthreads();
    return *s_epoch;
}

// The launch that runs a call's last step counts its groups out; the last one advances the call number.
__device__ __forceinline__ void end_call(const ArParams& p, uint32_t e) {
    if (!(p.steps & p.last_step) || threadIdx.x != 0) return;
    if (atomicAdd(p.state + 1, 1u) == gridDim.x - 1) {
        atomicExch(p.state + 1, 0u);
        __hip_atomic_store(p.state, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
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
    if (p.steps & kStepPublish) {
        for (int64_t i = first; i < p.nvec; i += stride) mine[i] = p.in[i];
        signal_peers(p, 0, e);
    }
    if (p.steps & kStepReduce) {
        if (wait_peers(p, 0, e, &s_ok))
            for (int64_t i = first; i < p.nvec; i += stride) p.out[i] = sum_ranks<W>(p, stage0, i);
    }
    end_call(p, e);
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
    if (p.steps & kStepPublish) {
        for (int64_t j = first; j < cv; j += stride) {
            constexpr int N = W ? W : MSGL_AR_MAX_WORLD;
            uint4 v[N];
#pragma unroll
            for (int c = 0; c < N; ++c)
                if (c < world && c * cv + j < p.nvec) v[c] = p.in[c * cv + j];
#pragma unroll
            for (int c = 0; c < N; ++c)
                if (c < world && c * cv + j < p.nvec) mine[stage0 + c * cv + j] = v[c];
        }
        signal_peers(p, 0, e);
    }
    if (p.steps & kStepReduce) {
        // after a timeout nothing is signalled: the peers' waits run out too and every rank's error word is set
        if (wait_peers(p, 0, e, &s_ok)) {
            for (int64_t j = first; j < cv; j += stride) {
                const int64_t i = p.rank * cv + j;
                if (i < p.nvec) mine[stage1 + j] = sum_ranks<W>(p, stage0, i);
            }
            signal_peers(p, 1, e);
        }
    }
    if (p.steps & kStepCollect) {
        if (wait_peers(p, 1, e, &s_ok))
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
    }
    end_call(p, e);
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
    int timeout_ms;
    void* data;
    uint32_t* flags;
    uint32_t* state;  // device: calls completed, groups done with the current call, error word
    void* peer_data[MSGL_AR_MAX_WORLD];
    void* peer_flags[MSGL_AR_MAX_WORLD];
    bool opened;
};

static constexpr size_t kFlagBytes = 2 * (size_t)kArMaxBlocks * MSGL_AR_MAX_WORLD * sizeof(uint32_t);
static constexpr size_t kStateBytes = 4 * sizeof(uint32_t);

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
    c->rank = rank, c->world = world, c->max_bytes = max_bytes, c->opened = false, c->timeout_ms = kArDefaultTimeoutMs;
    hipError_t err = hipMalloc(&c->data, 4 * max_bytes);
    if (err == hipSuccess) err = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->flags), kFlagBytes, hipDeviceMallocUncached);
    if (err == hipSuccess) err = hipMalloc(reinterpret_cast<void**>(&c->state), kStateBytes);
    if (err == hipSuccess) err = hipMemset(c->flags, 0, kFlagBytes);
    if (err == hipSuccess) err = hipMemset(c->state, 0, kStateBytes);
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

extern "C" int msgl_ar_set_timeout_ms(msgl_ar_t* c, int milliseconds) {
    MSGL_REQUIRE(c, "msgl_ar_set_timeout_ms: null communicator");
    MSGL_REQUIRE(milliseconds >= 1 && milliseconds <= 60000, "msgl_ar_set_timeout_ms: %d ms outside 1 .. 60000", milliseconds);
    c->timeout_ms = milliseconds;
    return MSGL_OK;
}

static bool ar_one_shot(const msgl_ar* c, int64_t numel, size_t one_shot_bytes) {
    // two ranks: the one-step form moves the same bytes per link with one synchronisation less
    if (one_shot_bytes == 0) one_shot_bytes = c->world == 2 ? c->max_bytes : (size_t)256 << 10;
    return (size_t)numel * 2 <= one_shot_bytes;
}

extern "C" int msgl_ar_num_steps(msgl_ar_t* c, int64_t numel, size_t one_shot_bytes) {
    if (!c || numel <= 0) return 0;
    return ar_one_shot(c, numel, one_shot_bytes) ? 2 : 3;
}

// steps: kStep* bits to run in this launch (all of the form's steps = the fused call)
static int ar_launch(msgl_ar_t* c, const void* in, void* out, int64_t numel, size_t one_shot_bytes, int step, void* stream, const char* who) {
    MSGL_REQUIRE(c && c->opened, "%s: communicator not open", who);
    if (numel == 0) return MSGL_OK;
    MSGL_REQUIRE(in && out && aligned16(in) && aligned16(out), "%s: in / out must be 16-byte aligned device pointers", who);
    MSGL_REQUIRE(numel > 0 && numel % 8 == 0, "%s: numel %lld must be a multiple of 8", who, (long long)numel);
    MSGL_REQUIRE((size_t)numel * 2 <= c->max_bytes, "%s: %lld bytes exceed the communicator's %zu", who, (long long)numel * 2, c->max_bytes);
    const bool one_shot = ar_one_shot(c, numel, one_shot_bytes);
    const int nsteps = one_shot ? 2 : 3;
    MSGL_REQUIRE(step >= -1 && step < nsteps, "%s: step %d of a %d-step call", who, step, nsteps);
    ArParams p;
    for (int r = 0; r < MSGL_AR_MAX_WORLD; ++r) {
        p.peer_data[r] = static_cast<const uint4*>(c->peer_data[r]);
        p.peer_flags[r] = static_cast<uint32_t*>(c->peer_flags[r]);
    }
    p.state = c->state;
    p.in = static_cast<const uint4*>(in);
    p.out = static_cast<uint4*>(out);
    p.nvec = numel / 8;
    p.area_vec = (int64_t)(c->max_bytes / 16);
    p.rank = c->rank, p.world = c->world;
    p.chunk_vec = (p.nvec + c->world - 1) / c->world;
    p.timeout_ticks = (long long)c->timeout_ms * kArTicksPerMs;
    p.last_step = one_shot ? kStepReduce : kStepCollect;
    p.steps = step < 0 ? (one_shot ? kStepPublish | kStepReduce : kStepPublish | kStepReduce | kStepCollect) : 1 << step;
    const int64_t per_group = one_shot ? p.nvec : p.chunk_vec;
    const int64_t want = (per_group + kArThreads - 1) / kArThreads;
    const int cap = one_shot ? 32 : 64;
    static_assert(64 <= kArMaxBlocks, "flag words per set");
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

extern "C" int msgl_ar_allreduce_bf16(msgl_ar_t* c, const void* in, void* out, int64_t numel, size_t one_shot_bytes, void* stream) {
    return ar_launch(c, in, out, numel, one_shot_bytes, -1, stream, "msgl_ar_allreduce_bf16");
}

extern "C" int msgl_ar_allreduce_bf16_step(msgl_ar_t* c, const void* in, void* out, int64_t numel, size_t one_shot_bytes, int step,
                                           void* stream) {
    MSGL_REQUIRE(step >= 0, "msgl_ar_allreduce_bf16_step: step %d", step);
    return ar_launch(c, in, out, numel, one_shot_bytes, step, stream, "msgl_ar_allreduce_bf16_step");
}

extern "C" int msgl_ar_check(msgl_ar_t* c) {
    MSGL_REQUIRE(c, "msgl_ar_check: null communicator");
    MSGL_HIP_OK(hipDeviceSynchronize(), "msgl_ar_check");
    uint32_t st[3] = {0, 0, 0};
    MSGL_HIP_OK(hipMemcpy(st, c->state, sizeof(st), hipMemcpyDeviceToHost), "msgl_ar_check");
    if (st[2] != 0) {
        set_error("msgl_ar: a kernel of rank %d gave up waiting for a peer after %d ms (%u calls completed)", c->rank, c->timeout_ms, st[0]);
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
