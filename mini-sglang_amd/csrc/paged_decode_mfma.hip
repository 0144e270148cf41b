// Paged decode attention for GQA groups of 8 query heads per kv head (Qwen3-32B) on the matrix cores.
//
// The gather kernel of paged_decode.hip does its arithmetic on the vector pipe, and with 8 query heads per
// K/V row that arithmetic (384 multiply-adds and dot products out of ~520 vector instructions per 16 tokens and
// wave) is what the kernel waits for, not HBM.  Here the same work items (one work group per (request, split), all
// local kv heads, a wave per kv head or several waves per head taking every n-th tile) feed 16-token tiles to
// v_mfma_f32_16x16x32_bf16 (S^T = K.Q^T, the 8 query heads padded to 16 columns) and v_mfma_f32_16x16x16_bf16
// (O^T += V^T.P^T); the matrix pipe is then busy a few percent of the time and the kernel is paced by its loads.
//
//   * K and V rows go from HBM straight into LDS (global_load_lds_dwordx4: no staging registers, no ds_write),
//     into a ring of DEP 8 KiB tile slots PER WAVE: tiles are private to a wave, so the loop has no barrier, only
//     counted s_waitcnt vmcnt.  One instruction moves 4 rows of 256 B; K(n+DEP) is issued as soon as the K fragments
//     of tile n are in registers, V(n+DEP) likewise, so each wave keeps about DEP tiles in flight: 16 KiB with the
//     two slots of the eight-wave groups, 32 KiB with the four slots of the four-wave groups (128 KiB per CU either way).
//   * The LDS destination of such a load is lane-linear, the source address is per lane: the XOR swizzles that make
//     the fragment reads conflict free are applied to the SOURCE (lane p of a row fetches chunk p ^ f(row)).
//       K rows are read as MFMA A operands with ds_read_b128 (row = token, 16-byte chunk = 8 dims): f(row) = row;
//       V rows are read transposed with ds_read_b64_tr_b16 (lane <- one dim of 4 tokens):           f(row) = 2 (row & 7).
//   * S^T comes out with the query head on the lane (lane & 15) and 4 tokens in registers, so a head's row maximum
//     is 3 in-lane max + two lane swaps, the exponentiated scores packed to bf16 ARE the B operand of the second
//     product, and the O^T accumulator (8 x 4 registers) has the head on the lane as well: rescaling is in-lane.
//   * Requires page_size % 16 == 0 (a tile's 16 token rows are consecutive slots, one scalar table read per tile).
// Split-K, partials, launch plans, start offsets and the merge kernels are those of paged_decode.hip.
#include <type_traits>

#include "msgl_common.h"
#include "decode_params.h"

namespace msgl {

typedef __attribute__((ext_vector_type(8))) __bf16 mbf16x8_t;
typedef __attribute__((ext_vector_type(4))) short mshort4_t;
typedef __attribute__((ext_vector_type(4))) float mf32x4_t;

// Diagnostic builds only (-DDEC_STAMPS, scripts/stamp_decode.sh): wave 0 of every work group leaves the value of the
// chip-wide 100 MHz counter at six points of its life; tools/decode_bench prints where a launch's time goes.
#ifdef DEC_STAMPS
__device__ unsigned long long g_dec_stamps[8192][6];
#define DSTAMP(i)                                                                                       \
    do {                                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 8192) g_dec_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memrealtime(); \
    } while (0)
#else
#define DSTAMP(i)
#endif

constexpr int kMHalf = 4096;            // bytes of a K or V tile (16 rows of 256 B)
// A wave's ring: DEP K tiles, then DEP V tiles (one of each being read, the others on their way).
constexpr int wave_ring_bytes(int dep) { return 2 * dep * kMHalf; }

// Four LDS-DMA loads of 4 rows x 256 B each into 4 KiB at LDS byte address `dst`; lane offsets off[i] (bytes from
// `base`) select the row and the swizzled chunk.  The statement first waits for the wave's outstanding LDS reads (the
// fragments of the tile that occupied these bytes).  `base` was just written by scalar arithmetic: s_nop 4 before the
// first load reads it; M0 is written and read inside the statement.
template <bool NT>
__device__ __forceinline__ void dma_rows16(const void* base, uint32_t dst, const uint32_t (&off)[4]) {
    unsigned keep;
    if (NT)
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_nop 4\n\t"
            "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1 nt\n\t"
            "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1 nt\n\t"
            "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1 nt\n\t"
            "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1 nt\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(base), "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(dst), "s"(dst + 1024), "s"(dst + 2048), "s"(dst + 3072)
            : "memory");
    else
        asm volatile(
            "s_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_nop 4\n\t"
            "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
            "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
            "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
            "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "s"(base), "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(dst), "s"(dst + 1024), "s"(dst + 2048), "s"(dst + 3072)
            : "memory");
}

// wait until at most `groups` of the wave's four-load groups are outstanding (loads complete in issue order)
__device__ __forceinline__ void wait_dma_groups(int groups) {
    switch (groups) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(28)" ::: "memory"); break;
    }
}

// max over the four lanes that hold the same query head (lane, lane ^ 16, lane ^ 32, lane ^ 48)
__device__ __forceinline__ float max_over_lane_groups(float x) {
    unsigned u = __builtin_bit_cast(unsigned, x);
    // (results through scalars: a bit cast applied straight to an element of the result vector reads element 0
    // whichever element is named)
    const auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    const unsigned a0 = a[0], a1 = a[1];
    x = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
    u = __builtin_bit_cast(unsigned, x);
    const auto b = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    const unsigned b0 = b[0], b1 = b[1];
    return fmaxf(__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1));
}

// G = query heads per kv head (any count up to 16 fits the 16 padded columns; built for 8 and, for ranks with one or two
// kv heads, 4), HPB = local kv heads.
template <int G, int HPB, int NW = kWaves, int DEP = 2>
__global__ __launch_bounds__(NW * 64) void paged_decode_mfma_kernel(DecodeParams p) {
    static_assert(G <= 16, "the query heads of a kv head are the 16 columns of S^T");
    static_assert(DEP >= 2 && DEP <= 4, "at most 2 * DEP - 1 = 7 load groups outstanding (wait_dma_groups)");
    constexpr int kMWaveBytes = wave_ring_bytes(DEP);
    DSTAMP(0);
    constexpr int WPH = NW / HPB;  // waves sharing one kv head
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    int item = blockIdx.x;
    if (p.plan) {
        item = p.plan[blockIdx.x];
        if (item < 0) return;
    }
    const int split = item / p.bs;
    const int b = item - split * p.bs;
    const int chunk = split_chunk(p);
    // (the request's table row is asked for together with its lengths, ahead of the early exit below: one round trip
    // to memory less in front of the first page-table read)
    const int req_row = p.req_idx[b];
    const SplitInfo si = split_info(p, b, chunk);
    asm volatile("" ::"s"(req_row));
    const int S = si.S, nsplit = si.nsplit;
    if (split >= nsplit) return;
    const int t0 = si.start + split * chunk;  // a multiple of 16: tiles never straddle a 16-slot run
    const int t1 = (split == p.max_splits - 1) ? S : min(S, t0 + chunk);
    const int ntiles = (t1 - t0 + kTile - 1) / kTile;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g4 = lane >> 4;   // S^T / O^T: which 4 rows of a 16-row block; LDS-DMA: which of an instruction's 4 rows
    const int qn = lane & 15;   // S^T / O^T: query head (8 .. 15 are padding); LDS-DMA: 16-byte position inside the row
    const int h = wave / WPH;
    const int sub = wave - h * WPH;
    const int my_tiles = (ntiles > sub) ? (ntiles - sub + WPH - 1) / WPH : 0;  // own tiles: sub, sub + WPH, ...

    const int32_t* r2t_row = p.r2t + (int64_t)req_row * p.r2t_stride;
    // LDS byte address of this wave's ring (what M0 takes)
    const uint32_t ring = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (uint32_t)wave * kMWaveBytes;
    const char* ring_ptr = smem + wave * kMWaveBytes;  // [DEP K tiles][DEP V tiles]

    // Q^T fragments (B operand of S^T = K.Q^T): lane (head qn, group g4) holds dims 32 ks + 8 g4 .. + 8.  Loaded by
    // statements the compiler does not count, ahead of the first LDS-DMA loads and without waiting for them there:
    // vector loads return in issue order, so once the first K tile has landed (the counted wait in front of the loop) so
    // has the query; a load the compiler counts would cost a round trip to memory before the first K/V load, or a full
    // vmcnt(0) inside the loop.  Padding heads (qn >= G) read head 0's row: their columns are never stored.
    typedef __attribute__((ext_vector_type(4))) unsigned int mu32x4_t;
    mu32x4_t qraw[4];
    {
        const uint16_t* qp = p.q + (int64_t)b * p.q_stride + (h * G + (qn < G ? qn : 0)) * kD + g4 * 8;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(qraw[ks]) : "v"(qp + ks * 32) : "memory");
    }

    // source offsets of the LDS-DMA loads: instruction i brings rows 4 i + g4 of the tile; position qn of row r gets chunk
    // qn ^ r (K) or qn ^ 2 (r & 7) (V).  Bytes from the tile's first row of this kv head.
    uint32_t koff[4], voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = 4 * i + g4;
        koff[i] = (uint32_t)(r * p.cache_stride * 2 + ((qn ^ r) << 4));
        voff[i] = (uint32_t)(r * p.cache_stride * 2 + ((qn ^ (2 * (r & 7))) << 4));
    }
    // LDS offsets of the fragment reads inside a tile slot.
    // K, k-step ks: row qn, chunk (4 ks + g4) ^ qn.
    uint32_t kread[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) kread[ks] = (uint32_t)(qn * 256 + (((4 * ks + g4) ^ qn) << 4));
    // V, block mb of 16 dims: lane i = qn of a 16-lane group addresses row 4 g4 + (i >> 2), 8-byte segment i & 3 of the
    // block's 32 bytes, and receives dim 16 mb + i of the group's 4 rows (tokens 4 g4 .. 4 g4 + 3).
    const int vrow = 4 * g4 + (qn >> 2), vseg = qn & 3;
    uint32_t vread[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb)
        vread[mb] = (uint32_t)(vrow * 256 + (((2 * mb + (vseg >> 1)) ^ (2 * (vrow & 7))) << 4) + (vseg & 1) * 8);

    mf32x4_t o[8];
#pragma unroll
    for (int mb = 0; mb < 8; ++mb) o[mb] = mf32x4_t{0.f, 0.f, 0.f, 0.f};
    float m = kNegBig, l = 0.f;  // running max (log2 units) of this lane's head; this lane's share of the running sum

    const int reuse_len = !p.kv_stream ? 0x7fffffff : (p.reuse_lens ? p.reuse_lens[b] : 0);
    // Page-table entry of own tile n (the slot of its first token; the other 15 follow it), by a scalar load that the
    // compiler does not count: a vector load here would make it wait for "all" vector memory operations, the LDS-DMA
    // loads in flight included.  Tiles past the wave's last one read the last one's entry.  table_wait makes the
    // value usable (the load and its wait are separate statements so that other work can sit between them).
    auto table_load = [&](int n) -> int {
        const int tok = t0 + (sub + min(n, my_tiles - 1) * WPH) * kTile;
        int idx;
        asm volatile("s_load_dword %0, %1, %2" : "=s"(idx) : "s"(r2t_row), "s"(tok * 4) : "memory");
        return idx;
    };
    auto table_wait = [&](int& idx) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(idx)::"memory"); };
    auto issue_k = [&](int n, int idx) {
        const uint16_t* kb = p.kc + ((int64_t)idx * p.cache_stride + h * kD);
        const uint32_t dst = ring + (uint32_t)(n % DEP) * kMHalf;
        if (t0 + (sub + n * WPH) * kTile >= reuse_len) dma_rows16<true>(kb, dst, koff);
        else dma_rows16<false>(kb, dst, koff);
    };
    auto issue_v = [&](int n, int idx) {
        const uint16_t* vb = p.vc + ((int64_t)idx * p.cache_stride + h * kD);
        const uint32_t dst = ring + (uint32_t)(DEP + n % DEP) * kMHalf;
        if (t0 + (sub + n * WPH) * kTile >= reuse_len) dma_rows16<true>(vb, dst, voff);
        else dma_rows16<false>(vb, dst, voff);
    };
    // Issue order of the wave's load groups (4 loads each): K0 V0 .. K(DEP-1) V(DEP-1) up front, then tile n issues
    // K(n+DEP) once its K fragments are in registers and V(n+DEP) once its V fragments are: the sequence is K0 V0 K1 V1
    // ... whatever T is, K(j) at position 2 j and V(j) at 2 j + 1.  Loads complete in that order, so "group at position
    // x has landed" is "at most (groups issued so far - 1 - x) groups outstanding".  When tile n starts, every group of
    // the tiles below min(T, n + DEP) has been issued.
    const int T = my_tiles;
    if (T > 0) {
        int first[DEP];
#pragma unroll
        for (int j = 0; j < DEP; ++j) first[j] = table_load(j);
        int idx_ahead = table_load(DEP);  // entry of tile n + DEP while tile n is worked on
#pragma unroll
        for (int j = 0; j < DEP; ++j) table_wait(first[j]);
        table_wait(idx_ahead);
        DSTAMP(1);
#pragma unroll
        for (int j = 0; j < DEP; ++j)
            if (j < T) issue_k(j, first[j]), issue_v(j, first[j]);
        // K(0) has landed: so has the query (see above); from here on the compiler may use it
        wait_dma_groups(2 * min(T, DEP) - 1);
        DSTAMP(2);
        uint4 qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            asm volatile("" : "+v"(qraw[ks]));
            qf[ks] = make_uint4(qraw[ks].x, qraw[ks].y, qraw[ks].z, qraw[ks].w);
        }
        for (int n = 0; n < T; ++n) {
            const char* kslot = ring_ptr + (n % DEP) * kMHalf;
            const char* vslot = ring_ptr + (DEP + n % DEP) * kMHalf;
            const int issued = 2 * min(T, n + DEP);
            const bool more = n + DEP < T;
            wait_dma_groups(issued - 1 - 2 * n);
            uint4 kf[4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) kf[ks] = *reinterpret_cast<const uint4*>(kslot + kread[ks]);
            if (more) issue_k(n + DEP, idx_ahead);  // (waits for the four reads above first)
            int idx_after = table_load(n + DEP + 1);  // used by the next iteration; waited for at the end of this one
            mf32x4_t s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mbf16x8_t, kf[ks]), __builtin_bit_cast(mbf16x8_t, qf[ks]), s, 0, 0, 0);
            // scores of tokens tok0 + r for head qn
            const int tok0 = t0 + (sub + n * WPH) * kTile + 4 * g4;
            float sc[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) sc[r] = (tok0 + r < t1) ? s[r] * p.scale_log2e : -INFINITY;
            const float mt = max_over_lane_groups(fmaxf(fmaxf(sc[0], sc[1]), fmaxf(sc[2], sc[3])));
            const float mn = fmaxf(m, mt);
            const float alpha = fast_exp2(m - mn);
            m = mn;
            float pr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[r] = fast_exp2(sc[r] - mn);
            l = l * alpha + ((pr[0] + pr[1]) + (pr[2] + pr[3]));
            if (!__all(alpha == 1.0f)) {  // the running max settles after a few tiles
#pragma unroll
                for (int mb = 0; mb < 8; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[mb][r] *= alpha;
            }
            // P^T as the B operand of O^T += V^T.P^T: k = tokens 4 g4 + j, n = head qn
            const uint2 pp = make_uint2(pack_bf16(pr[0], pr[1]), pack_bf16(pr[2], pr[3]));
            const mshort4_t pfrag = __builtin_bit_cast(mshort4_t, pp);
            wait_dma_groups(issued + (more ? 1 : 0) - 1 - (2 * n + 1));
            mshort4_t vf[8];
#pragma unroll
            for (int mb = 0; mb < 8; ++mb)
                vf[mb] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mshort4_t*)(vslot + vread[mb]));
            if (more) issue_v(n + DEP, idx_ahead);
#pragma unroll
            for (int mb = 0; mb < 8; ++mb) o[mb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vf[mb], pfrag, o[mb], 0, 0, 0);
            table_wait(idx_after);
            idx_ahead = idx_after;
        }
    }

    DSTAMP(3);
    // this lane's head: total of the four lane groups' sums
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    // every wave is done with its ring (no load outstanding, all fragments in registers): the memory becomes the
    // exchange area of the merge
    wait_dma_groups(0);
    __syncthreads();
    float* lds_o = reinterpret_cast<float*>(smem);       // [NW][G][128]
    float* lds_ml = lds_o + NW * G * kD;                  // [NW][G][2]
    if (qn < G) {
        float* dst = lds_o + (wave * G + qn) * kD + 4 * g4;
#pragma unroll
        for (int mb = 0; mb < 8; ++mb) *reinterpret_cast<float4*>(dst + 16 * mb) = make_float4(o[mb][0], o[mb][1], o[mb][2], o[mb][3]);
        if (g4 == 0) {
            lds_ml[(wave * G + qn) * 2] = m;
            lds_ml[(wave * G + qn) * 2 + 1] = l;
        }
    }
    __syncthreads();
    DSTAMP(4);
    merge_waves_and_store<G, WPH>(p, lds_o, lds_ml, h, sub, lane, b, split, nsplit == 1 && si.extra == 0);
    DSTAMP(5);
}

template <int G, int HPB, int NW = kWaves, int DEP = 2>
static int launch_mfma(const DecodeParams& p, hipStream_t stream) {
    constexpr int kLds = NW * wave_ring_bytes(DEP);
    static PerDeviceOnce once;  // the rings need more than the 64 KiB a kernel gets by default
    if (!once.ensure([] {
            return hipFuncSetAttribute(reinterpret_cast<const void*>(paged_decode_mfma_kernel<G, HPB, NW, DEP>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kLds) == hipSuccess;
        })) {
        set_error("msgl_paged_decode: cannot reserve %d bytes of LDS", kLds);
        return MSGL_E_LAUNCH;
    }
    hipLaunchKernelGGL((paged_decode_mfma_kernel<G, HPB, NW, DEP>), dim3(p.bs * p.max_splits), dim3(NW * 64), kLds, stream, p);
    return check_launch("paged_decode_mfma_kernel");
}

// The main launch of a decode call on 16-slot runs through the matrix-core kernel; the caller launches the merge.
int launch_decode_mfma(const DecodeParams& p, int G, int Hkv, hipStream_t stream) {
    if (G == 8) {
        switch (Hkv) {
            // One or two kv heads per rank, launches of about one work group per CU (long contexts at small batch): 4 waves
            // per (request, split) instead of 8 -- half the LDS, so two groups fit a CU, and half the waves to merge at
            // the end.  Measured at 128k contexts: one kv head 7 % faster at batch 1, 6 % at 2, 1-3 % at 4-8, equal at 16;
            // two kv heads 6 % at batch 1, 2 % at 4, 3 % SLOWER at 16.  msgl_set_variant(MSGL_VARIANT_DECODE_MFMA_WAVES, 4 / 8)
            // forces either.
            case 1:
            case 2: {
                const int forced = variant(MSGL_VARIANT_DECODE_MFMA_WAVES);
                const int64_t grid = (int64_t)p.bs * p.max_splits, cus = device_cus();
                const bool four = forced > 0 ? forced == 4 : grid <= (Hkv == 1 ? 2 * cus : cus);
                // Four tile slots per wave instead of two (the whole 128 KiB of a CU's ring space in flight from one group)
                // for launches of at most one four-wave group per CU whose waves have at least 16 tiles each: measured
                // 3 % ahead at 8 requests x 131072 tokens in 32 splits; with 8 tiles per wave (one request in 256 splits)
                // the deeper start-up burst delays the first tile by 2 us and the launch ends 1.5 us later.  A split size
                // only known on the device (captured launches) takes two slots.
                const int depth = variant(MSGL_VARIANT_DECODE_MFMA_DEPTH);
                const bool deep = four && (depth > 0 ? depth == 4 : (grid <= cus && p.chunk_dev == nullptr && p.chunk >= 16 * kTile * (4 / Hkv)));
                if (Hkv == 1) return deep ? launch_mfma<8, 1, 4, 4>(p, stream) : four ? launch_mfma<8, 1, 4>(p, stream) : launch_mfma<8, 1>(p, stream);
                return deep ? launch_mfma<8, 2, 4, 4>(p, stream) : four ? launch_mfma<8, 2, 4>(p, stream) : launch_mfma<8, 2>(p, stream);
            }
            case 4: return launch_mfma<8, 4>(p, stream);
            case 8: return launch_mfma<8, 8>(p, stream);
        }
    } else if (G == 4) {
        switch (Hkv) {
            case 1: {  // 4 waves per group: 1-2 % ahead of 8 at every shape tried
                return variant(MSGL_VARIANT_DECODE_MFMA_WAVES) == 8 ? launch_mfma<4, 1>(p, stream) : launch_mfma<4, 1, 4>(p, stream);
            }
            case 2: return launch_mfma<4, 2>(p, stream);
            case 4: return launch_mfma<4, 4>(p, stream);
            case 8: return launch_mfma<4, 8>(p, stream);
        }
    }
    set_error("msgl_paged_decode: matrix-core kernel has no build for %d query heads per kv head x %d kv heads", G, Hkv);
    return MSGL_E_INVALID;
}

}  // namespace msgl

#ifdef DEC_STAMPS
extern "C" void msgl_decode_stamps(unsigned long long* out, int groups) {
    (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(msgl::g_dec_stamps), sizeof(unsigned long long) * 6 * (groups < 8192 ? groups : 8192));
}
#endif
