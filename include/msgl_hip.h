/*
 * msgl_hip.h — C ABI of libmsgl_hip.so: the MI355X (gfx950) kernels behind
 * mini-sglang's attention-backend interface and its RMSNorm / SiluAndMul /
 * RotaryEmbedding layers.
 *
 * Conventions (all entry points):
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch-allocated
 *     tensors in the host mirror); the library never allocates or frees device
 *     memory and keeps no state besides a thread-local error string (and the
 *     test-only overrides of msgl_set_variant, which default to "library chooses");
 *   - activations / caches are bf16 (2-byte elements), index tensors int32,
 *     strides are in ELEMENTS of the tensor they describe;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     work is only enqueued, never synchronised, so every call is legal under
 *     hipGraph stream capture;
 *   - return value 0 = success; negative = error (MSGL_E_*), with a message
 *     retrievable through msgl_last_error().  The Python mirror raises
 *     RuntimeError(msgl_last_error()) on any non-zero return, which matches the
 *     reference's behaviour of letting a kernel error kill the scheduler rank
 *     (reference: mini_sglang/managers/scheduler.py:559-567).
 *
 * The reference is pure Python; the interface each entry point replaces is the
 * Python call named in its comment (paths relative to the reference root).
 */
#ifndef MSGL_HIP_H
#define MSGL_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSGL_OK 0
#define MSGL_E_INVALID (-1)   /* bad argument (shape, alignment, unsupported size) */
#define MSGL_E_LAUNCH (-2)    /* hipLaunchKernel reported an error */
#define MSGL_E_WORKSPACE (-3) /* workspace too small */

/* ABI version of this header; bumped when a signature changes. */
int msgl_abi_version(void);

/* Last error message of the calling thread ("" if none). */
const char* msgl_last_error(void);

/* ---------------------------------------------------------------------------
 * Kernel-form overrides for tests and measurements (no counterpart in the reference).
 * Several entry points choose between kernel forms that compute the same result
 * (to rounding); by default the library chooses from the call's shape alone and
 * reads nothing from its environment.  msgl_set_variant pins one choice process-wide
 * until it is set back to MSGL_VARIANT_AUTO; a serving process never calls it.
 * Returns the previous value, or MSGL_E_INVALID for an unknown `which`.
 *   MSGL_VARIANT_DECODE_MFMA      0 = gather kernel only, 8 = matrix-core kernel only for 8 heads
 *                                 per kv head, 4 = also for 4 heads per kv head on every shape
 *   MSGL_VARIANT_DECODE_MFMA_WAVES  4 / 8 waves per (request, split) of the matrix-core decode kernel
 *   MSGL_VARIANT_DECODE_MFMA_DEPTH  2 / 4 tile slots per wave of its four-wave form
 *   MSGL_VARIANT_EXTEND_HEADS     1 / 2 query heads per extend work group
 *   MSGL_VARIANT_EXTEND_WIDE      0 = eight waves (one head each), 1 = four waves (both heads each)
 *   MSGL_VARIANT_EXTEND_RING      2 / 3 slots in the K/V tile ring of the eight-wave extend kernel
 *   MSGL_VARIANT_EXTEND_KEY_SPLIT 1..8 key runs per extend work item (needs the workspace)
 * ------------------------------------------------------------------------- */
#define MSGL_VARIANT_AUTO (-1)
#define MSGL_VARIANT_DECODE_MFMA 0
#define MSGL_VARIANT_DECODE_MFMA_WAVES 1
#define MSGL_VARIANT_EXTEND_HEADS 2
#define MSGL_VARIANT_EXTEND_WIDE 3
#define MSGL_VARIANT_EXTEND_RING 4
#define MSGL_VARIANT_EXTEND_KEY_SPLIT 5
#define MSGL_VARIANT_DECODE_MFMA_DEPTH 6
#define MSGL_VARIANT_COUNT 7
int msgl_set_variant(int which, int value);
int msgl_get_variant(int which);

/* ---------------------------------------------------------------------------
 * KV-pool store.
 * Replaces MHAKVPool.set_kv_buffer (mini_sglang/mem_cache/token2kv.py:133-142):
 *   k_cache[loc[t]] = k[t]; v_cache[loc[t]] = v[t]   for t in [0, T).
 * k, v: [T, Hkv, D] bf16 with row strides k_stride / v_stride (v is a strided
 * view of the qkv GEMM output in the reference, qwen3.py:87, attention.py:32);
 * k_cache, v_cache: [slots, Hkv, D] bf16 with slot stride cache_stride.
 * loc: int32 [T] (BatchInfo.out_cache_loc).  Duplicate slots (graph padding rows
 * all target dummy slot 0, cuda_graph_runner.py:142-151) race benignly.
 * ------------------------------------------------------------------------- */
int msgl_store_kv(const void* k, int64_t k_stride, const void* v, int64_t v_stride,
                  const int32_t* loc, void* k_cache, void* v_cache, int64_t cache_stride,
                  int T, int Hkv, int D, void* stream);

/* ---------------------------------------------------------------------------
 * Paged decode attention (one query token per request).
 * Replaces TorchNativeAttnBackend._run_sdpa_forward_decode
 * (mini_sglang/layers/attn/torch_attn_backend.py:114-182) and the FA3 call at
 * fa3_attn_backend.py:136-150:
 *   for each request b: tok = req_to_token[req_pool_indices[b], :seq_lens[b]]
 *   out[b, h] = softmax(scale * q[b, h] . K[tok, h / (Hq/Hkv)]^T) . V[tok, h / (Hq/Hkv)]
 * non-causal over all seq_lens[b] slots (the slot written by this step included).
 *
 * q:   [bs, Hq, D] bf16, request stride q_stride;  out: same shape, stride out_stride.
 * k_cache / v_cache: [slots, Hkv, D] bf16, slot stride cache_stride.
 * req_to_token: int32 [max_reqs, max_ctx], row stride r2t_stride
 *   (ReqToTokenPool.req_to_token, mem_cache/req2token.py:43-45).  The kernel reads
 *   this table directly — no per-step page-table copy.  page_size is a layout
 *   promise: when page_size % 16 == 0, every aligned run of 16 token positions of a
 *   request occupies 16 consecutive slots (batch_info.py:192-234, 277-291), so the
 *   kernel reads one index per 16 tokens; any other page_size reads one per token.
 * seq_lens, req_pool_indices: int32 [bs] (device).
 *
 * Split-K: the grid is fixed at bs * max_splits work groups; a request uses
 * ceil(seq_len / chunk) of them (chunk = *chunk_dev if chunk_dev != NULL else
 * chunk_tokens; must be a positive multiple of 16 with
 * ceil(max seq_len / chunk) <= max_splits) and the rest exit at once, so the launch
 * shape never depends on the sequence lengths (hipGraph replay).
 * Cache policy: kv_stream = 0 reads every K/V row with ordinary (cached) loads.  kv_stream = 1
 * reads with the non-temporal hint, except positions below reuse_lens[b] (int32 [bs], may be
 * NULL = no such positions): those are the prefix slots that the radix cache shares between
 * requests (batch_info.py:184-186, 215-219) and are kept cacheable so that the other requests'
 * re-reads hit L2 / the Infinity Cache.  The policy never changes results.
 * plan: NULL (work groups in split-major order), or the bs * max_splits int32 entries
 *   msgl_decode_plan wrote for these very seq_lens / max_splits / chunk: the launch order
 *   for ragged batches.  Scheduling only, never changes results.
 * start_lens / extra_splits (both NULL, or int32 [bs]): request b's positions below start_lens[b]
 *   (a multiple of 16 when page_size % 16 == 0) are not read by this kernel; another kernel
 *   (msgl_shared_prefix_decode) has put extra_splits[b] partial results for them into the
 *   workspace slots behind this kernel's ceil((seq_len - start) / chunk) own ones, and the merge
 *   covers both.  Requires ceil((seq_len - start) / chunk) + extra_splits[b] <= max_splits.
 * workspace: >= msgl_paged_decode_workspace_bytes(...) bytes of device memory,
 *   16-byte aligned; contents are scratch.
 * Supported: D == 128; Hq/Hkv in {1, 2, 4, 8}; Hkv in {1, 2, 4, 8}.
 * ------------------------------------------------------------------------- */
size_t msgl_paged_decode_workspace_bytes(int bs, int Hq, int D, int max_splits);

int msgl_paged_decode(const void* q, int64_t q_stride,
                      const void* k_cache, const void* v_cache, int64_t cache_stride,
                      const int32_t* req_to_token, int64_t r2t_stride,
                      const int32_t* req_pool_indices, const int32_t* seq_lens,
                      void* out, int64_t out_stride,
                      void* workspace, size_t workspace_bytes,
                      int bs, int Hq, int Hkv, int D, int page_size, float scale,
                      int max_splits, int chunk_tokens, const int32_t* chunk_dev,
                      int kv_stream, const int32_t* reuse_lens, const int32_t* plan,
                      const int32_t* start_lens, const int32_t* extra_splits,
                      void* stream);

/* msgl_paged_decode for callers that cannot allocate and have no workspace to hand over
 * (code under hipGraph capture, tight serving loops, harnesses that forbid hipMalloc).  It has
 * no counterpart in the reference.  Same arguments, kernels and results as msgl_paged_decode
 * without start_lens / extra_splits; the split workspace is a static device array inside the
 * library, MSGL_PAGED_DECODE_STATIC_WORKSPACE_BYTES per GPU the library runs on, held until the
 * process ends.  Calls whose bs x max_splits x Hq needs more
 * (msgl_paged_decode_workspace_bytes) are refused with MSGL_E_WORKSPACE.  The call allocates
 * nothing and frees nothing.  The array's device address is looked up on the first call on a
 * device, so make that first call (a warm-up) outside graph capture.  All calls on one GPU share
 * the array: calls that may overlap there (different streams, no ordering between them) race on
 * it; calls on one stream, or ordered by events, do not.  Its contents need no clearing between
 * calls. */
#define MSGL_PAGED_DECODE_STATIC_WORKSPACE_BYTES (32u << 20)
int msgl_paged_decode_static(const void* q, int64_t q_stride,
                             const void* k_cache, const void* v_cache, int64_t cache_stride,
                             const int32_t* req_to_token, int64_t r2t_stride,
                             const int32_t* req_pool_indices, const int32_t* seq_lens,
                             void* out, int64_t out_stride,
                             int bs, int Hq, int Hkv, int D, int page_size, float scale,
                             int max_splits, int chunk_tokens, const int32_t* chunk_dev,
                             int kv_stream, const int32_t* reuse_lens, const int32_t* plan,
                             void* stream);

/* Launch order of msgl_paged_decode's (split, request) work items for a batch of unequal
 * lengths, built on the device (the lengths are never read back): every full split first, then
 * each request's partial last split by falling length; item = split * bs + request, unused
 * entries -1.  One call per decode step serves all layers: it belongs where the FA3 backend
 * refreshes its per-step metadata (fa3_attn_backend.py:43-82, 189-233) and can be captured in
 * the step's graph.  plan: int32 [bs * max_splits].  sorted_dev: NULL, or a device word read at
 * run time: 0 writes the identity order (= the launch without a plan) instead, so that one
 * captured graph serves uniform and ragged batches. */
int msgl_decode_plan(const int32_t* seq_lens, int bs, int max_splits,
                     int chunk_tokens, const int32_t* chunk_dev,
                     const int32_t* sorted_dev, const int32_t* start_lens,
                     int32_t* plan, void* stream);

/* ---------------------------------------------------------------------------
 * Decode attention over a radix prefix that several requests of the batch share
 * (rows of req_to_token that start with the same slots, batch_info.py:184-186,
 * 215-219), computed ONCE per group instead of once per request: the decode
 * queries of the group's members, all heads of a kv head, form the rows of a
 * matrix product against the shared keys (the extend kernel's tile loop), and the
 * result goes, as split partials, into the decode workspace.  Call it before
 * msgl_paged_decode on the same stream, workspace, seq_lens, max_splits and chunk;
 * that call then gets start_lens[b] = the group's prefix length (0 for requests in
 * no group) and extra_splits[b] = ceil(prefix / key_chunk), skips those positions
 * and merges these partials with its own.  The reference has no counterpart: its
 * backends read a shared prefix once per request.
 * Group g has group_count[g] members sharing positions [0, group_prefix[g]) of their
 * req_to_token rows (int32 device arrays of num_groups entries).  Its members are the
 * batch entries group_members[group_first[g] + i], i < group_count[g], or, with
 * group_members == NULL, the consecutive entries group_first[g] + i.  max_members / max_prefix:
 * host-side upper bounds (grid size).  key_chunk: keys per work group, a multiple
 * of 64.
 * Captured launches (hipGraph) cannot size the grid by the batch: pass items =
 * num_items int32 work items in device memory (group << 20 | 128-row tile of the
 * group's members x heads << 14 | key slice; -1 = unused entry; max_members /
 * max_prefix are then ignored) and, if the slice length changes between replays,
 * key_chunk_dev (device int32, overrides key_chunk).  The group arrays, start_lens
 * and the caller's extra_splits are then static buffers refreshed before a replay.
 * ------------------------------------------------------------------------- */
int msgl_shared_prefix_decode(const void* q, int64_t q_stride,
                              const void* k_cache, const void* v_cache, int64_t cache_stride,
                              const int32_t* req_to_token, int64_t r2t_stride,
                              const int32_t* req_pool_indices,
                              const int32_t* seq_lens, const int32_t* start_lens,
                              const int32_t* group_first, const int32_t* group_count,
                              const int32_t* group_prefix, const int32_t* group_members,
                              int num_groups,
                              int max_members, int max_prefix, int key_chunk,
                              void* workspace, size_t workspace_bytes,
                              int bs, int Hq, int Hkv, int D, float scale,
                              int max_splits, int chunk_tokens, const int32_t* chunk_dev,
                              const int32_t* items, int num_items, const int32_t* key_chunk_dev,
                              void* stream);

/* ---------------------------------------------------------------------------
 * Paged extend (prefill) attention, variable-length queries, bottom-right causal.
 * Replaces TorchNativeAttnBackend._run_sdpa_forward_extend
 * (torch_attn_backend.py:28-111) and the FA3 call at fa3_attn_backend.py:101-115:
 *   request b owns query rows [cu_q[b], cu_q[b+1]) (Sq_b = input_seq_lens[b]);
 *   query row i of b sits at position prefix_lens[b] + i and attends to token
 *   positions [0, prefix_lens[b] + i] of req_to_token[req_pool_indices[b]].
 * q / out: [sum Sq, Hq, D] bf16 (token stride q_stride / out_stride).
 * cu_seqlens_q: int32 [bs+1] exclusive prefix sum of the per-request query counts.
 * prefix_lens: int32 [bs].  max_seqlen_q: upper bound on Sq_b (sizes the grid).
 * max_seqlen_k: upper bound on prefix_lens[b] + Sq_b, the max_seq_len_k of the
 * FA3 metadata (fa3_attn_backend.py:43-82), or 0 when the caller does not know
 * it.  plan: NULL, or the launch order msgl_extend_plan wrote for these very
 * cu_seqlens_q / prefix_lens / max_seqlen_q.  Both only steer scheduling (kernel
 * variant, block order): the plan never changes a bit of the output, the kernel
 * variants agree to rounding (they order their sums and rescalings differently).
 * The new tokens' K/V must already be in the pool (msgl_store_kv first), as in
 * torch_attn_backend.py:195-196.
 * ------------------------------------------------------------------------- */
/* workspace (optional, NULL / 0 = none): msgl_paged_extend_workspace_bytes(total_tokens, Hq, D) bytes
 * of 16-byte aligned device memory, total_tokens = cu_seqlens_q[bs] (the host knows it: it
 * sized q).  With it, a launch of few work items over long key ranges (a short prompt on a
 * long cached prefix, few heads per GPU) cuts every item's keys into up to 8 runs, one work
 * group each, and a second small kernel joins the partial results; without it such a launch
 * leaves compute units idle or half occupied.  Results agree to fp32 rounding of the sums. */
size_t msgl_paged_extend_workspace_bytes(int64_t total_tokens, int Hq, int D);
/* The same, sized for the launch at hand instead of the worst case: msgl_paged_extend_key_splits returns the number of
 * key runs (1..8, 1 = no split) the library takes for a batch of bs requests with these bounds when the workspace
 * allows, and msgl_paged_extend_workspace_bytes_for the bytes that many runs need (0 for key_splits <= 1).  A smaller
 * workspace than that is legal too: the library then takes as many runs as fit. */
int msgl_paged_extend_key_splits(int bs, int max_seqlen_q, int max_seqlen_k, int Hq, int Hkv);
size_t msgl_paged_extend_workspace_bytes_for(int64_t total_tokens, int Hq, int D, int key_splits);

int msgl_paged_extend(const void* q, int64_t q_stride,
                      const void* k_cache, const void* v_cache, int64_t cache_stride,
                      const int32_t* req_to_token, int64_t r2t_stride,
                      const int32_t* req_pool_indices, const int32_t* cu_seqlens_q,
                      const int32_t* prefix_lens,
                      void* out, int64_t out_stride,
                      int bs, int max_seqlen_q, int max_seqlen_k, const int32_t* plan,
                      int Hq, int Hkv, int D, int page_size,
                      float scale, void* workspace, size_t workspace_bytes, int64_t total_tokens,
                      void* stream);

/* ---------------------------------------------------------------------------
 * Launch order for msgl_paged_extend on a ragged batch, computed on the device
 * once per forward pass and reused by every layer: the place the FA3 backend
 * builds its per-batch metadata (fa3_attn_backend.py:43-82, init_forward_metadata).
 * Work items (request b, 128-row query tile qt), encoded b * tiles + qt with
 * tiles = ceil(max_seqlen_q / 128), are sorted by their number of 64-key tiles
 * (prefix included), longest first; plan has msgl_extend_plan_items(bs,
 * max_seqlen_q) int32 entries, the unused tail is -1.  The order among items of
 * equal length is unspecified.
 * ------------------------------------------------------------------------- */
int64_t msgl_extend_plan_items(int bs, int max_seqlen_q);
int msgl_extend_plan(const int32_t* cu_seqlens_q, const int32_t* prefix_lens,
                     int bs, int max_seqlen_q, int32_t* plan, void* stream);

/* ---------------------------------------------------------------------------
 * RMSNorm.  Replaces RMSNorm.forward(x) (mini_sglang/layers/layernorm.py:39-50,
 * residual is None):  out = bf16( x_f32 * rsqrt(mean(x_f32^2) + eps) * w_f32 ).
 * x: [rows, cols] bf16 with row stride x_stride (q/k-norm rows are 128-wide slices of
 * the qkv output, qwen3.py:88-93); out row stride out_stride; w: [cols] bf16.
 * cols % 8 == 0, cols <= 16384.  out may alias x.
 * ------------------------------------------------------------------------- */
int msgl_rmsnorm(const void* x, int64_t x_stride, const void* w, void* out, int64_t out_stride,
                 int rows, int cols, float eps, void* stream);

/* Fused residual-add RMSNorm, in place on both tensors.  Replaces
 * RMSNorm.forward(x, residual) (layernorm.py:41-49; in-place form 33-35):
 *   s = x_f32 + residual_f32; residual = bf16(s);
 *   x = bf16( s * rsqrt(mean(s^2) + eps) * w_f32 )   (norm uses the UNROUNDED s).
 * x, residual: [rows, cols] bf16 contiguous rows (stride = cols). */
int msgl_fused_add_rmsnorm(void* x, void* residual, const void* w,
                           int rows, int cols, float eps, void* stream);

/* SiLU-gate.  Replaces SiluAndMul.forward (mini_sglang/layers/activation.py:12-14):
 *   out[r, c] = bf16( bf16(silu_f32(x[r, c])) * x[r, n + c] ),  c in [0, n).
 * x: [rows, 2n] bf16 with row stride x_stride; out: [rows, n] contiguous.  n % 8 == 0. */
int msgl_silu_and_mul(const void* x, int64_t x_stride, void* out, int rows, int n, void* stream);

/* Neox rotary embedding, in place on q and k.  Replaces RotaryEmbedding.forward
 * (mini_sglang/layers/rotary_embedding.py:70-95; table layout 59-67):
 *   cos = cache[pos, :D/2], sin = cache[pos, D/2:]  (fp32 table [max_pos, D])
 *   y1 = x1*cos - x2*sin ; y2 = x1*sin + x2*cos,  x1 = x[..., :D/2], x2 = x[..., D/2:].
 * rounding: 0 = reproduce the reference's torch fallback bit for bit (cos/sin rounded
 *   to bf16 and every multiply / add rounded to bf16, rotary_embedding.py:28-33);
 *   1 = fp32 arithmetic with one final rounding (what sgl_kernel's kernel does,
 *   rotary_embedding.py:77-84).
 * positions: int32 [T]; q: [T, Hq, D] token stride q_stride; k: [T, Hkv, D] stride k_stride. */
int msgl_rope_neox(const int32_t* positions, void* q, int64_t q_stride, void* k, int64_t k_stride,
                   const float* cos_sin_cache, int T, int Hq, int Hkv, int D, int rounding,
                   void* stream);

/* Fused per-head q/k RMSNorm -> neox RoPE -> KV-pool store for one layer's qkv GEMM
 * output.  Replaces the op sequence at mini_sglang/models/qwen3.py:88-94 plus
 * set_kv_buffer (token2kv.py:141-142), with each stage rounded to bf16 exactly where the
 * unfused reference ops round.
 *   qkv: [T, (Hq + 2*Hkv) * D] bf16, token stride qkv_stride; q and k parts are
 *   normalised and rotated IN PLACE; the rotated k and the untouched v are copied to
 *   k_cache / v_cache at slot loc[t] (skipped when loc == NULL). */
int msgl_qknorm_rope_store(void* qkv, int64_t qkv_stride, const void* q_norm_w, const void* k_norm_w,
                           float eps, const int32_t* positions, const float* cos_sin_cache,
                           const int32_t* loc, void* k_cache, void* v_cache, int64_t cache_stride,
                           int T, int Hq, int Hkv, int D, int rounding, void* stream);

/* Greedy sampling helper: out[b] = argmax_j logits[b, j] over fp32(logits), first index
 * on ties.  Replaces the temperature == 0 branch of Sampler.forward
 * (mini_sglang/layers/sampler.py:14-15, 23).  logits: [bs, vocab] bf16 or fp32
 * (elem_bytes 2 or 4), row stride in elements; out: int32 [bs]. */
int msgl_argmax(const void* logits, int64_t row_stride, int elem_bytes, int32_t* out,
                int bs, int vocab, void* stream);

/* The same, also returning the winning value: out_val[b] = fp32(logits[b, out[b]]) (NULL = not wanted).
 * With the vocabulary sharded over tensor-parallel ranks (lm_head rows, mini_sglang/layers/embed_head.py:56-80)
 * every rank reduces its own shard and the ranks exchange bs (value, index) pairs instead of the
 * [bs, vocab] logits that mini_sglang/layers/logits_processor.py:56-64 all-gathers. */
int msgl_argmax_val(const void* logits, int64_t row_stride, int elem_bytes, int32_t* out, float* out_val,
                    int bs, int vocab, void* stream);

/* Greedy token straight from the final hidden state, for decode-sized batches: out[b] = argmax_v bf16(x[b, :] . w[v, :]),
 * first index on ties, NaN largest; out_val[b] = that logit as fp32 (NULL = not wanted).  Replaces
 * LogitsProcessor._get_logits + the temperature == 0 branch of Sampler.forward
 * (mini_sglang/layers/logits_processor.py:56-59, mini_sglang/layers/sampler.py:14-15) on one rank's vocabulary shard
 * (ParallelLMHead rows, mini_sglang/layers/embed_head.py:56-80) without writing the [bs, vocab] logits: the product is
 * accumulated in fp32 on the matrix cores, rounded to bf16 as F.linear's output would be, and reduced in the same pass.
 * Under tensor parallelism the ranks then exchange bs (value, index) pairs, as with msgl_argmax_val.
 *   x: [bs, hidden] bf16, row stride x_stride elements; w: [vocab, hidden] bf16, row stride w_stride (both 16-byte
 *   aligned, strides multiples of 8); hidden a multiple of 256; out: int32 [bs] (row index inside w).
 *   workspace: msgl_lm_head_argmax_workspace_bytes(bs, vocab) bytes of device memory (partial winners per block of
 *   128 rows), contents irrelevant before and after.  Sums differ from another GEMM's in the last fp32 bits, so a logit
 *   can round to the neighbouring bf16 value: the token equals argmax(F.linear(x, w)) whenever the top two logits are
 *   more than one bf16 ulp apart. */
size_t msgl_lm_head_argmax_workspace_bytes(int bs, int vocab);
int msgl_lm_head_argmax(const void* x, int64_t x_stride, const void* w, int64_t w_stride, int32_t* out, float* out_val,
                        int bs, int vocab, int hidden, void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * Decode-step preparation on the device.  Replaces BatchInfo.prepare_for_decode
 * (mini_sglang/managers/batch_info.py:247-302) together with the allocator call inside it
 * (PageAllocator.alloc takes pages from the FRONT of the free list,
 * mini_sglang/mem_cache/token2kv.py:186-196), so that decode steps can follow each other
 * on the GPU without the host:
 *   old = seq_lens[i];  positions[i] = old;  seq_lens[i] = old + 1;
 *   slot = (page_size == 1 || old % page_size == 0)
 *            ? free_pages[free_range[0] + (number of requests j < i that also take a page)] * page_size
 *            : last_slots[i] + 1;
 *   out_cache_loc[i] = last_slots[i] = req_to_token[req_pool_indices[i], old] = slot;
 *   input_ids[i] = next_tokens[i]            (both NULL: the caller sets input_ids itself)
 *   free_range[0] += number of pages taken.
 * free_pages: int32 [free_capacity], the allocator's free list in its order; free_range: device
 * int32 [2] = {index of the first page not yet handed out, end of the valid entries}: both live on
 * the device so that a captured launch keeps working when the host rewrites the list between
 * steps.  last_slots: int32 [bs], slot of each request's last stored token.  All-or-nothing: if the free list cannot serve the step
 * (*status = 1) or a request already fills its table row of max_ctx entries (*status = 2)
 * nothing is written; *status is left untouched on success (the caller zeroes it once).
 * ------------------------------------------------------------------------- */
int msgl_prepare_decode(const int32_t* req_pool_indices, int32_t* seq_lens, int32_t* last_slots,
                        const int32_t* free_pages, int32_t free_capacity, int32_t* free_range,
                        int32_t* req_to_token, int64_t r2t_stride, int32_t max_ctx,
                        int32_t* positions, int32_t* out_cache_loc,
                        const int32_t* next_tokens, int32_t* input_ids,
                        int32_t* status, int bs, int page_size, void* stream);
/* The same, also keeping a page table: req_to_page[req_pool_indices[i], old / page_size] = page id whenever request i
 * takes a page (page_size > 1; NULL = no page table).  Replaces the page half of ReqToTokenPool's bookkeeping
 * (mini_sglang/mem_cache/req2token.py:43-51, 77-91) for steps prepared on the device; r2p_stride in elements. */
int msgl_prepare_decode_paged(const int32_t* req_pool_indices, int32_t* seq_lens, int32_t* last_slots,
                              const int32_t* free_pages, int32_t free_capacity, int32_t* free_range,
                              int32_t* req_to_token, int64_t r2t_stride, int32_t* req_to_page, int64_t r2p_stride,
                              int32_t max_ctx, int32_t* positions, int32_t* out_cache_loc,
                              const int32_t* next_tokens, int32_t* input_ids,
                              int32_t* status, int bs, int page_size, void* stream);

/* ---------------------------------------------------------------------------
 * Measurement utility (no counterpart in the reference): reads `bytes` bytes at src
 * once, 16 bytes per lane with the non-temporal hint, and stores nothing (sink: 256
 * bytes of device memory that are practically never written).  bench.py times it as
 * the on-box read-only streaming rate beside the decode kernel's.
 * ------------------------------------------------------------------------- */
int msgl_stream_read(const void* src, size_t bytes, void* sink, void* stream);
/* The same with the kernel's two tuning knobs exposed: unroll = 16-byte loads in flight per lane (1, 2, 4 or 8),
 * groups_per_cu = resident 256-thread work groups per compute unit the grid is capped at (1..64).
 * msgl_stream_read is (8, 16).  bench.py sweeps a few pairs and quotes the best as its yardstick. */
int msgl_stream_read_cfg(const void* src, size_t bytes, void* sink, int unroll, int groups_per_cu, void* stream);
/* The same bytes read by contiguous spans of span_kib KiB (a multiple of 4), one work group per span front to back, spans dealt
 * round robin over at most groups_per_cu resident work groups per compute unit: the order in which a gather of whole KV pages
 * walks memory.  Part of the yardstick sweep. */
int msgl_stream_read_span(const void* src, size_t bytes, void* sink, int unroll, int groups_per_cu, int span_kib, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MSGL_HIP_H */
