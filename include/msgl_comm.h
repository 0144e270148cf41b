/*
 * msgl_comm.h — C ABI of the tensor-parallel sum all-reduce in libmsgl_hip.so:
 * one process per GPU, every rank reads its peers' buffers directly over xGMI
 * (IPC-mapped device memory), no ring.  gfx950 only.
 *
 * It stands where the reference calls torch.distributed.all_reduce on the
 * partial outputs of its row-parallel linears and of the vocab-parallel
 * embedding (mini_sglang/layers/linear.py:244-248, embed_head.py:45-53):
 * [tokens, hidden] bf16, 2 x layers + 1 times per forward, latency-bound at
 * decode batch sizes.
 *
 * Conventions as in msgl_hip.h: int status (0 = ok), msgl_last_error() for the
 * message, streams as void* (hipStream_t).  A communicator belongs to the
 * device that was current when it was created and is used by one host thread.
 */
#ifndef MSGL_COMM_H_
#define MSGL_COMM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSGL_AR_MAX_WORLD 8
#define MSGL_AR_EXPORT_BYTES 128 /* two IPC memory handles: staging, flags */

typedef struct msgl_ar msgl_ar_t;

/* Allocates this rank's staging area (4 x max_bytes of device memory) and its
 * flag words (uncached device memory, zeroed).  world in [2, 8]; max_bytes =
 * largest message, a multiple of 16 x world.  The only call here that
 * allocates. */
int msgl_ar_create(int rank, int world, size_t max_bytes, msgl_ar_t** out);

/* IPC handles of this rank's staging area and flags, MSGL_AR_EXPORT_BYTES bytes.
 * The caller exchanges them among the ranks by any means (the reference's
 * process group: one all_gather of 128 bytes at start-up). */
int msgl_ar_export(msgl_ar_t* c, void* handles_out);

/* all_handles: world x MSGL_AR_EXPORT_BYTES bytes in rank order (this rank's own
 * entry is ignored).  Maps every peer's staging area and flags.  Every rank must
 * have returned from msgl_ar_create before any rank calls msgl_ar_allreduce —
 * the handle exchange in between guarantees it. */
int msgl_ar_open(msgl_ar_t* c, const void* all_handles);

/* out[i] = bf16( sum over ranks r = 0 .. world-1, in that order, of in_r[i] as
 * f32 ): one rounding, the same bits on every rank (torch.distributed's ring
 * rounds after every hop, so its result may differ from this one by bf16
 * ulps).  in / out: device bf16, 16-byte aligned, numel a multiple of 8,
 * numel * 2 <= max_bytes; out may be in.  Every rank calls with the same numel,
 * in the same order.  Enqueues one kernel on `stream`; nothing is allocated and
 * no host state changes, so the call can be captured in a hipGraph and
 * replayed.  Messages up to one_shot_bytes (0 = default) take the one-step
 * form (each rank sums all peers' inputs), longer ones the two-step form
 * (reduce a 1/world slice, then gather the slices). */
int msgl_ar_allreduce_bf16(msgl_ar_t* c, const void* in, void* out, int64_t numel,
                           size_t one_shot_bytes, void* stream);

/* The same call issued step by step, for set-ups in which the ranks' kernels
 * cannot be relied upon to run at the same time (several ranks sharing ONE
 * device in a rehearsal: a kernel that spins for its peer may keep the peer's
 * kernel from starting).  msgl_ar_num_steps gives the number of steps of the
 * form the call would take (2 or 3; 0 for numel <= 0).  Every rank issues step
 * s = 0, 1, .. with the same arguments, and between two steps every rank waits
 * until step s has COMPLETED on all ranks (stream synchronisation + a host
 * barrier); the device-side waits then find their flags raised.  Same device
 * code, flags, call numbering and result as the one-launch call.  The reference
 * has no counterpart (its all-reduce is NCCL's). */
int msgl_ar_num_steps(msgl_ar_t* c, int64_t numel, size_t one_shot_bytes);
int msgl_ar_allreduce_bf16_step(msgl_ar_t* c, const void* in, void* out, int64_t numel,
                                size_t one_shot_bytes, int step, void* stream);

/* How long a kernel waits for a peer before it gives up (default 1000 ms;
 * 1 .. 60000).  Applies to calls issued afterwards. */
int msgl_ar_set_timeout_ms(msgl_ar_t* c, int milliseconds);

/* Synchronises the device and returns 0, or MSGL_E_LAUNCH when a kernel gave up
 * waiting for a peer (a rank died, the ranks disagree on the call sequence, or
 * the peer's kernel never ran beside this one); the waiting is bounded so that
 * no kernel spins forever, and a work group that gave up signals nothing
 * further, so the peers' waits of that call run out as well. */
int msgl_ar_check(msgl_ar_t* c);

/* Unmaps the peers and frees this rank's memory.  The caller makes sure (with a
 * host-side barrier) that no rank is still inside an all-reduce. */
int msgl_ar_destroy(msgl_ar_t* c);

#ifdef __cplusplus
}
#endif
#endif /* MSGL_COMM_H_ */
