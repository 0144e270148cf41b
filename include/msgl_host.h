/*
 * msgl_host.h — C ABI of libmsgl_host.so: host-side (CPU) data structures of the KV-cache path.
 * Plain C types only; no device memory is touched.  The library is single-threaded per handle, like the
 * scheduler that owns the reference's cache (mini_sglang/managers/scheduler.py:167-180).
 *
 * Radix prefix cache.  Replaces the tree inside RadixCache (mini_sglang/mem_cache/radix_cache.py:22-214,
 * 216-238, 367-418): token-id keys compared page by page, int32 KV slot values, lock counts from a node to
 * the root, least-recently-used eviction of unlocked leaves.  The request-level methods of the reference class
 * (cache_unfinished_req / cache_finished_req, radix_cache.py:270-365) stay in the Python mirror
 * (mini_sglang_amd/mem_cache/radix_cache.py) because they edit torch tensors.
 *
 * Page allocator: see the second half of this header.
 *
 * Node handles are int64 ids; 0 is the root.  A handle stays valid while the node exists: splitting a node keeps
 * the id on the LOWER part (as the reference keeps the object), eviction deletes ids of unlocked leaves only.
 */
#ifndef MSGL_HOST_H
#define MSGL_HOST_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msgl_radix msgl_radix;

/* RadixCache.__init__ / reset (radix_cache.py:54-83).  page_size >= 1.  NULL on failure. */
msgl_radix* msgl_radix_create(int page_size);
void msgl_radix_destroy(msgl_radix* t);
void msgl_radix_reset(msgl_radix* t);

/* match_prefix (radix_cache.py:117-168): the key is cut to a multiple of page_size, the tree is walked, a node
 * is split when the match ends inside it.  Writes the matched slots to out[0..cap) and the last node's handle
 * to *node; returns the matched length, or -(length) if cap is too small (nothing written, the tree is still
 * updated exactly as the reference updates it on a match). */
int64_t msgl_radix_match_prefix(msgl_radix* t, const int32_t* key, int64_t n, int32_t* out, int64_t cap, int64_t* node);

/* insert (radix_cache.py:170-214): returns the length of the prefix that was already present. */
int64_t msgl_radix_insert(msgl_radix* t, const int32_t* key, const int32_t* value, int64_t n);

/* inc_lock_ref / dec_lock_ref (radix_cache.py:216-238).  0 on success, -1 for an unknown handle. */
int msgl_radix_inc_lock_ref(msgl_radix* t, int64_t node);
int msgl_radix_dec_lock_ref(msgl_radix* t, int64_t node);

/* evict (radix_cache.py:367-390): pops unlocked leaves in order of last access until at least num_tokens slots
 * are evicted.  The evicted nodes' slot runs are appended to slots[0..slot_cap) and their lengths to
 * lens[0..len_cap) in eviction order; *n_nodes = nodes evicted.  Returns the number of slots evicted, or -1 if
 * slot_cap < msgl_radix_evictable_size() or len_cap < msgl_radix_node_count() (then NOTHING is evicted). */
int64_t msgl_radix_evict(msgl_radix* t, int64_t num_tokens, int32_t* slots, int64_t slot_cap, int64_t* lens, int64_t len_cap,
                         int64_t* n_nodes);

/* evictable_size / protected_size (radix_cache.py:392-396) and the number of nodes below the root. */
int64_t msgl_radix_evictable_size(const msgl_radix* t);
int64_t msgl_radix_protected_size(const msgl_radix* t);
int64_t msgl_radix_node_count(const msgl_radix* t);

/* ---------------------------------------------------------------------------
 * Page allocator of the KV pool.  Replaces PageAllocator (mini_sglang/mem_cache/token2kv.py:154-212): page p owns
 * slots [p * page_size, (p + 1) * page_size); pages 1..page_num start free in ascending order (page 0 is the dummy
 * page padded rows write to); alloc takes from the FRONT of the free list, free appends at the BACK in the order
 * given.  The order is part of the contract: it decides every slot a request gets.
 * ------------------------------------------------------------------------- */
typedef struct msgl_pages msgl_pages;

/* PageAllocator.__init__ / clear (token2kv.py:166-183, 209-212).  NULL on failure. */
msgl_pages* msgl_pages_create(int64_t page_num, int page_size);
void msgl_pages_destroy(msgl_pages* a);
void msgl_pages_clear(msgl_pages* a);

/* len(free_page): available_size() = free pages * page_size (token2kv.py:185-186). */
int64_t msgl_pages_free_count(const msgl_pages* a);

/* alloc (token2kv.py:188-202): the first num free pages, in list order, into out[0..num).  Returns num, or -1 when
 * fewer are free (nothing is taken; the mirror raises the reference's RuntimeError), -2 for bad arguments. */
int64_t msgl_pages_alloc(msgl_pages* a, int64_t num, int32_t* out);

/* free (token2kv.py:204-207): appends pages[0..n) at the back.  Returns the new number of free pages, or -1 when
 * an id is outside 1..page_num (nothing is added), -2 for bad arguments. */
int64_t msgl_pages_free(msgl_pages* a, const int32_t* pages, int64_t n);

/* The free list in order, for callers that mirror it elsewhere (the device-side decode preparation keeps a copy in
 * HBM): copies min(free count, cap) ids to out and returns the free count. */
int64_t msgl_pages_peek(const msgl_pages* a, int32_t* out, int64_t cap);

/* Replaces the free list by pages[0..n) (the reference's callers assign `free_page` directly when they lay out a
 * pool by hand).  Returns n, or -1 for an id outside 1..page_num (list unchanged). */
int64_t msgl_pages_assign(msgl_pages* a, const int32_t* pages, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* MSGL_HOST_H */
