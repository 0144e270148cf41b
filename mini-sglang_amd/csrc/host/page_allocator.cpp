// Page allocator of the KV pool (host side): a FIFO free list of page ids behind the C ABI of include/msgl_host.h.
// Semantics of the reference's PageAllocator (mini_sglang/mem_cache/token2kv.py:154-212): pages 1..page_num start
// free in ascending order (page 0 is the dummy page), alloc takes from the FRONT, free appends at the BACK in the
// order given -- the order is part of the contract, because which page a request gets decides its slots and with
// them every page table the kernels read.  A ring buffer instead of the reference's tensor slicing / concatenation:
// alloc and free cost the pages they move, not the length of the list.
#include <stddef.h>
#include <stdint.h>

#include <new>
#include <vector>

#include "../../../include/msgl_host.h"

struct msgl_pages {
    int64_t page_num;
    int page_size;
    // ring of capacity page_num + 1 slots would do if every page were freed at most once; a caller that frees a page
    // twice (the reference does not guard against it either) must not corrupt memory, so the ring grows instead
    std::vector<int32_t> ring;
    int64_t head = 0;   // index of the first free page
    int64_t count = 0;  // free pages

    void reset() {
        ring.assign((size_t)page_num + 1, 0);
        for (int64_t i = 0; i < page_num; ++i) ring[(size_t)i] = (int32_t)(i + 1);
        head = 0;
        count = page_num;
    }
    void grow(int64_t need) {
        if (need <= (int64_t)ring.size()) return;
        std::vector<int32_t> bigger((size_t)(need * 2));
        for (int64_t i = 0; i < count; ++i) bigger[(size_t)i] = ring[(size_t)((head + i) % (int64_t)ring.size())];
        ring.swap(bigger);
        head = 0;
    }
};

extern "C" msgl_pages* msgl_pages_create(int64_t page_num, int page_size) {
    if (page_num < 0 || page_num > 0x7ffffffe || page_size < 1) return nullptr;
    msgl_pages* a = new (std::nothrow) msgl_pages();
    if (!a) return nullptr;
    a->page_num = page_num;
    a->page_size = page_size;
    a->reset();
    return a;
}
extern "C" void msgl_pages_destroy(msgl_pages* a) { delete a; }
extern "C" void msgl_pages_clear(msgl_pages* a) {
    if (a) a->reset();
}
extern "C" int64_t msgl_pages_free_count(const msgl_pages* a) { return a ? a->count : -1; }

extern "C" int64_t msgl_pages_alloc(msgl_pages* a, int64_t num, int32_t* out) {
    if (!a || num < 0 || (num > 0 && !out)) return -2;
    if (num > a->count) return -1;  // nothing is taken
    const int64_t cap = (int64_t)a->ring.size();
    for (int64_t i = 0; i < num; ++i) out[i] = a->ring[(size_t)((a->head + i) % cap)];
    a->head = (a->head + num) % cap;
    a->count -= num;
    return num;
}

extern "C" int64_t msgl_pages_free(msgl_pages* a, const int32_t* pages, int64_t n) {
    if (!a || n < 0 || (n > 0 && !pages)) return -2;
    for (int64_t i = 0; i < n; ++i)
        if (pages[i] < 1 || pages[i] > a->page_num) return -1;  // nothing is added
    a->grow(a->count + n);
    const int64_t cap = (int64_t)a->ring.size();
    for (int64_t i = 0; i < n; ++i) a->ring[(size_t)((a->head + a->count + i) % cap)] = pages[i];
    a->count += n;
    return a->count;
}

extern "C" int64_t msgl_pages_peek(const msgl_pages* a, int32_t* out, int64_t cap_out) {
    if (!a || cap_out < 0 || (cap_out > 0 && !out)) return -2;
    const int64_t n = a->count < cap_out ? a->count : cap_out;
    const int64_t cap = (int64_t)a->ring.size();
    for (int64_t i = 0; i < n; ++i) out[i] = a->ring[(size_t)((a->head + i) % cap)];
    return a->count;
}

extern "C" int64_t msgl_pages_assign(msgl_pages* a, const int32_t* pages, int64_t n) {
    if (!a || n < 0 || (n > 0 && !pages)) return -2;
    for (int64_t i = 0; i < n; ++i)
        if (pages[i] < 1 || pages[i] > a->page_num) return -1;
    a->head = 0;
    a->count = 0;
    a->grow(n + 1);
    for (int64_t i = 0; i < n; ++i) a->ring[(size_t)i] = pages[i];
    a->count = n;
    return n;
}
