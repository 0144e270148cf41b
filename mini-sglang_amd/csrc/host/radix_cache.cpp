// Radix prefix cache over token ids (host side).  Keys are compared a page at a time, values are KV slot
// indices; every node remembers when it was last touched, and eviction removes unlocked leaves oldest first.
// The access clock advances on the same events as the reference implementation's
// (mini_sglang/mem_cache/radix_cache.py:30-47: node creation and every update_access_time call in
// match_prefix / insert), so both pick the same victims in the same order.
#include <algorithm>
#include <cstdint>
#include <map>
#include <new>
#include <queue>
#include <unordered_map>
#include <vector>

#include "../../../include/msgl_host.h"

namespace {

struct Node {
    std::map<std::vector<int32_t>, Node*> children;  // first page of the child's key -> child
    Node* parent = nullptr;
    std::vector<int32_t> key, value;
    int64_t lock_ref = 0;
    int64_t time = 0;
    int64_t id = 0;
};

}  // namespace

struct msgl_radix {
    int ps = 1;
    int64_t clock = 0;
    int64_t next_id = 0;
    int64_t evictable = 0, protected_ = 0;
    Node* root = nullptr;
    std::unordered_map<int64_t, Node*> by_id;

    int64_t tick() { return ++clock; }

    Node* make_node() {
        Node* n = new Node();
        n->time = tick();
        n->id = next_id++;
        by_id[n->id] = n;
        return n;
    }

    void free_subtree(Node* n) {
        for (auto& kv : n->children) free_subtree(kv.second);
        by_id.erase(n->id);
        delete n;
    }

    void reset() {
        if (root) free_subtree(root);
        next_id = 0;
        root = make_node();  // id 0
        root->lock_ref = 1;
        evictable = protected_ = 0;
    }

    std::vector<int32_t> child_key(const int32_t* key, int64_t n) const {
        return std::vector<int32_t>(key, key + std::min<int64_t>(n, ps));
    }

    // number of leading tokens equal page by page (a shorter last page never equals a full one)
    int64_t match_len(const int32_t* a, int64_t na, const std::vector<int32_t>& b) const {
        const int64_t n = std::min<int64_t>(na, (int64_t)b.size());
        int64_t i = 0;
        while (i < n) {
            const int64_t la = std::min<int64_t>(ps, na - i), lb = std::min<int64_t>(ps, (int64_t)b.size() - i);
            if (la != lb || !std::equal(a + i, a + i + la, b.begin() + i)) break;
            i += ps;
        }
        return i;
    }

    // upper part becomes a new node, `child` keeps its identity as the lower part
    Node* split(Node* child, int64_t pos) {
        Node* up = make_node();
        up->parent = child->parent;
        up->key.assign(child->key.begin(), child->key.begin() + pos);
        up->value.assign(child->value.begin(), child->value.begin() + pos);
        up->lock_ref = child->lock_ref;
        child->key.erase(child->key.begin(), child->key.begin() + pos);
        child->value.erase(child->value.begin(), child->value.begin() + pos);
        child->parent = up;
        up->children[child_key(child->key.data(), (int64_t)child->key.size())] = child;
        up->parent->children[child_key(up->key.data(), (int64_t)up->key.size())] = up;
        return up;
    }
};

extern "C" {

msgl_radix* msgl_radix_create(int page_size) {
    if (page_size < 1) return nullptr;
    msgl_radix* t = new (std::nothrow) msgl_radix();
    if (!t) return nullptr;
    t->ps = page_size;
    t->reset();
    return t;
}

void msgl_radix_destroy(msgl_radix* t) {
    if (!t) return;
    if (t->root) t->free_subtree(t->root);
    delete t;
}

void msgl_radix_reset(msgl_radix* t) { t->reset(); }

int64_t msgl_radix_match_prefix(msgl_radix* t, const int32_t* key, int64_t n, int32_t* out, int64_t cap, int64_t* node_out) {
    if (node_out) *node_out = 0;
    if (n <= 0) return 0;
    if (t->ps != 1) n = n / t->ps * t->ps;
    Node* node = t->root;
    node->time = t->tick();
    std::vector<const std::vector<int32_t>*> parts;
    int64_t total = 0;
    while (n > 0) {
        auto it = node->children.find(t->child_key(key, n));
        if (it == node->children.end()) break;
        Node* child = it->second;
        child->time = t->tick();
        const int64_t m = t->match_len(key, n, child->key);
        if (m < (int64_t)child->key.size()) {
            node = t->split(child, m);
            parts.push_back(&node->value);
            total += (int64_t)node->value.size();
            break;
        }
        parts.push_back(&child->value);
        total += (int64_t)child->value.size();
        node = child;
        key += m;
        n -= m;
    }
    if (node_out) *node_out = node->id;
    if (total > cap) return -total;
    for (const auto* p : parts) out = std::copy(p->begin(), p->end(), out);
    return total;
}

int64_t msgl_radix_insert(msgl_radix* t, const int32_t* key, const int32_t* value, int64_t n) {
    if (n <= 0) return 0;
    Node* node = t->root;
    node->time = t->tick();
    int64_t total = 0;
    while (n > 0) {
        auto it = node->children.find(t->child_key(key, n));
        if (it == node->children.end()) break;
        node = it->second;
        node->time = t->tick();
        const int64_t m = t->match_len(key, n, node->key);
        total += m;
        key += m;
        value += m;
        n -= m;
        if (m < (int64_t)node->key.size()) node = t->split(node, m);
    }
    if (n > 0) {
        Node* leaf = t->make_node();
        leaf->parent = node;
        leaf->key.assign(key, key + n);
        leaf->value.assign(value, value + n);
        node->children[t->child_key(key, n)] = leaf;
        t->evictable += n;
    }
    return total;
}

int msgl_radix_inc_lock_ref(msgl_radix* t, int64_t id) {
    auto it = t->by_id.find(id);
    if (it == t->by_id.end()) return -1;
    for (Node* n = it->second; n != t->root; n = n->parent) {
        if (n->lock_ref == 0) {
            t->evictable -= (int64_t)n->value.size();
            t->protected_ += (int64_t)n->value.size();
        }
        ++n->lock_ref;
    }
    return 0;
}

int msgl_radix_dec_lock_ref(msgl_radix* t, int64_t id) {
    auto it = t->by_id.find(id);
    if (it == t->by_id.end()) return -1;
    for (Node* n = it->second; n != t->root; n = n->parent) {
        --n->lock_ref;
        if (n->lock_ref == 0) {
            t->evictable += (int64_t)n->value.size();
            t->protected_ -= (int64_t)n->value.size();
        }
    }
    return 0;
}

int64_t msgl_radix_evict(msgl_radix* t, int64_t num_tokens, int32_t* slots, int64_t slot_cap, int64_t* lens, int64_t len_cap,
                         int64_t* n_nodes) {
    if (n_nodes) *n_nodes = 0;
    // buffers must hold the worst case (everything evictable goes), so the call never stops half way
    if (slot_cap < t->evictable || len_cap < (int64_t)t->by_id.size() - 1) return -1;
    struct Older {
        bool operator()(const Node* a, const Node* b) const { return a->time > b->time; }
    };
    std::priority_queue<Node*, std::vector<Node*>, Older> leaves;
    std::vector<Node*> stack{t->root};
    while (!stack.empty()) {
        Node* cur = stack.back();
        stack.pop_back();
        if (cur->children.empty()) leaves.push(cur);
        else for (auto& kv : cur->children) stack.push_back(kv.second);
    }
    int64_t done = 0, nodes = 0;
    while (done < num_tokens && !leaves.empty()) {
        Node* x = leaves.top();
        leaves.pop();
        if (x == t->root) break;
        if (x->lock_ref > 0) continue;
        const int64_t len = (int64_t)x->value.size();
        std::copy(x->value.begin(), x->value.end(), slots + done);
        lens[nodes++] = len;
        done += len;
        Node* parent = x->parent;
        for (auto it = parent->children.begin(); it != parent->children.end(); ++it)
            if (it->second == x) {
                parent->children.erase(it);
                break;
            }
        t->evictable -= (int64_t)x->key.size();
        t->by_id.erase(x->id);
        delete x;
        if (parent->children.empty()) leaves.push(parent);
    }
    if (n_nodes) *n_nodes = nodes;
    return done;
}

int64_t msgl_radix_evictable_size(const msgl_radix* t) { return t->evictable; }
int64_t msgl_radix_protected_size(const msgl_radix* t) { return t->protected_; }
int64_t msgl_radix_node_count(const msgl_radix* t) { return (int64_t)t->by_id.size() - 1; }

}  // extern "C"
