// rc_tree.cpp -- librubiktree.so: host-side trees of the lockstep search (include/rubiktree.h).
// Plain C++17 + OpenMP, no GPU code: g++ -O2 -fopenmp -ffp-contract=off -fPIC -shared.
// Restates mcts.py:52-81 (traverse), :83-113 (node insertion), :115-130 (back-propagation), :132-154 (PUCT) with the
// reference's arithmetic types and CPython's random.randint, so a lockstep search reproduces stand-alone runs.
#include "../../include/rubiktree.h"

#include <omp.h>

#include <cmath>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

constexpr int kMaxA = 12, kMaxSlots = 20;

// CPython's Mersenne Twister (Modules/_randommodule.c): state = 624 words + index
struct Mt {
    uint32_t mt[624];
    uint32_t idx;
    uint32_t next() {
        if (idx >= 624) {
            for (int k = 0; k < 624; ++k) {
                const uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        return y;
    }
    // random.randint(0, n - 1) = randrange(n) = _randbelow_with_getrandbits(n): k = n.bit_length() bits, rejection
    int below(int n) {
        int k = 0;
        while ((n >> k) != 0) ++k;
        uint32_t r = next() >> (32 - k);
        while ((int)r >= n) r = next() >> (32 - k);
        return (int)r;
    }
};

struct Node {
    float policy[kMaxA];
    double value[kMaxA];     // value_min (a Python float) until a float32 leaf value replaces it
    double vloss[kMaxA];
    int32_t visits[kMaxA];
    int32_t child[kMaxA];    // resolved node index of child a, -1 = not looked up / not in the tree yet
    uint8_t key[kMaxA][kMaxSlots];
    uint8_t done[kMaxA];
};

struct Tree {
    std::vector<Node> nodes;                            // node 0 = the root once it is expanded
    std::unordered_map<std::string, int32_t> index;     // state code -> node
    std::vector<std::pair<int32_t, uint8_t>> trail;     // (node, action) of the last descent
    std::string leaf_key;                               // code of the leaf of the last descent ("" = the root itself)
    std::vector<uint8_t> solution;
    bool solved = false, active = false;
    int32_t sims = 0;
};

}  // namespace

struct rc_tree {
    int n, A, slots;
    double c, vl, vmin;
    bool shared = false, have_rng = false;
    int threads = 1;
    std::vector<Tree> trees;
    std::vector<Mt> rng;
    int last_depth = 0;
};

namespace {

int puct_best(const Node &nd, int A, double c) {
    long total = 0;
    for (int i = 0; i < A; ++i) total += nd.visits[i];
    const double root = std::sqrt((double)total);
    const float c32 = (float)c;
    int arg = 0;
    float best = 0.f;
    for (int i = 0; i < A; ++i) {
        // c * P * (sqrt(total) / (1 + N)) + W - L with numpy >= 2 promotion: every operand becomes float32 at use
        const float cp = c32 * nd.policy[i];
        const float x = (float)(root / (double)(1 + nd.visits[i]));
        const float u = cp * x;
        const float uw = u + (float)nd.value[i];
        const float s = uw - (float)nd.vloss[i];
        if (i == 0 || s > best) { best = s; arg = i; }      // max(range(A), key=...): the first maximum
    }
    return arg;
}

// One descent (mcts.py:52-81).  `gen` == nullptr: stop BEFORE the first step that needs a random draw (a node whose
// visit counts are all zero) and return that node; the caller resumes there with the generator.  The steps before it
// consume no randomness and touch only this root's tree, so they can run for all roots in parallel even when the
// roots share one generator that has to be consumed in root order.
int32_t descend_from(rc_tree *t, Tree &tr, int32_t cur, Mt *gen) {
    while (cur >= 0) {
        Node &nd = tr.nodes[(size_t)cur];
        long total = 0;
        for (int i = 0; i < t->A; ++i) total += nd.visits[i];
        if (total == 0 && gen == nullptr) return cur;
        const int a = total == 0 ? gen->below(t->A) : puct_best(nd, t->A, t->c);   // mcts.py:69-72
        nd.vloss[a] += t->vl;
        tr.trail.emplace_back(cur, (uint8_t)a);
        int32_t nxt = nd.child[a];
        if (nxt < 0) {
            const std::string key(reinterpret_cast<const char *>(nd.key[a]), (size_t)t->slots);
            const auto it = tr.index.find(key);
            if (it != tr.index.end()) nxt = nd.child[a] = it->second;
            else tr.leaf_key = key;
        }
        cur = nxt;
    }
    return -1;
}

int32_t descend_begin(rc_tree *t, int r, Mt *gen) {
    Tree &tr = t->trees[(size_t)r];
    tr.trail.clear();
    tr.leaf_key.clear();
    tr.active = !tr.solved;
    if (!tr.active) return -1;
    ++tr.sims;
    return descend_from(t, tr, tr.nodes.empty() ? -1 : 0, gen);       // the root is node 0 once expanded
}

}  // namespace

#ifndef RC_SRC_HASH
#define RC_SRC_HASH unhashed
#endif
#define RC_STR2(x) #x
#define RC_STR(x) RC_STR2(x)
static const char k_build_id[] = "rc-build-id:" RC_STR(RC_SRC_HASH);

extern "C" {

const char *rc_tree_build_id(void) { return k_build_id + 12; }

rc_tree *rc_tree_create(int n_roots, int n_actions, int n_slots, double cpuct, double virtual_loss, double value_min) {
    if (n_roots <= 0 || n_actions <= 0 || n_actions > kMaxA || n_slots <= 0 || n_slots > kMaxSlots) return nullptr;
    rc_tree *t = new rc_tree;
    t->n = n_roots; t->A = n_actions; t->slots = n_slots;
    t->c = cpuct; t->vl = virtual_loss; t->vmin = value_min;
    t->trees.resize((size_t)n_roots);
    return t;
}

void rc_tree_destroy(rc_tree *t) { delete t; }

int rc_tree_set_threads(rc_tree *t, int threads) {
    if (!t || threads < 1) return -1;
    const int mx = omp_get_max_threads();
    t->threads = threads < mx ? threads : mx;
    return t->threads;
}

int rc_tree_set_rng(rc_tree *t, int shared, const uint32_t *states) {
    if (!t || !states) return -1;
    t->shared = shared != 0;
    t->rng.resize(t->shared ? 1 : (size_t)t->n);
    for (size_t g = 0; g < t->rng.size(); ++g) {
        memcpy(t->rng[g].mt, states + g * 625, 624 * sizeof(uint32_t));
        t->rng[g].idx = states[g * 625 + 624];
        if (t->rng[g].idx > 624) return -1;
    }
    t->have_rng = true;
    return 0;
}

int rc_tree_get_rng(const rc_tree *t, uint32_t *states) {
    if (!t || !states || !t->have_rng) return -1;
    for (size_t g = 0; g < t->rng.size(); ++g) {
        memcpy(states + g * 625, t->rng[g].mt, 624 * sizeof(uint32_t));
        states[g * 625 + 624] = t->rng[g].idx;
    }
    return 0;
}

int rc_tree_select(rc_tree *t) {
    if (!t || !t->have_rng) return -1;
    if (t->shared) {
        std::vector<int32_t> resume((size_t)t->n);
#pragma omp parallel for schedule(dynamic, 16) num_threads(t->threads)
        for (int r = 0; r < t->n; ++r) resume[(size_t)r] = descend_begin(t, r, nullptr);   // the draw-free prefix of every descent
        for (int r = 0; r < t->n; ++r)                                                      // one generator: strictly in root order
            if (resume[(size_t)r] >= 0) descend_from(t, t->trees[(size_t)r], resume[(size_t)r], &t->rng[0]);
    } else {
#pragma omp parallel for schedule(dynamic, 16) num_threads(t->threads)
        for (int r = 0; r < t->n; ++r) descend_begin(t, r, &t->rng[(size_t)r]);
    }
    int depth = 0;
    for (const Tree &tr : t->trees)
        if (tr.active && (int)tr.trail.size() > depth) depth = (int)tr.trail.size();
    t->last_depth = depth;
    return depth;
}

int rc_tree_paths(const rc_tree *t, uint8_t *paths, int pitch) {
    if (!t || !paths || pitch < t->last_depth) return -1;
    for (int r = 0; r < t->n; ++r) {
        uint8_t *row = paths + (size_t)r * (size_t)pitch;
        memset(row, t->A, (size_t)pitch);
        const Tree &tr = t->trees[(size_t)r];
        if (tr.active)
            for (size_t d = 0; d < tr.trail.size(); ++d) row[d] = tr.trail[d].second;
    }
    return 0;
}

int rc_tree_update(rc_tree *t, const uint8_t *leaf_code, const uint8_t *child_code, const uint8_t *solved, const float *value,
                   const float *policy) {
    if (!t || !leaf_code || !child_code || !solved || !value || !policy) return -1;
    const int A = t->A, SL = t->slots;
#pragma omp parallel for schedule(dynamic, 16) num_threads(t->threads)
    for (int r = 0; r < t->n; ++r) {
        Tree &tr = t->trees[(size_t)r];
        if (!tr.active) continue;
        Node nd;
        bool any = false;
        int first = -1;
        for (int a = 0; a < A; ++a) {
            nd.policy[a] = policy[(size_t)r * A + a];
            nd.value[a] = t->vmin;
            nd.vloss[a] = 0.0;
            nd.visits[a] = 0;
            nd.child[a] = -1;
            memcpy(nd.key[a], child_code + ((size_t)r * A + a) * SL, (size_t)SL);
            nd.done[a] = solved[(size_t)r * A + a] != 0;
            if (nd.done[a] && !any) { any = true; first = a; }
        }
        const int32_t id = (int32_t)tr.nodes.size();
        tr.nodes.push_back(nd);
        const std::string own(reinterpret_cast<const char *>(leaf_code + (size_t)r * SL), (size_t)SL);
        if (id == 0) tr.index.emplace(own, 0);                      // the root, also reachable under its own code
        else tr.index.emplace(tr.leaf_key, id);                     // the key its parent stored (== own)
        const float v = value[r];
        for (auto it = tr.trail.rbegin(); it != tr.trail.rend(); ++it) {           // mcts.py:115-130
            Node &p = tr.nodes[(size_t)it->first];
            const int a = it->second;
            if ((double)v > p.value[a]) p.value[a] = (double)v;      // max(W, v): keeps W on ties
            p.vloss[a] -= 150.0;                                     // the reference's literal (mcts.py:127)
            p.visits[a] += 1;
        }
        if (any) {                                                   // mcts.py:44-49
            tr.solution.clear();
            for (const auto &pa : tr.trail) tr.solution.push_back(pa.second);
            tr.solution.push_back((uint8_t)first);
            tr.solved = true;
        }
        tr.active = false;
    }
    int done = 0;
    for (const Tree &tr : t->trees) done += tr.solved;
    return done;
}

int rc_tree_solution(const rc_tree *t, int root, uint8_t *out, int cap) {
    if (!t || root < 0 || root >= t->n) return -2;
    const Tree &tr = t->trees[(size_t)root];
    if (!tr.solved) return -1;
    const int len = (int)tr.solution.size();
    if (out)
        for (int i = 0; i < len && i < cap; ++i) out[i] = tr.solution[(size_t)i];
    return len;
}

int rc_tree_sims_used(const rc_tree *t, int32_t *out) {
    if (!t || !out) return -1;
    for (int r = 0; r < t->n; ++r) out[r] = t->trees[(size_t)r].sims;
    return 0;
}

int rc_tree_root_stats(const rc_tree *t, int root, int32_t *visits, double *values) {
    if (!t || root < 0 || root >= t->n) return -1;
    const Tree &tr = t->trees[(size_t)root];
    for (int a = 0; a < t->A; ++a) {
        if (visits) visits[a] = tr.nodes.empty() ? 0 : tr.nodes[0].visits[a];
        if (values) values[a] = tr.nodes.empty() ? -1.0 : tr.nodes[0].value[a];
    }
    return (int)tr.nodes.size();
}

}  // extern "C"
