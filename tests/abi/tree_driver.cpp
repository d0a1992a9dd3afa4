// tree_driver.cpp -- a plain C++ consumer of include/rubiktree.h with no Python in the process: the lockstep search's host
// bookkeeping (rc_tree_select / rc_tree_update, mirroring /root/reference/mcts.py:52-154) driven by a synthetic "device step"
// (leaf key, child keys, value, policy and rare solved children are a pure function of the action path, with transpositions).
//
// Two uses:
//   * tests/test_tree_native.py builds it with g++ and runs it: results must not depend on the OpenMP thread count, with one
//     generator per root and with the shared generator consumed in root order;
//   * tools/sanitize_cpu.sh builds it together with rc_tree.cpp under -fsanitize=address,undefined (g++) and under
//     -fsanitize=thread (clang++ / libomp + its TSan-aware tool library), where a Python host process would drown the report.
//
// usage: tree_driver [roots] [simulations] [threads]     prints "tree_driver ok ..." and exits 0, or a diagnostic and exits 1
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "rubiktree.h"

namespace {

constexpr int A = 12, SL = 20;

uint64_t mix(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// a move followed by its inverse (a ^ 1) cancels: states reached along different paths share a key (transpositions)
int normalise(const uint8_t *path, int len, uint8_t *out) {
    int n = 0;
    for (int i = 0; i < len; ++i) {
        if (path[i] >= A) continue;                     // the no-op padding
        if (n && out[n - 1] == (path[i] ^ 1)) --n;
        else out[n++] = path[i];
    }
    return n;
}

void key_of(const uint8_t *norm, int n, uint8_t *key) {
    std::memset(key, 0, SL);
    const int from = n > SL ? n - SL : 0;
    for (int i = from; i < n; ++i) key[i - from] = (uint8_t)(norm[i] + 1);
}

struct Result {
    std::vector<int32_t> sims, visits;
    std::vector<double> values;
    std::vector<std::vector<uint8_t>> solutions;
    std::vector<uint32_t> rng;
};

bool run(int roots, int sims, int threads, int shared, Result &res) {
    rc_tree *t = rc_tree_create(roots, A, SL, 1.0, 150.0, -10.0);
    if (!t) return false;
    if (rc_tree_set_threads(t, threads) < 1) return false;
    const int gens = shared ? 1 : roots;
    std::vector<uint32_t> st((size_t)gens * 625);
    for (int g = 0; g < gens; ++g) {
        for (int i = 0; i < 624; ++i) st[(size_t)g * 625 + i] = (uint32_t)mix((uint64_t)g * 1000003u + i + (shared ? 77 : 0));
        st[(size_t)g * 625 + 624] = 624;                // index: the next draw regenerates the block
    }
    if (rc_tree_set_rng(t, shared, st.data()) != 0) return false;
    std::vector<uint8_t> paths, leaf((size_t)roots * SL), child((size_t)roots * A * SL), solved((size_t)roots * A), norm(4096);
    std::vector<float> value(roots), policy((size_t)roots * A);
    for (int s = 0; s < sims; ++s) {
        const int depth = rc_tree_select(t);
        if (depth < 0) return false;
        const int pitch = depth > 0 ? depth : 1;
        paths.assign((size_t)roots * pitch, (uint8_t)A);
        if (rc_tree_paths(t, paths.data(), pitch) != 0) return false;
        for (int r = 0; r < roots; ++r) {
            const int n = normalise(&paths[(size_t)r * pitch], depth, norm.data());
            key_of(norm.data(), n, &leaf[(size_t)r * SL]);
            uint64_t h = mix((uint64_t)r * 2654435761u);
            for (int i = 0; i < n; ++i) h = mix(h ^ norm[i]);
            value[r] = (float)((double)(h >> 40) / (double)(1 << 24) * 2.0 - 1.0);
            float sum = 0.f;
            for (int a = 0; a < A; ++a) {
                norm[n] = (uint8_t)a;
                uint8_t tmp[4096];
                const int m = normalise(norm.data(), n + 1, tmp);
                key_of(tmp, m, &child[((size_t)r * A + a) * SL]);
                const uint64_t ha = mix(h + 31 * (a + 1));
                policy[(size_t)r * A + a] = 0.05f + (float)(ha & 1023) / 1024.f;
                sum += policy[(size_t)r * A + a];
                solved[(size_t)r * A + a] = (uint8_t)(n >= 3 && ha % 701 == 0);
            }
            for (int a = 0; a < A; ++a) policy[(size_t)r * A + a] /= sum;
        }
        if (rc_tree_update(t, leaf.data(), child.data(), solved.data(), value.data(), policy.data()) < 0) return false;
    }
    res.sims.resize(roots);
    rc_tree_sims_used(t, res.sims.data());
    res.visits.resize((size_t)roots * A);
    res.values.resize((size_t)roots * A);
    res.solutions.resize(roots);
    for (int r = 0; r < roots; ++r) {
        rc_tree_root_stats(t, r, &res.visits[(size_t)r * A], &res.values[(size_t)r * A]);
        uint8_t buf[4096];
        const int k = rc_tree_solution(t, r, buf, (int)sizeof(buf));
        if (k >= 0) res.solutions[r].assign(buf, buf + (k < (int)sizeof(buf) ? k : (int)sizeof(buf)));
        else res.solutions[r].assign(1, (uint8_t)255);
    }
    res.rng.resize((size_t)gens * 625);
    rc_tree_get_rng(t, res.rng.data());
    rc_tree_destroy(t);
    return true;
}

bool same(const Result &a, const Result &b) {
    return a.sims == b.sims && a.visits == b.visits && a.values == b.values && a.solutions == b.solutions && a.rng == b.rng;
}

}  // namespace

int main(int argc, char **argv) {
    const int roots = argc > 1 ? std::atoi(argv[1]) : 96, sims = argc > 2 ? std::atoi(argv[2]) : 40, threads = argc > 3 ? std::atoi(argv[3]) : 4;
    if (std::strlen(rc_tree_build_id()) == 0) return 1;
    long finished = 0, total_sims = 0;
    for (int shared = 0; shared < 2; ++shared) {
        Result one, many;
        if (!run(roots, sims, 1, shared, one) || !run(roots, sims, threads, shared, many)) {
            std::printf("tree_driver: a call failed (shared=%d)\n", shared);
            return 1;
        }
        if (!same(one, many)) {
            std::printf("tree_driver: results differ between 1 and %d threads (shared=%d)\n", threads, shared);
            return 1;
        }
        for (int r = 0; r < roots; ++r) {
            finished += many.solutions[r].size() != 1 || many.solutions[r][0] != 255;
            total_sims += many.sims[r];
        }
    }
    if (finished == 0 || finished == 2L * roots) {      // the synthetic step must exercise both finished and unfinished roots
        std::printf("tree_driver: degenerate workload (%ld of %d finished)\n", finished, 2 * roots);
        return 1;
    }
    std::printf("tree_driver ok: %d roots x %d simulations, 1 vs %d threads identical (per-root and shared generators), %ld finished, %ld simulations used\n",
                roots, sims, threads, finished, total_sims);
    return 0;
}
