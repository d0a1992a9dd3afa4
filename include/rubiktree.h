/*
 * rubiktree.h -- C ABI of librubiktree.so: HOST-side bookkeeping of the lockstep tree search (BatchedMCTS).
 *
 * The reference's search (mcts.py, paths relative to /root/reference) keeps, per searched cube, a dict keyed by the
 * printed one-hot state: children keys, policy, max-backed-up values W, visit counts N, virtual losses L, done flags
 * (mcts.py:12-35,83-113); `traverse` walks it with PUCT + virtual loss (mcts.py:52-81,132-154) and `backpropagate`
 * updates it (mcts.py:115-130).  Searching R cubes in lockstep, one simulation is
 *     rc_tree_select    R tree descents on the host (this library)             -> R action paths
 *     librubikhip.so    replay the paths, expand the R leaves, encode them; the caller's value net runs once on them
 *     rc_tree_update    R node insertions + back-propagations on the host (this library)
 * The device step takes ~0.14 ms for 4096 roots; the same bookkeeping in Python takes hundreds of milliseconds.
 *
 * Semantics are the reference's, bit for bit, so that every root of a lockstep search ends exactly like a stand-alone
 * run of mcts.py with the same generator (fixture G8):
 *   - node key = the 20-byte (7-byte) compact one-hot code of a state; states reached along different paths share a
 *     node (the reference's dict has the same effect); the root is also reachable under its own code;
 *   - score_a = c * P_a * (sqrt(sum N) / (1 + N_a)) + W_a - L_a evaluated in the reference's types under numpy >= 2:
 *     float32(c) * P_a (float32), times float32 of the double quotient, plus float32(W_a), minus float32(L_a);
 *     the first maximal action wins (mcts.py:132-154);
 *   - a node whose visit counts are all zero picks random.randint(0, A-1) (mcts.py:69-70): CPython's MT19937 +
 *     _randbelow_with_getrandbits (k = A.bit_length() bits per draw, rejection), continued from a state obtained
 *     with random.Random.getstate(), one generator per root or one shared generator consumed in root order;
 *   - W_a = max(W_a, leaf value) with the leaf value a float32, L_a += virtual_loss on the way down and -= 150 (the
 *     reference's literal, mcts.py:127) on the way up, N_a += 1; a root is finished when an expanded leaf has a
 *     solved child: solution = path + [lowest solved action] (mcts.py:44-49).
 */
#ifndef RUBIKTREE_H
#define RUBIKTREE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rc_tree rc_tree;

/* n_roots independent trees; n_actions = 12 | 6, n_slots = 20 | 7 (key bytes).  NULL on bad arguments. */
/* Identity of the sources this binary was built from (sha256 over rc_tree.cpp + this header, first 16 hex digits, passed as
 * -DRC_SRC_HASH by __graft_entry__.build_tree); "unhashed" otherwise.  Static storage. */
const char *rc_tree_build_id(void);

rc_tree *rc_tree_create(int n_roots, int n_actions, int n_slots, double cpuct, double virtual_loss, double value_min);
void rc_tree_destroy(rc_tree *t);
/* OpenMP threads for select / update (default 1; pass the job's CPU share, not the host's thread count).  Returns the
 * value in effect.  Roots are independent; with a shared generator only the draw-free prefix of each descent runs in
 * parallel and the rest is finished in root order, so results do not depend on the thread count. */
int rc_tree_set_threads(rc_tree *t, int threads);

/* Generator state(s) in random.Random.getstate()[1] layout: 624 MT words + the index (625 uint32).
 * shared = 0: states[n_roots][625], one generator per root (roots may then be processed in parallel);
 * shared = 1: states[1][625], ONE generator consumed by the roots in index order (the global `random` module). */
int rc_tree_set_rng(rc_tree *t, int shared, const uint32_t *states);
int rc_tree_get_rng(const rc_tree *t, uint32_t *states);

/* One descent per unfinished root (mcts.py:52-81).  Returns the longest path length (>= 0), negative on error. */
int rc_tree_select(rc_tree *t);
/* The paths of the last rc_tree_select: paths[r * pitch + d] = action d of root r, padded with n_actions (the no-op of
 * librubikhip's kernels); finished roots get an all-no-op row.  pitch >= the value rc_tree_select returned. */
int rc_tree_paths(const rc_tree *t, uint8_t *paths, int pitch);

/* Insert the leaves reached by the last select and back-propagate (mcts.py:83-130).  Row r describes root r's leaf:
 * leaf_code [n_roots][n_slots], child_code [n_roots][n_actions][n_slots], solved [n_roots][n_actions] (0 | 1),
 * value [n_roots] (float32 leaf values), policy [n_roots][n_actions] (softmax output, float32).
 * Returns the number of finished roots so far, negative on error. */
int rc_tree_update(rc_tree *t, const uint8_t *leaf_code, const uint8_t *child_code, const uint8_t *solved,
                   const float *value, const float *policy);

/* Results.  rc_tree_solution: action list of root r into out (capacity cap), returns its length or -1 if unfinished. */
int rc_tree_solution(const rc_tree *t, int root, uint8_t *out, int cap);
int rc_tree_sims_used(const rc_tree *t, int32_t *out /* [n_roots] */);
/* Visit counts and values of the root node's edges (0 / -1 if the root is not expanded yet); returns node count of the tree. */
int rc_tree_root_stats(const rc_tree *t, int root, int32_t *visits /* [A] */, double *values /* [A] */);

#ifdef __cplusplus
}
#endif
#endif /* RUBIKTREE_H */
