/*
 * rubikhip.h -- C ABI of librubikhip.so: the MI355X (gfx950) cube-environment hot path.
 *
 * The reference (SUNGBEOMCHOI/Rubiks-Cube-Solver) has no FFI: its env path is Python
 * (SURVEY.md section 8b).  Each entry point below names the reference code it replaces,
 * paths relative to /root/reference.  A binding only needs plain pointers and sizes; see
 * INTEGRATION.md for the ctypes stub a maintainer of the reference would add.
 *
 * Conventions
 *   - All buffers are DEVICE memory owned by the caller (e.g. PyTorch-ROCm tensors); the
 *     library allocates nothing but its constant tables and one status word per device (rc_apply_moves_ws takes a
 *     caller-owned workspace).
 *   - State layout.  Cube states are structure-of-arrays uint8, values 0..5, S = 54
 *     (cube_size 3) or 24 (cube_size 2) rows, in TILES of `pitch` cubes:
 *         sticker s of cube n lives at st[(n / pitch) * S * pitch + s * pitch + n % pitch].
 *     One tile (pitch >= n_cubes, pitch % 16 == 0) is plain SoA, st[s * pitch + n].  Several
 *     tiles need a power-of-two pitch >= 512 (the widest wave span); 16384-32768 measured best on MI355X at 4M cubes
 *     (+12..24 % HBM throughput over one 4M-wide tile: every wave's 54 row segments then sit
 *     within 54 * pitch bytes).  RC_FMT_CODE buffers are tiled the same way with SLOTS rows.
 *     Buffers described as "plain" below are one tile.  Base pointers are 16-byte aligned.
 *   - actions are uint8 in the env's action order U,U',F,F',R,R'[,D,D',B,B',L,L']
 *     (gym-cube/gym_cube/envs/cube_env.py:24-28); A = 12 | 6.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls are
 *     stream-ordered, never synchronise (except rc_read_status, rc_facade_step(wait=1) and
 *     rc_get_tables), are re-entrant, and keep no mutable state besides the per-device status word
 *     (one word per DEVICE, shared by all streams).
 *   - Row offsets inside one tile are 32-bit scalar offsets of buffer instructions: rows * pitch
 *     must stay below 2^32 (a single tile of more than ~79 M 3x3x3 cubes has to be split into tiles).
 *   - Return value: RC_OK or a negative RC_E* code; rc_last_error() gives the message of
 *     the calling thread's last failure.  Nothing is thrown across the ABI.
 *   - The action value A itself (12 | 6) is a NO-OP: the cube is left unchanged (its done /
 *     reward / one-hot are still written).  Batched rollouts use it to park finished cubes and to
 *     pad scrambles of different lengths.  The reference has no such action.
 *   - Out-of-range actions (> A) are the reference's IndexError (cube_env.py:86,96).  On the
 *     device they cannot raise: the cube is left in an unspecified (in-bounds) state and
 *     bit RC_STATUS_BAD_ACTION of the status word is set; rc_read_status() reports it.
 *
 * One-hot formats (`fmt`)
 *   RC_FMT_NONE   nothing written
 *   RC_FMT_CODE   compact, lossless: uint8 code[slot * pitch + n] (tiled like states), slot = 0..19 | 0..6,
 *                 code = piece*3+ori (corner slots) or piece*2+ori (edge slots)
 *                 (getOP_3, assets/py333.py:224-227).  3x3x3: one-hot column of row `slot`
 *                 (pos_to_state_3, py333.py:235-246).  2x2x2: row = code/3,
 *                 column = slot*3 + code%3 (cube_env.py:142-147).
 *   RC_FMT_U8 / RC_FMT_F16 / RC_FMT_BF16 / RC_FMT_F32
 *                 dense [n][R][C] (R,C = 20,24 | 7,21), contiguous per cube, values {0,1}:
 *                 exactly what model.py:31-45 consumes after `.float()`.
 */
#ifndef RUBIKHIP_H
#define RUBIKHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RC_OK 0
#define RC_EINVAL (-1)    /* bad argument (null, alignment, cube_size, fmt, sizes) */
#define RC_EHIP (-2)      /* a HIP runtime call failed */
#define RC_ENODEV (-3)    /* rc_init not called / no gfx950 device */

#define RC_FMT_NONE 0
#define RC_FMT_CODE 1
#define RC_FMT_U8 2
#define RC_FMT_F16 3
#define RC_FMT_F32 4
#define RC_FMT_BF16 5

#define RC_STATUS_BAD_ACTION 1u

/* Library / ABI version (major*100 + minor). */
int rc_version(void);

/* Identity of the sources this binary was built from: the first 16 hex digits of a sha256 over rubikhip.hip, rc_device.h,
 * rc_tables.h and this header (rubiks-cube-solver_amd/_build.py), passed as -DRC_SRC_HASH at build time; "unhashed" for a build
 * that was not given one.  The Python binding compares it with the tree's sources and refuses a stale library.  Static storage.
 * (Tooling; no reference counterpart.) */
const char *rc_build_id(void);

/* Check that `device` is a gfx950 GPU, make the code object resident there and clear its status
 * word.  Idempotent; the caller's current device is not changed (kernels run on the device of the
 * stream / current device the caller set, one process per GPU).  Replaces: importing py333 (module-level table construction,
 * gym-cube/gym_cube/envs/assets/py333.py:41-198). */
int rc_init(int device);

/* Host copies of the tables the kernels use (any pointer may be NULL).
 *   perm        [A][S]   new[i] = old[perm[a][i]]            (py333.py:46-138 moveDefs)
 *   solved      [S]                                           (py333.py:211-218)
 *   corner_defs [NC][3], edge_defs [NE][2]                    (py333.py:140-164)
 *   corner_code [72], edge_code [72]  hash -> piece*3+ori / piece*2+ori, 0 where the
 *                reference table has no entry or ends        (py333.py:171-198)
 * dims, if non-NULL, receives {S, A, NC, NE, R, C}. */
int rc_get_tables(int cube_size, uint8_t *perm, uint8_t *solved, uint8_t *corner_defs,
                  uint8_t *edge_defs, uint8_t *corner_code, uint8_t *edge_code, int32_t dims[6]);

/* st[s][n] = s / face_size for n < n_cubes.  Replaces initState_3 (py333.py:211-218) /
 * CubeEnv.init_state (cube_env.py:33-42). */
int rc_fill_solved(uint8_t *st, int64_t n_cubes, int64_t pitch, int cube_size, void *stream);

/* One env step for n_cubes independent cubes: out = in moved by actions[n], then
 * done[n] = solved, reward[n] = done ? +1.0f : -1.0f, and the one-hot of the NEW state.
 * `out` may alias `in` (in-place); reward, done, onehot may be NULL (onehot NULL requires
 * fmt RC_FMT_NONE).  code_pitch is the row pitch of an RC_FMT_CODE buffer (ignored for
 * dense formats).  Replaces CubeEnv.step (cube_env.py:71-111) = doMove_3 (py333.py:220-222)
 * + sim_state_to_state (cube_env.py:132-152) + isSolved_3 (py333.py:229-233). */
int rc_apply_moves(const uint8_t *in, uint8_t *out, const uint8_t *actions, int64_t n_cubes,
                   int64_t pitch_in, int64_t pitch_out, int cube_size, float *reward,
                   uint8_t *done, void *onehot, int fmt, int64_t code_pitch, void *stream);
/* rc_apply_moves with a caller-owned WORKSPACE (device memory, 16-byte aligned, rc_workspace_bytes(RC_OP_STEP, ...) bytes; its
 * contents are scratch; it must not overlap in, out, actions, reward, done or onehot: RC_EINVAL).  Results are identical to rc_apply_moves.  With a dense `fmt` on large 3x3x3 batches the workspace lets the
 * call run as two launches -- step + reward + done + compact code into the workspace, then the front writer
 * (k_code_to_dense_front) -- so that the dense stream leaves the chip as one sweeping window: 0.80-0.94 of the HBM peak on every
 * allocation instead of 0.67-0.87 depending on where the buffer lives (DESIGN.md "Dense one-hot").  A NULL or too small workspace,
 * or a call that has no use for one (rc_workspace_bytes() == 0), runs exactly rc_apply_moves.  The library still allocates nothing. */
int64_t rc_workspace_bytes(int op, int cube_size, int64_t n_cubes, int fmt);
int rc_apply_moves_ws(const uint8_t *in, uint8_t *out, const uint8_t *actions, int64_t n_cubes,
                      int64_t pitch_in, int64_t pitch_out, int cube_size, float *reward, uint8_t *done,
                      void *onehot, int fmt, int64_t code_pitch, void *workspace, int64_t workspace_bytes, void *stream);
/* rc_encode with a caller-owned workspace (the same rc_workspace_bytes(RC_OP_STEP, ...) bytes): a dense one-hot of a large 3x3x3
 * batch is then encoded into the workspace as compact codes and expanded by the front writer.  Results identical to rc_encode. */
int rc_encode_ws(const uint8_t *st, int64_t n_cubes, int64_t pitch, int cube_size, void *onehot, int fmt, int64_t code_pitch,
                 void *workspace, int64_t workspace_bytes, void *stream);
/* Same with a per-call tuning override (see "Tuning override" at the end; 0 = rc_apply_moves). */
int rc_apply_moves_ex(const uint8_t *in, uint8_t *out, const uint8_t *actions, int64_t n_cubes,
                      int64_t pitch_in, int64_t pitch_out, int cube_size, float *reward,
                      uint8_t *done, void *onehot, int fmt, int64_t code_pitch, void *stream, int variant);

/* CubeEnv.step for ONE cube with the lowest host latency (cube_env.py:71-111; the batch-1 facade):
 * st is a device state buffer whose cube 0 is stepped in place (sticker s at st[s * pitch]); the action
 * travels by value; the kernel writes into `host_out`, 512 bytes of HOST-MAPPED pinned memory
 * (hipHostMalloc / torch pin_memory): [0, R*C) the dense uint8 one-hot of the new state, [496] done,
 * [504..507] `seq` (non-zero, written last after a system-scope fence).  The kernel writes through the buffer's DEVICE alias
 * (hipPointerGetAttributes), the host polls the host address, so hipHostRegister'ed memory works too.  wait != 0: returns once `seq`
 * is visible in host_out (spin on the flag; no stream synchronisation, no copies); wait == 0: returns
 * after the launch and the caller polls host_out itself.  Out-of-range actions set RC_STATUS_BAD_ACTION. */
int rc_facade_step(uint8_t *st, int64_t pitch, int cube_size, int action, uint8_t *host_out,
                   uint32_t seq, int wait, void *stream);
/* Same for a SEQUENCE of moves (`actions`: n_actions bytes in HOST memory, copied into the launch arguments; n_actions
 * may be 0 = just report the current state): one launch per 60 moves instead of one per move; host_out describes the
 * final state.  This is a whole tree descent of MCTS.traverse (mcts.py:52-81, one env.step per level there). */
int rc_facade_steps(uint8_t *st, int64_t pitch, int cube_size, const uint8_t *actions, int n_actions,
                    uint8_t *host_out, uint32_t seq, int wait, void *stream);

/* Node expansion of a single-root tree search for ONE cube (cube 0 of st; MCTS.expand, mcts.py:83-113: the
 * reference does 12 env.step + 13 deepcopy(env) per leaf; also the child loop of get_target_value,
 * cube_env.py:212-236).  One launch, results in `host_out` (host-mapped pinned memory, as above; 512 bytes,
 * or 512 + A*R*C = 6272 with dense != 0): [0, SLOTS) the cube's own compact code (RC_FMT_CODE bytes: the node
 * key), [32 + a * SLOTS, ...) the code of child a, [288 + a] solved flag of child a, [504..507] `seq`,
 * and with dense != 0 [512 + a * R*C, ...) the dense uint8 one-hot of child a. */
int rc_facade_expand(const uint8_t *st, int64_t pitch, int cube_size, uint8_t *host_out, uint32_t seq,
                     int dense, int wait, void *stream);

/* The facade entry points cache the device alias of every host_out they validated, process-wide, keyed on (host address, current
 * device); a call with seq == 1 (a caller's first use of a buffer) always validates afresh.  Call this before freeing a host_out
 * buffer, from ANY thread (NULL drops everything cached); CubeEnv.close() / garbage collection does. */
int rc_facade_release(const uint8_t *host_out);

/* Device address of HOST-MAPPED pinned memory (hipHostMalloc / hipHostRegister / torch pin_memory), for callers that let a kernel
 * read a small input straight from host memory instead of uploading it first: e.g. the action paths of a lockstep tree search, written
 * by the host trees and replayed by rc_scramble(actions_in = the alias) -- one H2D copy less per simulation.  RC_EINVAL for other memory. */
int rc_host_alias(const void *host, void **device_alias);

/* `depth` moves applied in place to every cube: the scramble loop of CubeEnv.reset
 * (cube_env.py:65-67) for n_cubes cubes at once.  actions_in[d * act_pitch + n] replays given
 * moves (e.g. the host's legacy-numpy draws, for bit-exact reset(seed)); NULL draws them on
 * the device exactly like rc_adi_generate (walk = walk_offset + n).  actions_out (same layout,
 * NULL to skip) receives the moves used.  done/reward (NULL to skip) describe the final state. */
int rc_scramble(uint8_t *st, int64_t n_cubes, int64_t pitch, int cube_size, int depth, uint64_t seed,
                uint64_t stream_id, int64_t walk_offset, const uint8_t *actions_in, uint8_t *actions_out,
                int64_t act_pitch, uint8_t *done, float *reward, void *stream);

/* rc_scramble that reads the start states from `src` (same layout and pitch as st) and leaves them untouched: st = src moved.  The
 * lockstep tree search replays every simulation's descents from its root states this way (mcts.py:37,80: one deepcopy + one env.step
 * per tree level there).  depth == 0 copies. */
int rc_scramble_from(const uint8_t *src, uint8_t *st, int64_t n_cubes, int64_t pitch, int cube_size, int depth,
                     uint64_t seed, uint64_t stream_id, int64_t walk_offset, const uint8_t *actions_in,
                     uint8_t *actions_out, int64_t act_pitch, uint8_t *done, float *reward, void *stream);

/* One record per root for the host trees of a lockstep search (include/rubiktree.h rc_tree_update; mcts.py:96-101 keeps the same three
 * things per expanded node): from the leaves' codes (RC_FMT_CODE layout, [tile][SLOTS][pitch]) and the outputs of rc_expand_children
 * with pitch_out = pitch (child_code [A][tiles][SLOTS][pitch], child_solved [A][tiles * pitch]) to
 * leaf_out [n][SLOTS], child_out [n][A][SLOTS], solved_out [n][A], contiguous. */
int rc_search_pack(const uint8_t *leaf_code, const uint8_t *child_code, const uint8_t *child_solved, int64_t n_cubes,
                   int64_t pitch, int cube_size, uint8_t *leaf_out, uint8_t *child_out, uint8_t *solved_out, void *stream);

/* The scramble draws of CubeEnv.reset(seed, k) (cube_env.py:62-68) for n envs at once, bit for bit:
 * np.random.seed(seeds[i]); np.random.randint(action_dim, size=k_i) of numpy's LEGACY generator
 * (MT19937 + masked rejection) runs on the device, one env per lane.  k_i = counts[i] (device memory;
 * values outside 0..kmax are clamped to that range) or count_uniform when counts is NULL.  actions_out[d * pitch + i] for d < k_i, the no-op value A
 * for k_i <= d < kmax: ready to be replayed by rc_scramble(actions_in).  pitch >= n, pitch % 16 == 0. */
int rc_legacy_scramble_actions(const uint32_t *seeds, const int32_t *counts, int count_uniform, int kmax,
                               int64_t n_envs, int cube_size, uint8_t *actions_out, int64_t pitch,
                               void *stream);
/* Same with the generator form chosen by the caller (RC_VARIANT_LEGACY_*; 0 = rc_legacy_scramble_actions).  Two forms produce the
 * same bytes: the LDS form keeps MT19937's 624-word state per env in LDS (any count; the twist is done lazily, 64 words at a time);
 * the STREAMING form computes outputs 0..622 of the first generation from a few init_genrand chain iterators in registers (no state at
 * all) and is followed by a fix-up launch of the LDS form for the waves in which a lane needed more -- the default up to kmax = 400. */
int rc_legacy_scramble_actions_ex(const uint32_t *seeds, const int32_t *counts, int count_uniform, int kmax,
                                  int64_t n_envs, int cube_size, uint8_t *actions_out, int64_t pitch,
                                  void *stream, int variant);

/* done / reward of the given states, no move.  Replaces isSolved_3 (py333.py:229-233). */
int rc_is_solved(const uint8_t *st, int64_t n_cubes, int64_t pitch, int cube_size,
                 uint8_t *done, float *reward, void *stream);

/* One-hot of the given states, no move.  Replaces sim_state_to_state (cube_env.py:132-152)
 * = pos_to_state_3(getOP_3(s)) (py333.py:224-246). */
int rc_encode(const uint8_t *st, int64_t n_cubes, int64_t pitch, int cube_size, void *onehot,
              int fmt, int64_t code_pitch, void *stream);

/* Dense one-hot [n][R][C] from a compact code buffer (RC_FMT_CODE layout). fmt = U8/F16/BF16/F32. */
int rc_onehot_from_code(const uint8_t *code, int64_t n_cubes, int64_t code_pitch, int cube_size,
                        void *onehot, int fmt, void *stream);
int rc_onehot_from_code_ex(const uint8_t *code, int64_t n_cubes, int64_t code_pitch, int cube_size,
                           void *onehot, int fmt, void *stream, int variant);
/* The same for n_blocks EQUALLY TILED code buffers in one launch: block b is read at code + b * src_block_stride bytes (n_cubes cubes,
 * tiling code_pitch) and written at onehot + b * dst_block_stride cubes (dst_block_stride >= n_cubes; every block 16-byte aligned).
 * What the 2x2x2 ADI pipeline needs: the A child-code buffers of each depth ([depth][A][tile][SLOTS][pitch] of rc_adi_generate) become
 * the packed [depth * A][dst_block_stride] input of ONE forward of the value net (cube_env.py:239-251 evaluates 12 + 1 rows at a time)
 * instead of blocks padded to whole tiles (the 3x3x3 pipeline uses rc_onehot_from_family_depths). */
int rc_onehot_from_code_blocks(const uint8_t *code, int64_t n_cubes, int64_t code_pitch, int cube_size, void *onehot, int fmt,
                               int n_blocks, int64_t src_block_stride, int64_t dst_block_stride, void *stream);

/* All A children of every cube.  Outputs use one tiling (pitch_out, tiles = ceil(n / pitch_out)
 * when n > pitch_out, else 1), child-major:
 *   children     [A][tiles][S][pitch_out]        child a = a tiled state buffer of its own
 *   child_code   [A][tiles][SLOTS][pitch_out]
 *   child_solved [A][tiles * pitch_out]          flag of child a of cube n at [a][n]
 * (any may be NULL).  With one tile this is children[(a*S + s) * pitch_out + n].  `in` may be tiled too.
 * Replaces the child loops of CubeEnv.get_target_value (cube_env.py:212-236) and MCTS.expand (mcts.py:96-101). */
int rc_expand_children(const uint8_t *in, int64_t n_cubes, int64_t pitch_in, int cube_size,
                       uint8_t *children, uint8_t *child_solved, uint8_t *child_code,
                       int64_t pitch_out, void *stream);
int rc_expand_children_ex(const uint8_t *in, int64_t n_cubes, int64_t pitch_in, int cube_size,
                          uint8_t *children, uint8_t *child_solved, uint8_t *child_code,
                          int64_t pitch_out, void *stream, int variant);

/* ADI scramble generator: n_walks random walks of `depth` moves from the solved cube, each
 * step expanded to all A children.  Replaces the env work of CubeEnv.get_random_samples
 * (cube_env.py:177-194) + get_target_value's child loop (cube_env.py:212-236).
 *   actions_in   NULL: moves are drawn on the device, walk w (global index walk_offset + w)
 *                uses xoroshiro128+ seeded by splitmix64 from (seed, stream_id, walk)
 *                (DESIGN.md "RNG"); non-NULL: replay actions_in[d * pitch + w].
 * Layouts, with tiles = ceil(n_walks / pitch) when n_walks > pitch (then pitch is a power of two
 * >= 512), else 1, and Wp = tiles * pitch (any pointer may be NULL to skip that output):
 *   actions_in / actions_out  [depth][Wp]
 *   parents      [depth][tiles][S][pitch]          one tiled state buffer per depth
 *   parent_code  [depth][tiles][SLOTS][pitch]
 *   children     [depth][A][tiles][S][pitch]       one tiled state buffer per (depth, child)
 *   child_code   [depth][A][tiles][SLOTS][pitch]
 *   child_solved [depth][A][Wp]
 * With one tile these are the plain [depth][A][S][pitch] arrays. */
int rc_adi_generate(uint64_t seed, uint64_t stream_id, int64_t walk_offset, int64_t n_walks,
                    int depth, int cube_size, int64_t pitch, const uint8_t *actions_in,
                    uint8_t *actions_out, uint8_t *parents, uint8_t *parent_code,
                    uint8_t *children, uint8_t *child_code, uint8_t *child_solved, void *stream);
int rc_adi_generate_ex(uint64_t seed, uint64_t stream_id, int64_t walk_offset, int64_t n_walks,
                       int depth, int cube_size, int64_t pitch, const uint8_t *actions_in,
                       uint8_t *actions_out, uint8_t *parents, uint8_t *parent_code,
                       uint8_t *children, uint8_t *child_code, uint8_t *child_solved, void *stream,
                       int variant);

/* ADI target assembly for `n` parents with A children each.
 * Replaces cube_env.py:229-232,239-251:
 *   if any child is solved: target_value = 1.0, target_policy = lowest solved index;
 *   else target_value = max_a(child_value[a] + (-1.0f)), target_policy = first arg max;
 *   error = |double(parent_value) - double(target_value)| * weight[n]   (weight = d**-T, host).
 * child_value[a * pitch + n] (float), child_solved[a * pitch + n], parent_value[n]. */
/* The same walks with the FAMILY record instead of parent / child codes (3x3x3 and 2x2x2): a face turn carries whole cubies, so every
 * slot code of a parent and of its A children is one of NF shared look-ups -- 51 for the 3x3x3 (31 corner + 20 edge (slot, reading
 * order) pairs), 15 for the 2x2x2.  family[((d * tiles + tile) * NF + row) * pitch + walk] holds them (tiled like the codes, NF rows):
 * 51 bytes per (walk, depth) instead of 13 x 20.  rc_family_layout gives NF and rows[(A + 1)][SLOTS]: the family row that IS slot p's
 * code of child a (a = A: the parent) -- the pick of cube_env.py:212-236 / py333.py:224-227 as a table.  rc_onehot_from_family
 * (3x3x3) expands one depth's rows ([tile][NF][pitch], n walks) to the dense one-hots of all A children and the parent in one
 * launch: child a at onehot + a * block_stride cubes, the parent as block A (block_stride >= n). */
int rc_adi_generate_family(uint64_t seed, uint64_t stream_id, int64_t walk_offset, int64_t n_walks,
                           int depth, int cube_size, int64_t pitch, const uint8_t *actions_in,
                           uint8_t *actions_out, uint8_t *parents, uint8_t *family, uint8_t *child_solved,
                           void *stream, int variant);
int rc_family_layout(int cube_size, uint8_t *rows, int32_t *n_rows);
int rc_onehot_from_family(const uint8_t *family, int64_t n_cubes, int64_t pitch, int cube_size, void *onehot,
                          int fmt, int64_t block_stride, void *stream);
/* The same for n_depths CONSECUTIVE depths of one rc_adi_generate_family call in one launch: `family` points at the first depth's
 * record ([n_depths][tile][NF][pitch]); depth g, block a (child a; a = A: the parent) goes to onehot + (g * (A + 1) + a) * block_stride
 * cubes -- the [depth][A + 1][block_stride] input of ONE forward of the value net (cube_env.py:239-251 evaluates it 2 x per sample). */
int rc_onehot_from_family_depths(const uint8_t *family, int64_t n_cubes, int64_t pitch, int cube_size, void *onehot,
                                 int fmt, int64_t block_stride, int n_depths, void *stream);

int rc_adi_targets(const float *child_value, const uint8_t *child_solved, const float *parent_value,
                   const double *weight, int64_t n, int64_t pitch, int cube_size,
                   float *target_value, int32_t *target_policy, double *error, void *stream);
/* The same rule for n_depths depths at once, reading the value net's output in place and writing WALK-MAJOR results (the layout of
 * the replay sink, utils.py:253-260 one record per (cube, depth)):
 *   child_value[g * cv_depth_stride + a * cv_child_stride + w], parent_value[g * pv_depth_stride + w]  (floats; strides in elements),
 *   child_solved[(g * A + a) * solved_pitch + w]  (the [depth][A][Wp] flags of rc_adi_generate*),  weight[g] = (g0 + g + 1) ** -T,
 *   target_value / target_policy / error [w * out_stride + g]   for w < n, g < n_depths  (out_stride >= n_depths). */
int rc_adi_targets_depths(const float *child_value, int64_t cv_depth_stride, int64_t cv_child_stride,
                          const uint8_t *child_solved, int64_t solved_pitch, const float *parent_value,
                          int64_t pv_depth_stride, const double *weight, int64_t n, int n_depths, int cube_size,
                          float *target_value, int32_t *target_policy, double *error, int64_t out_stride, void *stream);

/* Read-and-clear the device status word: ONE atomic exchange on the device, ordered after the work queued
 * on `stream` (which it synchronises).  The word is per device, not per stream: a bit set by a kernel
 * that is still running on another stream is not lost -- it shows up in a later read. */
int rc_read_status(uint32_t *status, void *stream);

/* Which kernel instantiation and launch geometry a call WOULD use, as text (e.g. "k_step<Cube3,V=2,move,store,POL=1>
 * grid=8192 block=64"), decided by the same host functions the launchers call: benchmarks label their records with it.
 *   op       RC_OP_STEP (rc_apply_moves / rc_is_solved / rc_encode), RC_OP_EXPAND, RC_OP_ADI, RC_OP_CODE_TO_DENSE, RC_OP_FAMILY_TO_DENSE
 *   n        cubes / parents / walks;  depth: RC_OP_ADI only
 *   outputs  RC_OUT_* bits: STATES = the out buffer of a step (move + store) or the children stickers, CODE = compact codes,
 *            FLAGS = child_solved, REWARD / DONE = the step's reward / done arrays, INPLACE = out aliases in (step)
 *   fmt      the one-hot format of a step / code-to-dense call;  variant: the tuning override (0 = defaults)
 * Nothing is launched; no reference counterpart (tooling).  `variant` must only use the RC_VARIANT_* fields of `op`'s group. */
#define RC_OP_STEP 1
#define RC_OP_EXPAND 2
#define RC_OP_ADI 3
#define RC_OP_CODE_TO_DENSE 4
#define RC_OP_FAMILY_TO_DENSE 5 /* rc_onehot_from_family_depths: n = walks, depth = depths per launch, fmt = the dense format; variant must be 0 */
#define RC_OUT_STATES 1u
#define RC_OUT_CODE 2u
#define RC_OUT_FLAGS 4u
#define RC_OUT_REWARD 8u
#define RC_OUT_INPLACE 16u
#define RC_OUT_DONE 32u
#define RC_OUT_FAMILY 128u     /* RC_OP_ADI: the family record instead of codes (rc_adi_generate_family) */
#define RC_OUT_WORKSPACE 64u   /* RC_OP_STEP with a dense fmt: what rc_apply_moves_ws launches when given its workspace */
int rc_describe_dispatch(int op, int cube_size, int64_t n, int depth, unsigned outputs, int fmt, int variant,
                         char *buf, int buflen);

/* Message of the calling thread's last failed call ("" if none). */
const char *rc_last_error(void);

/* ---------------------------------------------------------------------------------------------------------------------------------
 * Tuning override (`variant`) of the *_ex entry points and of rc_describe_dispatch.  TOOLING: benchmarks and tests that must reach
 * every kernel instantiation; there is NO process-global knob and a binding of the reference needs none of this (INTEGRATION.md lists
 * the twelve entry points it does need).  Everything named *_ex, rc_describe_dispatch, rc_family_layout and rc_workspace_bytes is
 * tooling in that sense.  0 = the measured defaults.  A value is the SUM of macros of ONE group below; an entry point rejects
 * (RC_EINVAL) any decimal field its group does not define, so a value built for one entry point never means something else in another.
 *
 * rc_apply_moves_ex / RC_OP_STEP */
#define RC_VARIANT_STEP_PACK(v) (v)                   /* 1, 2: 4, 8 cubes per lane */
#define RC_VARIANT_STEP_POLICY(p) ((p) * 10)          /* row traffic: 1 stream in / stream out, 2 default-cached, 3 stream in / keep the
                                                         output in the Infinity Cache, 4 state default-cached / side outputs streamed */
#define RC_VARIANT_STEP_DENSE_TILE(t) ((t) * 100000)  /* fused dense writer: 1 -> 64-cube tiles, 2 -> 256-cube tiles; any non-zero value
                                                         also makes rc_apply_moves_ws ignore its workspace */
/* rc_expand_children_ex / RC_OP_EXPAND */
#define RC_VARIANT_EXPAND_PACK(v) (v)                 /* 1, 2 */
#define RC_VARIANT_EXPAND_STREAM(h) ((h) * 100)       /* streaming form (few persistent waves): 1..7 -> 128, 192, 256, 384, 512, 768, 1024
                                                         waves, 8 -> never */
#define RC_VARIANT_EXPAND_PARTS(p) ((p) * 1000)       /* 1..A parts per walk group (rounded up to a divisor of A) */
/* rc_adi_generate_ex, rc_adi_generate_family / RC_OP_ADI */
#define RC_VARIANT_ADI_PACK(v) (v)                    /* 1, 2 */
#define RC_VARIANT_ADI_PARTS(p) ((p) * 1000)          /* 1..A */
#define RC_VARIANT_ADI_SEGS(s) ((s) * 1000000)        /* 1..16 depth segments per walk group (clamped to the depth) */
/* rc_onehot_from_code_ex / RC_OP_CODE_TO_DENSE */
#define RC_VARIANT_DENSE_FORM(f) ((f) * 100000)       /* 1 -> 64-cube tiles, 2 -> 256-cube tiles, 3 -> the wide form (960-thread workgroups
                                                         sweeping contiguous tile ranges; superseded, kept for A/B), 4 -> the front form (one
                                                         3840-byte pass per workgroup; the default from 2^15 / 2^16 / 2^18 cubes, f32 / 16-bit / u8) */
#define RC_VARIANT_DENSE_WIDE_GROUPS16(g) ((g) * 1000) /* form 3: wanted workgroups / 16 (1..99) */
#define RC_VARIANT_DENSE_WIDE_SKEW(k) ((k) * 10)      /* form 3: sweep skew 1..9 */
#define RC_VARIANT_DENSE_FRONTS(f) (f)                /* form 4 / default front: 1, 2, 4 fronts per XCD per workgroup */
#define RC_VARIANT_DENSE_FRONT_FETCH(t) ((t) * 10)    /* form 4 / default front: 2 = one linear front, 3 = code bytes by a gather per lane,
                                                         4 = by one load of wave 0 + LDS */
/* rc_legacy_scramble_actions_ex (not decimal: mode + 16 * limit) */
#define RC_VARIANT_LEGACY_LDS 1                       /* the LDS form alone */
#define RC_VARIANT_LEGACY_STREAM(limit) (2 + 16 * (limit)) /* the streaming form with `limit` outputs per lane (1..623; 0 = 623) + fix-up */

#ifdef __cplusplus
}
#endif
#endif /* RUBIKHIP_H */
