// rubikhip.hip -- kernels and C ABI of librubikhip.so (see include/rubikhip.h).
// gfx950 (MI355X, CDNA4) only; no other back end, no compatibility layer.
//
// Kernel map (DESIGN.md has the roofline of each):
//   k_fill_solved     initState_3                      py333.py:211-218
//   k_step            CubeEnv.step for N cubes          cube_env.py:71-111
//                     (move + solved/reward + compact code, all in registers; row traffic = raw buffer
//                     instructions whose cache policy follows the working set, RowPolicy)
//   k_step_dense      same + dense one-hot, the [slot][cube] code tile staged in LDS so the
//                     [N][R][C] rows leave as coalesced 16-byte stores
//   k_code_to_dense   compact code -> dense one-hot (same LDS stage): small batches, 2x2x2
//   k_code_to_dense_front   the same for large 3x3x3 batches: ONE 3840-byte pass per workgroup, one write front per XCD (the
//                     shape of a fill kernel: 0.85-0.94 of the HBM peak wherever the buffer lives); also expands the ADI
//                     family record to its 13 dense blocks; second launch of rc_apply_moves_ws / rc_encode_ws
//   k_scramble        reset()'s scramble loop, in place or out of untouched start states (the lockstep search's replay)    cube_env.py:65-67
//   k_legacy_actions_stream / k_legacy_actions   numpy's legacy MT19937 draws of reset(seed, k), one env per lane: the first generation (outputs 0..622)
//                     from a few init_genrand chain iterators in registers, the general form with the state in LDS    cube_env.py:62-65
//   k_expand          12 children of every cube         cube_env.py:212-236, mcts.py:96-101
//   k_adi             ADI walks + expansion, persistent over depth, per-walk xoroshiro128+; outputs: stickers, picked codes,
//                     or the 51-row FAMILY record (the shared look-ups themselves)    cube_env.py:177-194,212-236
//   k_adi_targets     target value/policy/error, one depth or a group of depths straight from the net's output    cube_env.py:229-232,239-251
//   k_search_pack     codes + flags of one expansion laid out per root for the host trees of a lockstep search    mcts.py:96-101
//   k_facade_step     CubeEnv.step / a whole move list for ONE cube, results into host-mapped memory
//   k_facade_expand   key + 12 child keys + solved flags (+ dense one-hots) of ONE cube    mcts.py:83-113
//   k_read_status     atomic read-and-clear of the per-device status word
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <vector>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../../include/rubikhip.h"
#include "rc_device.h"

using namespace rc;

// Cache policy of the write-once expansion / ADI output streams (rc_device.h "buffer row access"): system-scope
// non-temporal stores; A/B against default-cached and plain nt in profiles/r02_design_ab.json.
#ifndef RC_OUT_AUX
#define RC_OUT_AUX kAuxStreamStore
#endif

namespace {

constexpr int kWave = 64;
__device__ uint32_t g_status;  // RC_STATUS_* bits, per device (one code object instance per device)

// Tiled structure-of-arrays addressing (include/rubikhip.h "State layout"): cube n of a buffer
// with `rows` rows lives in tile n / pitch; tiles follow each other, [tile][row][pitch].
// `shift` = log2(pitch) when the buffer has several tiles, 63 when it has one (then tile = 0 and
// any pitch % 16 == 0 works).  g0 is a wave-uniform cube index that is a multiple of the wave's
// span (<= 512 cubes: 8 per lane at most), so a wave never straddles tiles (multi-tile pitch is a multiple of 512 = kMinTile).
__device__ __forceinline__ int64_t tile_off(int64_t g0, int64_t pitch, int shift, int rows) {
    return g0 + (g0 >> shift) * (rows - 1) * pitch;
}

// ------------------------------------------------------------------------------ fill
template <class T>
__global__ void __launch_bounds__(kWave) k_fill_solved(uint8_t *base, int64_t n, int64_t pitch, int shift) {
    const int64_t g0 = (int64_t)blockIdx.x * (kWave * 8);      // wave span 512 cubes = the smallest tile
    const uint32_t lo = threadIdx.x * 8;
    if (g0 + lo >= n) return;
    const __amdgpu_buffer_rsrc_t r = make_srd(base + tile_off(g0, pitch, shift, T::S));
    const uint32_t rs = (uint32_t)pitch;
#pragma unroll
    for (int s = 0; s < T::S; ++s) bst<2, kAuxCached>(r, lo, s * rs, splat<2>((uint32_t)(s / T::FACE) * 0x01010101u));
}

// ------------------------------------------------------------------------------ step
struct StepArgs {
    const uint8_t *in;
    uint8_t *out;
    const uint8_t *actions;
    int64_t n, pitch_in, pitch_out;
    float *reward;
    uint8_t *done;
    uint8_t *code;
    int64_t code_pitch;
    int sh_in, sh_out, sh_code;   // tile shifts (see tile_off)
};

template <int V>
__device__ __forceinline__ void store_reward(float *reward, int64_t n0, int64_t n, Pk<V> dn) {
    if (n0 + 4 * V <= n) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float4 f = make_float4(reward_of(dn.d[k], 0), reward_of(dn.d[k], 1), reward_of(dn.d[k], 2), reward_of(dn.d[k], 3));
            *reinterpret_cast<float4 *>(reward + n0 + 4 * k) = f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4 * V; ++j)
            if (n0 + j < n) reward[n0 + j] = reward_of(dn.d[j >> 2], j & 3);
    }
}

// Row-traffic policy of the step kernel (aux bits in rc_device.h):
//   POL 0  everything default-cached: the launch's working set fits the 256 MiB Infinity Cache;
//   POL 1  inputs streamed (nt), outputs written through but kept (sc0 sc1): the next launch reads what this one
//          wrote and the output alone still fits the Infinity Cache (state ping-pong of 2^22 cubes);
//   POL 2  inputs streamed, outputs streamed (sc0 sc1 nt): beyond that;
//   POL 3  state rows default-cached, side outputs streamed: the STATE fits the Infinity Cache but state + code + reward
//          would not (in-place steps of 2^22 cubes with the fused code, ping-pong of 2^21).
// SIDE = the outputs nobody re-reads in the next launch (compact code, done, reward): under POL 1 they are streamed past
// the Infinity Cache so that they do not push the kept state rows out of it (226 MB of state + 84 MB of code would not fit).
#ifndef RC_SIDE_AUX_POL1
#define RC_SIDE_AUX_POL1 kAuxStreamStore
#endif
template <int POL> struct RowPolicy;
template <> struct RowPolicy<0> { static constexpr int LD = kAuxCached, ST = kAuxCached, SIDE = kAuxCached; };
template <> struct RowPolicy<1> { static constexpr int LD = kAuxStreamLoad, ST = kAuxKeepStore, SIDE = RC_SIDE_AUX_POL1; };
template <> struct RowPolicy<2> { static constexpr int LD = kAuxStreamLoad, ST = kAuxStreamStore, SIDE = kAuxStreamStore; };
template <> struct RowPolicy<3> { static constexpr int LD = kAuxCached, ST = kAuxCached, SIDE = kAuxStreamStore; };
// POL 4 (rc_apply_moves_ws beyond the Infinity Cache): state rows streamed both ways, the side outputs -- the compact code the front
// writer reads back in the very next launch -- written through and KEPT.  Streamed past the cache instead, the code comes back from
// HBM with a miss latency per front workgroup: the two-launch route drops from 0.89 to 0.52 of peak at 2^21 cubes.
template <> struct RowPolicy<4> { static constexpr int LD = kAuxStreamLoad, ST = kAuxStreamStore, SIDE = kAuxKeepStore; };

// One lane = 4*V consecutive cubes.  MOVE: apply actions; STORE: write the state rows;
// CODE: write the compact code rows.  FULL: every pack of the wave lies
// inside the batch (all but the last wave): no per-byte tail paths in the instruction stream.
template <class T, int V, bool MOVE, bool STORE, bool CODE, int POL, bool FULL>
__device__ __forceinline__ void step_body(const StepArgs &a, int64_t g0, uint32_t lo) {
    using P = RowPolicy<POL>;
    const int64_t n0 = g0 + lo;
    const int64_t n = FULL ? n0 + 4 * V : a.n;               // FULL: the tail helpers take their vector path
    Pk<V> s[T::S];
    {
        const __amdgpu_buffer_rsrc_t r = make_srd(a.in + tile_off(g0, a.pitch_in, a.sh_in, T::S));
        const uint32_t rs = (uint32_t)a.pitch_in;
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = bld<V, P::LD>(r, lo, i * rs);
    }
    if constexpr (MOVE) {
        const Pk<V> act = ld_tail<V>(a.actions, n0, n, 0);
        Pk<V> m[T::A];
        const Pk<V> bad = action_masks<T, V>(act, m);
        if (any_bad<V>(bad, n0, n)) atomicOr(&g_status, RC_STATUS_BAD_ACTION);
        Pk<V> o[T::S];
        apply_move<T, V>(s, m, o);
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = o[i];
    }
    if constexpr (STORE) {
        const __amdgpu_buffer_rsrc_t r = make_srd(a.out + tile_off(g0, a.pitch_out, a.sh_out, T::S));
        const uint32_t rs = (uint32_t)a.pitch_out;
#pragma unroll
        for (int i = 0; i < T::S; ++i) bst<V, P::ST>(r, lo, i * rs, s[i]);
    }
    if (a.done != nullptr || a.reward != nullptr) {
        const Pk<V> dn = done_bytes(unsolved<T, V>(s));
        if constexpr (FULL) {                                  // whole packs: buffer stores with the side-output policy
            if (a.done) bst<V, P::SIDE>(make_srd(a.done + g0), lo, 0, dn);
            if (a.reward) {
                // every store instruction writes 1 KiB of CONTIGUOUS floats (streamed partial lines are slow: a lane's own 8 cubes
                // would put its two 16-byte pieces 32 bytes apart).  V = 2: instruction h covers cubes 256h + 4l .. + 3 of lane l,
                // whose done bytes sit in dword (l & 1) of lane 32h + l/2 -- one wave shuffle per dword.
                const __amdgpu_buffer_rsrc_t r = make_srd(a.reward + g0);
                const int lane = (int)(threadIdx.x & (kWave - 1));
#pragma unroll
                for (int h = 0; h < V; ++h) {
                    uint32_t dd = dn.d[0];
                    if constexpr (V == 2) {
                        const int src = 32 * h + (lane >> 1);
                        const uint32_t d0 = (uint32_t)__shfl((int)dn.d[0], src), d1 = (uint32_t)__shfl((int)dn.d[1], src);
                        dd = (lane & 1) ? d1 : d0;
                    }
                    Pk<4> f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) f.d[j] = __float_as_uint(reward_of(dd, j));
                    bst<4, P::SIDE>(r, (uint32_t)lane * 16u, (uint32_t)h * 1024u, f);
                }
            }
        } else {
            if (a.done) st_tail<V>(a.done, n0, n, dn);
            if (a.reward) store_reward<V>(a.reward, n0, n, dn);
        }
    }
    if constexpr (CODE) {
        Pk<V> c[T::SLOTS];
        encode<T, V>(s, c);
        const __amdgpu_buffer_rsrc_t r = make_srd(a.code + tile_off(g0, a.code_pitch, a.sh_code, T::SLOTS));
        const uint32_t rs = (uint32_t)a.code_pitch;
#pragma unroll
        for (int p = 0; p < T::SLOTS; ++p) bst<V, P::SIDE>(r, lo, p * rs, c[p]);
    }
}

template <class T, int V, bool MOVE, bool STORE, bool CODE, int POL, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_step(StepArgs a) {
    // wave-uniform part of the cube index in SGPRs, 32-bit lane offset in one VGPR
    const int64_t g0 = (int64_t)blockIdx.x * (BLOCK * 4 * V);
    const uint32_t lo = threadIdx.x * (4 * V);
    if (g0 + BLOCK * 4 * V <= a.n) {                           // uniform: the whole workgroup is inside the batch
        step_body<T, V, MOVE, STORE, CODE, POL, true>(a, g0, lo);
    } else if (g0 + lo < a.n) {
        step_body<T, V, MOVE, STORE, CODE, POL, false>(a, g0, lo);
    }
}

// ----------------------------------------------------------- step + dense one-hot (LDS stage)
constexpr int kDenseBlock = 256;
// 2x2x2: the dense writer's unit is a 147-chunk pass (dense_write_222).  1- and 2-byte elements: 320 threads, 294 of them cover two passes
// per round (instruction-bound in the generic loop: 0.45 / 0.23 against 0.53 / 0.52; profiles/r05_dense222.json).  f32: 640 threads, 588
// of them cover four passes per round (rc_device.h RC_D222_F32: 0.67 against 0.65 for the generic 256-thread loop at 1M cubes, 0.58 against
// 0.49 at 64k; profiles/r06_d222_f32.json).
template <class T, class E> constexpr int kDenseThreads = (T::SIZE == 2 && sizeof(E) <= 2) ? 320 : (T::SIZE == 2 && RC_D222_F32 != 0) ? RC_D222_F32 : kDenseBlock;
// TILE cubes per workgroup (64 | 256): the first TILE/4 lanes compute the pack's codes, then
// all 256 threads stream the dense rows.  Small tiles keep small batches (MCTS leaves) spread over
// the chip: 4096 cubes are 64 workgroups at TILE = 64 but only 4 at TILE = 1024.
// LDS row pitch TILE + 4 bytes: rows fall on different banks for the byte reads of dense_write.

// One wave's share of a dense step: the 4 cubes of this lane (tile0 + lo ..) are loaded, moved, stored, flagged and encoded,
// their codes land in the LDS code tile `lds_code` ([SLOTS][tp] bytes).  tile0 is wave-uniform and a multiple of 256.
template <class T, bool MOVE, bool STORE>
__device__ __forceinline__ void dense_produce(const StepArgs &a, int64_t tile0, uint32_t lo, uint8_t *lds_code, int tp) {
    const int64_t n0 = tile0 + lo;
    Pk<1> s[T::S];
    {
        const __amdgpu_buffer_rsrc_t r = make_srd(a.in + tile_off(tile0, a.pitch_in, a.sh_in, T::S));
        const uint32_t rs = (uint32_t)a.pitch_in;
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = bld<1, kAuxCached>(r, lo, i * rs);
    }
    if constexpr (MOVE) {
        const Pk<1> act = ld_tail<1>(a.actions, n0, a.n, 0);
        Pk<1> m[T::A];
        const Pk<1> bad = action_masks<T, 1>(act, m);
        if (any_bad<1>(bad, n0, a.n)) atomicOr(&g_status, RC_STATUS_BAD_ACTION);
        apply_move_inplace<T, 1>(s, m);
    }
    if constexpr (STORE) {
        const __amdgpu_buffer_rsrc_t r = make_srd(a.out + tile_off(tile0, a.pitch_out, a.sh_out, T::S));
        const uint32_t rs = (uint32_t)a.pitch_out;
#pragma unroll
        for (int i = 0; i < T::S; ++i) bst<1, kAuxCached>(r, lo, i * rs, s[i]);
    }
    if (a.done != nullptr || a.reward != nullptr) {
        const Pk<1> dn = done_bytes(unsolved<T, 1>(s));
        if (a.done) st_tail<1>(a.done, n0, a.n, dn);
        if (a.reward) store_reward<1>(a.reward, n0, a.n, dn);
    }
    Pk<1> c[T::SLOTS];
    encode<T, 1>(s, c);
#pragma unroll
    for (int p = 0; p < T::SLOTS; ++p) *reinterpret_cast<uint32_t *>(lds_code + p * tp + lo) = c[p].d[0];
}

template <class T, class E, bool MOVE, bool STORE, int TILE>
__global__ void __launch_bounds__((kDenseThreads<T, E>)) k_step_dense(StepArgs a, E *dense) {
    constexpr int TP = TILE + 4;
    __shared__ __attribute__((aligned(16))) uint8_t lds_code[T::SLOTS * TP];
    const uint32_t lo = threadIdx.x * 4;
    // grid-stride over tiles: large batches run a capped grid (launch_dense_t), every workgroup a few tiles in turn
    for (int64_t tile0 = (int64_t)blockIdx.x * TILE; tile0 < a.n; tile0 += (int64_t)gridDim.x * TILE) {
#if RC_DENSE_CTRL < 2                                                  // (control experiment builds only: rc_device.h RC_DENSE_CTRL)
        if (lo < TILE && tile0 + lo < a.n) dense_produce<T, MOVE, STORE>(a, tile0, lo, lds_code, TP);
        __syncthreads();
#endif
        const int64_t left = a.n - tile0;
        const int ncubes = left < TILE ? (int)left : TILE;
        dense_write<T, E>(lds_code, TP, dense + tile0 * (T::R * T::C), ncubes, threadIdx.x, kDenseThreads<T, E>);
#if RC_DENSE_CTRL < 2
        __syncthreads();                                               // the code tile is reused by the next iteration
#endif
    }
}

template <class T, class E, int TILE>
__global__ void __launch_bounds__((kDenseThreads<T, E>)) k_code_to_dense(const uint8_t *code, int64_t n, int64_t code_pitch, int shift, E *dense,
                                                                         int64_t src_block_stride = 0, int64_t dst_block_stride = 0) {
    constexpr int TP = TILE + 4;
    __shared__ __attribute__((aligned(16))) uint8_t lds_code[T::SLOTS * TP];
    const uint32_t lo = threadIdx.x * 4;
    // rc_onehot_from_code_blocks: blockIdx.y = one of several equally tiled code buffers (src_block_stride bytes apart), written to its own
    // block of the output (dst_block_stride cubes apart); a plain launch has gridDim.y == 1 and both strides 0
    code += (int64_t)blockIdx.y * src_block_stride;
    dense += (int64_t)blockIdx.y * dst_block_stride * (T::R * T::C);
    for (int64_t tile0 = (int64_t)blockIdx.x * TILE; tile0 < n; tile0 += (int64_t)gridDim.x * TILE) {
#if RC_DENSE_CTRL < 2
        if (lo < TILE && tile0 + lo < n) {
            const __amdgpu_buffer_rsrc_t r = make_srd(code + tile_off(tile0, code_pitch, shift, T::SLOTS));
            const uint32_t rs = (uint32_t)code_pitch;
#pragma unroll
            for (int p = 0; p < T::SLOTS; ++p) *reinterpret_cast<uint32_t *>(lds_code + p * TP + lo) = bld<1, kAuxCached>(r, lo, p * rs).d[0];
        }
        __syncthreads();
#endif
        const int64_t left = n - tile0;
        const int ncubes = left < TILE ? (int)left : TILE;
        dense_write<T, E>(lds_code, TP, dense + tile0 * (T::R * T::C), ncubes, threadIdx.x, kDenseThreads<T, E>);
#if RC_DENSE_CTRL < 2
        __syncthreads();
#endif
    }
}

// WIDE form of the code -> dense writer (3x3x3, batches from 2^17 cubes): 960-thread workgroups (15 waves; 960 = 8 * 120
// covers whole cubes for every element size), about 112 of them, each sweeping ONE contiguous range of 256-cube tiles, so the
// chip writes ~112 long sequential streams of 15-KiB bursts instead of 2048-4096 interleaved 4-KiB ones.  Measured at 2^20
// cubes in the same process on the same buffers (profiles/r03_ab.json): bf16 0.68-0.72 -> 0.77-0.91 of the 8 TB/s peak, f32
// 0.66-0.83 -> 0.71-0.92, u8 0.67-0.74 -> 0.77-0.85 (the spread is between GPU boxes); at least level with the 256-thread form
// for every batch size tried (2^17 .. 2^22, powers of two and not).  Group counts from 64 to 512 were swept: ~112 is the best
// or within 2 % of it for all three element sizes; 64 groups are too few waves.  Group g starts its sweep 3g tiles into its range
// and wraps (kWideSkew): neutral at 112 groups, +3-5 % where the ranges are equal powers of two (128 groups at 2^20 / 2^21 cubes).
constexpr int kWideBlock = 960, kWideGroups = 112, kWideTile = 256, kWideSkew = 3;

template <class T, class E>
__global__ void __launch_bounds__(kWideBlock) k_code_to_dense_wide(const uint8_t *code, int64_t n, int64_t code_pitch, int shift, E *dense,
                                                                   int64_t tiles_per_group, int skew) {
    constexpr int TILE = kWideTile, TP = TILE + 4;
    __shared__ __attribute__((aligned(16))) uint8_t lds_code[2][T::SLOTS * TP];
    const uint32_t lo = threadIdx.x * 4;
    const int64_t first = (int64_t)blockIdx.x * tiles_per_group * TILE;
    int64_t last = first + tiles_per_group * TILE;
    if (last > n) last = n;
    // skew: group g starts its sweep (g * skew) tiles into its range and wraps, so that the concurrent streams do not advance
    // through equally spaced addresses in lockstep
    const int64_t mine = (last - first + TILE - 1) / TILE;
    const int64_t start = mine > 0 ? ((int64_t)blockIdx.x * skew) % mine : 0;
    int buf = 0;
    for (int64_t k = 0; k < mine; ++k, buf ^= 1) {
        const int64_t tile0 = first + ((start + k) % mine) * TILE;
#if RC_DENSE_CTRL < 2
        if (lo < TILE && tile0 + lo < n) {
            const __amdgpu_buffer_rsrc_t r = make_srd(code + tile_off(tile0, code_pitch, shift, T::SLOTS));
            const uint32_t rs = (uint32_t)code_pitch;
#pragma unroll
            for (int p = 0; p < T::SLOTS; ++p) *reinterpret_cast<uint32_t *>(lds_code[buf] + p * TP + lo) = bld<1, kAuxCached>(r, lo, p * rs).d[0];
        }
        __syncthreads();                                            // double-buffered tile: one barrier per tile is enough
#endif
        const int64_t left = n - tile0;
        const int ncubes = left < TILE ? (int)left : TILE;
        dense_write_333<T, E, kWideBlock>(lds_code[buf], TP, dense + tile0 * (T::R * T::C), ncubes, threadIdx.x);
    }
}

// FRONT form of the code -> dense writer (3x3x3, large batches; round 4, profiles/r04_dense_control.json): 3840-byte passes, each
// written by ONE store instruction per lane of a workgroup that then ends -- 240 threads write the 2 / 4 / 8 whole cubes (f32 /
// 16-bit / u8) of a pass.  Workgroups start in address order and live for one store round trip, so at any moment the chip writes a
// dense window of a few MB that sweeps the buffer front to back, the way a fill kernel does: a store-only kernel of this shape runs
// at 0.88-0.93 of the 8 TB/s peak on EVERY allocation, while any shape whose workgroups write a longer private stream (8 KiB
// chunks and up, persistent or not) lands at 0.65-0.85 depending on where the buffer lives (tools/dense_shape.hip).
// Only 2048 workgroups are resident and each holds 3840 bytes per front, so throughput = resident bytes / workgroup lifetime:
//   * fronts: blocks b, b + 8, ... (one XCD: blocks are dealt round-robin over the 8 XCDs) take CONSECUTIVE passes of one eighth
//     of the buffer, so the code lines a pass needs were fetched into this XCD's L2 by the passes just before it (8 fronts);
//   * F: every workgroup serves F such fronts per XCD (one pass each, all code bytes loaded before the first store), which
//     multiplies the bytes in flight per workgroup lifetime without lengthening any front's private stream.
// (Wave-uniform scalar loads of the rows + a select chain were tried and lost: 0.64 against 0.81, bf16.)
// FAM: `code` holds FAMILY rows ([tile][NF][pitch], rc_device.h FamilyLayout) and the launch writes A + 1 blocks -- child a's one-hots
// at dense + a * block_stride cubes, the parent's as block A: global pass = block * passes_per_block + pass; slot r of block a is family
// row kFamily.row[a][r] (the per-child pick).  Several DEPTHS of one ADI launch go in one launch too: blockIdx.y = depth * (A + 1) + block;
// depth g reads its family record at code + g * fam.src_depth_stride bytes and writes its A + 1 blocks at dense + g * fam.dst_depth_stride
// cubes, so the whole [depth][A + 1][block_stride] model input of a small batch is one launch.
// The pick table as dwords: block a's 20 row indices in 5 words (rows padded to 8 so that ONE s_load_dwordx8 fetches a row).  A FAM
// launch is two-dimensional -- blockIdx.y = the block (depth * (A + 1) + child), blockIdx.x = the pass inside it, swept by 8 fronts like
// the plain form -- so the block, and with it the table row, is known from the workgroup id alone: the row arrives by a scalar load
// issued together with the kernel arguments, and a lane picks its slot's byte with two 64-bit selects and a shift.  Round 4 read the
// table with a per-lane byte load, a second dependent memory round trip in front of every code gather, and a front workgroup LIVES
// for its round trips: 0.52 of the HBM peak at 43008 walks against 0.95 for the plain code -> dense form (profiles/r05_family_front.json).
struct alignas(32) FamilyRowWords {      // read as u32x8 (32-byte vectors): the symbol must be placed on that alignment
    uint32_t w[Cube3::A + 1][8];
    constexpr FamilyRowWords() : w{} {
        for (int a = 0; a <= Cube3::A; ++a)
            for (int p = 0; p < Cube3::SLOTS; ++p) w[a][p >> 2] |= (uint32_t)kFamily<Cube3>.row[a][p] << (8 * (p & 3));
    }
};
__constant__ FamilyRowWords c_family3_rows{};
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
struct FamilyBlocks { int64_t block_stride, src_depth_stride, dst_depth_stride; };
template <class T, class E, int F, bool LDS, bool FAM = false>
__global__ void __launch_bounds__(256) k_code_to_dense_front(const uint8_t *__restrict__ code, int64_t n, int64_t code_pitch, int shift, E *__restrict__ dense,
                                                             int64_t per_xcd, int64_t per_front, FamilyBlocks fam = {}) {
    static_assert(T::SIZE == 3);
    constexpr int ROWS = FAM ? kFamily<T>.nf : T::SLOTS;                         // rows of one tile of the input
    constexpr int EPT = 16 / (int)sizeof(E), CPC = 480 / EPT, CPP = 240 / CPC;   // elements per chunk, chunks per cube, cubes per pass
    constexpr int WORDS = CPP > 4 ? 2 : 1;                                       // dwords of one code row that hold a pass's cubes
    // LDS: the pass's code bytes are SLOTS x WORDS aligned dwords fetched by the first lanes of wave 0 (one load instruction, 20
    // lines) and handed over through LDS, instead of one byte gather per lane in every wave (4 x ~21 lines for the 1- and 2-byte
    // formats).  f32 passes hold 2 cubes (a wave's gather touches ~11 lines) and run faster WITHOUT the barrier: measured side by
    // side (profiles/r04_dense_control.json "front_code_fetch"): f32 0.95 gather / 0.87 LDS, bf16 0.81 / 0.87, u8 0.54 / 0.76.
    __shared__ uint32_t rows[LDS ? F : 1][T::SLOTS * WORDS];
    const int tid = threadIdx.x;
    const int sub = tid / CPC, k = tid - sub * CPC;                             // (tid >= 240: sub == CPP, no chunk)
    // per_xcd == 0: one linear front (pass = block); else the first pass of this block inside its XCD's range
    const int64_t pass0 = per_xcd > 0 ? (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : (int64_t)blockIdx.x;
    // rows / columns of this thread's chunk (fixed): sizeof(E) >= 2: one row; u8: the 16 elements lie in at most two rows
    const int ra = (k * EPT) / T::C, rb = sizeof(E) >= 2 ? ra : (k * 16 + 12) / T::C;
    uint32_t ca[F], cb[F];
    bool live[F];
    // FAM: which block (child / parent) a global pass belongs to, its pass inside the block, and the input row of slot r there
    int64_t lpass[F];                                                            // pass (inside the block, FAM)
#pragma unroll
    for (int f = 0; f < F; ++f) lpass[f] = pass0 + f * per_front;
    int64_t src_off = 0, dst_cube = 0;                                           // FAM: byte offset of the depth's record; first cube of the block
    uint64_t rlo = 0, rmid = 0, rhi = 0;                                         // FAM: the block's 20 row indices (wave-uniform)
    if constexpr (FAM) {
        const uint32_t b = blockIdx.y, g = b / (uint32_t)(T::A + 1), a = b - g * (uint32_t)(T::A + 1);
        src_off = (int64_t)g * fam.src_depth_stride;
        dst_cube = (int64_t)g * fam.dst_depth_stride + (int64_t)a * fam.block_stride;
        const u32x8 rw = *reinterpret_cast<const u32x8 *>(c_family3_rows.w[a]);
        rlo = rw[0] | (uint64_t)rw[1] << 32;
        rmid = rw[2] | (uint64_t)rw[3] << 32;
        rhi = rw[4];
    }
    auto in_row = [&](int, int r) -> int {
        if constexpr (!FAM) return r;
        const uint64_t w = r < 8 ? rlo : r < 16 ? rmid : rhi;
        return (int)((w >> (8 * (r & 7))) & 0xffu);
    };
    if constexpr (LDS) {
#pragma unroll
        for (int f = 0; f < F; ++f) {
            const int64_t cube0 = lpass[f] * CPP, a0 = cube0 & ~(int64_t)3;
            if (tid < T::SLOTS * WORDS && cube0 < n && (f == 0 || per_front > 0)) {
                const int r = tid / WORDS, w = tid - r * WORDS;
                rows[f][tid] = *reinterpret_cast<const uint32_t *>(code + src_off + tile_off(a0 + 4 * w, code_pitch, shift, ROWS) + (int64_t)in_row(f, r) * code_pitch);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
        const int64_t cube = lpass[f] * CPP + sub;
        live[f] = tid < 240 && cube < n && (f == 0 || per_front > 0);
        ca[f] = cb[f] = 0xff;
        if (live[f]) {
            if constexpr (LDS) {
                const int byte = (int)((lpass[f] * CPP) & 3) + sub;              // the cube's byte inside the row's dword(s)
                ca[f] = (rows[f][ra * WORDS + (byte >> 2)] >> (8 * (byte & 3))) & 0xffu;
                if constexpr (sizeof(E) == 1) cb[f] = (rows[f][rb * WORDS + (byte >> 2)] >> (8 * (byte & 3))) & 0xffu;
            } else {
                const uint8_t *src = code + src_off + tile_off(cube, code_pitch, shift, ROWS);   // row 0 of this cube's column
                ca[f] = src[(int64_t)in_row(f, ra) * code_pitch];
                if constexpr (sizeof(E) == 1) cb[f] = src[(int64_t)in_row(f, rb) * code_pitch];
            }
        }
    }
#pragma unroll
    for (int f = 0; f < F; ++f) {
        if (!live[f]) continue;
        Pk<4> u;
        if constexpr (sizeof(E) >= 2) {
            u = onehot_chunk<E>(ca[f] - (uint32_t)(k * EPT - ra * T::C));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = k * 16 + 4 * j, r = e / T::C;
                const uint32_t d = (r == ra ? ca[f] : cb[f]) - (uint32_t)(e - r * T::C);
                u.d[j] = d < 4u ? 1u << (8u * d) : 0u;
            }
        }
        // the pass is 3840 contiguous bytes: thread t owns bytes 16 t ..
        bst<4, RC_DENSE_AUX>(make_srd(dense + (dst_cube + lpass[f] * CPP) * 480), (uint32_t)tid * 16u, 0, u);
    }
}

// (A FRONT form for the 2x2x2 -- 2352-byte passes, 1 / 2 / 4 passes per workgroup-front, 1 / 2 fronts, code dwords through LDS -- was built
// and measured in round 6 and lost to the tile kernels for every format and size: f32 0.34-0.65 against 0.66, bf16 0.27-0.48 against 0.52,
// u8 0.17-0.27 against 0.52-0.56 at 1M-4M cubes; profiles/r06_front222.json, EXPERIMENTS.md.  The kernel is in the git history.)

// ---------------------------------------------------------------------------- expand
struct ExpandArgs {
    const uint8_t *in;
    int64_t n, pitch_in;
    uint8_t *children, *child_solved, *child_code;
    int64_t pitch_out, tiles_out;
    int parts, sh_in, sh_out;
};

// Where one expansion's outputs go.  Child a of the wave's cubes: a tiled state buffer of its own,
// children + a * tiles * S * pitch (layout [A][tile][S][pitch]); codes likewise with SLOTS rows;
// flags are [A][tiles * pitch].  The three pointers already include the wave's own offset
// (tile_off / g0), so only the per-child strides are left.  Every pointer is WAVE-UNIFORM (kernel
// argument + block-derived offset, SGPRs): each child gets a buffer descriptor of its own, the rows are
// scalar offsets, the lane contributes one 32-bit offset -- no address VGPRs at all.
struct ChildOut {
    uint8_t *children, *child_solved, *child_code;
    uint32_t pitch;
    int64_t tiles;
    bool flags_live;   // REACH only: false = no cube of the wave can have a solved child (wave-uniform), flags are all 0
};

// REACH: the parent is reachable from the solved cube (ADI walks): flags from the shared ring test
// (rc_device.h child_unsolved); otherwise the literal per-child face test.
template <class T, int V, int A_, bool CODE, bool REACH>
__device__ __forceinline__ void emit_child(const Pk<V> (&s)[T::S], const FamilyCodes<T, V> &fam, const ChildFlags<T, V> &cf,
                                           const ChildOut &o, uint32_t lo) {
    const int64_t wp = o.tiles * o.pitch;
    if (o.children) {
        const __amdgpu_buffer_rsrc_t r = make_srd(o.children + (int64_t)A_ * T::S * wp);
        sfor<T::S>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            bst<V, RC_OUT_AUX>(r, lo, i * o.pitch, s[kPerm<T>.v[A_][i]]);   // the child is a register renaming of the parent
        });
    }
    if (o.child_solved) {
        Pk<V> flag = splat<V>(0);
        if constexpr (REACH) {
            if (o.flags_live) flag = done_bytes(child_unsolved<T, V, A_>(s, cf));
        } else {
            Pk<V> c[T::S];
            fixed_move<T, V, A_>(s, c);
            flag = done_bytes(unsolved<T, V>(c));
        }
        bst<V, RC_OUT_AUX>(make_srd(o.child_solved + (int64_t)A_ * wp), lo, 0, flag);
    }
    if constexpr (CODE) {
        Pk<V> cc[T::SLOTS];
        family_pick<T, V, A_>(fam, cc);
        const __amdgpu_buffer_rsrc_t r = make_srd(o.child_code + (int64_t)A_ * T::SLOTS * wp);
#pragma unroll
        for (int p = 0; p < T::SLOTS; ++p) bst<V, RC_OUT_AUX>(r, lo, p * o.pitch, cc[p]);
    }
}

// children part, part+parts, ... of one parent pack; `fam` (the family's shared code look-ups) is only
// read when CODE, `cf` only when REACH
template <class T, int V, bool CODE, bool REACH>
__device__ __forceinline__ void emit_children(const Pk<V> (&s)[T::S], const FamilyCodes<T, V> &fam, const ChildFlags<T, V> &cf,
                                              int part, int parts, const ChildOut &o, uint32_t lo) {
    sfor<T::A>([&](auto ac) {
        constexpr int a = decltype(ac)::value;
        if ((a - part) % parts == 0 && a >= part) emit_child<T, V, a, CODE, REACH>(s, fam, cf, o, lo);
    });
}

template <class T, int V, bool CODE>
__global__ void __launch_bounds__(kWave) k_expand(ExpandArgs a) {
    const int64_t item = blockIdx.x;
    const int64_t g = item / a.parts;
    const int part = (int)(item - g * a.parts);
    const int64_t g0 = g * (kWave * 4 * V);
    const uint32_t lo = threadIdx.x * (4 * V);
    if (g0 + lo >= a.n) return;
    Pk<V> s[T::S];
    {
        const __amdgpu_buffer_rsrc_t r = make_srd(a.in + tile_off(g0, a.pitch_in, a.sh_in, T::S));
        const uint32_t rs = (uint32_t)a.pitch_in;
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = bld<V, kAuxCached>(r, lo, i * rs);
    }
    const ChildOut o{a.children ? a.children + tile_off(g0, a.pitch_out, a.sh_out, T::S) : nullptr,
                     a.child_solved ? a.child_solved + g0 : nullptr,
                     a.child_code ? a.child_code + tile_off(g0, a.pitch_out, a.sh_out, T::SLOTS) : nullptr, (uint32_t)a.pitch_out, a.tiles_out, true};
    FamilyCodes<T, V> fam;
    ChildFlags<T, V> cf;
    if constexpr (CODE) family_codes<T, V, false>(s, fam);
    emit_children<T, V, CODE, false>(s, fam, cf, part, a.parts, o, lo);
}

// Streaming form for large sticker expansions (round 3): FEW persistent waves, each taking walk groups g, g + grid, ... in turn
// with the next group's 54 rows prefetched while the current group's 12 x 54 child rows stream out -- the shape the ADI kernel
// has (196 waves, 0.84 of peak) instead of 2048 short-lived waves: 1M parents 116.4 -> 113 us (0.80 -> 0.83).  Stickers + flags only (the code-emitting and the small
// latency-bound expansions keep k_expand).
template <class T>
__global__ void __launch_bounds__(kWave) k_expand_stream(ExpandArgs a) {
    constexpr int V = 2;
    const uint32_t lo = threadIdx.x * (4 * V);
    const int64_t span = kWave * 4 * V, groups = (a.n + span - 1) / span;
    int64_t g = blockIdx.x;
    if (g >= groups) return;
    Pk<V> nxt[T::S];
    auto load = [&](int64_t grp) {
        const int64_t g0 = grp * span;
        const __amdgpu_buffer_rsrc_t r = make_srd(a.in + tile_off(g0, a.pitch_in, a.sh_in, T::S));
        const uint32_t rs = (uint32_t)a.pitch_in;
        const bool live = g0 + lo < a.n;                            // lanes beyond the batch read nothing (their rows may not exist)
#pragma unroll
        for (int i = 0; i < T::S; ++i) nxt[i] = live ? bld<V, kAuxStreamLoad>(r, lo, i * rs) : splat<V>(0);
    };
    load(g);
    while (true) {
        Pk<V> s[T::S];
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = nxt[i];
        const int64_t g0 = g * span, gn = g + gridDim.x;
        if (gn < groups) load(gn);                                   // in flight while this group's children are written
        if (g0 + lo < a.n) {
            const ChildOut o{a.children + tile_off(g0, a.pitch_out, a.sh_out, T::S), a.child_solved ? a.child_solved + g0 : nullptr, nullptr,
                             (uint32_t)a.pitch_out, a.tiles_out, true};
            FamilyCodes<T, V> fam;
            ChildFlags<T, V> cf;
            emit_children<T, V, false, false>(s, fam, cf, 0, 1, o, lo);
        }
        if (gn >= groups) break;
        g = gn;
    }
}

// ------------------------------------------------------------------------------- ADI
constexpr int kMaxSegs = 16;
struct AdiArgs {
    uint64_t seed, stream_id;
    int64_t walk_offset, n_walks, pitch, tiles;   // tiles * pitch = padded walk count of one [.] row
    int depth, parts, shift;
    const uint8_t *actions_in;
    uint8_t *actions_out, *parents, *parent_code, *children, *child_code, *child_solved;
    uint8_t *family;                              // [depth][tile][NF][pitch]: the shared look-ups themselves (rc_device.h FamilyLayout)
    int segs;                                     // depth segments per walk group (1..kMaxSegs)
    uint16_t seg_lo[kMaxSegs + 1];                // segment s emits depths [seg_lo[s], seg_lo[s + 1])
};

// One wave = 256*V walks, kept in registers for all `depth` steps (persistent over depth).
// `parts` waves share a walk group: each recomputes the (cheap) walk and writes its share of
// the children, which is where all the bytes go; part 0 also writes actions and parents.
// Large batches run V = 2 (8 walks per lane, 512 B per store instruction) with ONE wave per walk group
// (pick_geometry; DESIGN.md "ADI write path").
// `segs` waves split the DEPTH range of a walk group: the wave of segment s replays the moves of depths
// < seg_lo[s] without any output (RNG + move: ~1/5 of a full depth with codes), then emits its own depths.  Code-only
// generation is VALU-bound with one wave per 256 walks (fewer waves than SIMDs at 100k walks): depth segments fill the chip
// without duplicating the code look-ups the way `parts` does.
template <class T, int V, bool CODE, bool FAM = false>
__global__ void __launch_bounds__(kWave) k_adi(AdiArgs a) {
    static_assert(CODE || !FAM, "the family record is made of the code look-ups");
    const int64_t item = blockIdx.x;
    const int ps = a.parts * a.segs;
    const int64_t g = item / ps;
    const int sub = (int)(item - g * ps);
    const int seg = sub / a.parts, part = sub - seg * a.parts;
    const int d_lo = a.seg_lo[seg], d_hi = a.seg_lo[seg + 1];
    const int64_t g0 = g * (kWave * 4 * V);
    const uint32_t lo = threadIdx.x * (4 * V);
    const int64_t w0 = g0 + lo;
    if (w0 >= a.n_walks) return;
    Pk<V> s[T::S];
#pragma unroll
    for (int i = 0; i < T::S; ++i) s[i] = splat<V>((uint32_t)(i / T::FACE) * 0x01010101u);
    WalkRng rng[4 * V];
    if (a.actions_in == nullptr) {
#pragma unroll
        for (int j = 0; j < 4 * V; ++j) rng[j].seed(a.seed, a.stream_id, (uint64_t)(a.walk_offset + w0 + j));
    }
    const int64_t wp = a.tiles * a.pitch;
    const uint32_t rs = (uint32_t)a.pitch;
    const int64_t st_off = tile_off(g0, a.pitch, a.shift, T::S), code_off = tile_off(g0, a.pitch, a.shift, T::SLOTS);
    for (int d = 0; d < d_hi; ++d) {
        Pk<V> act;
        if (a.actions_in != nullptr) {
            act = bld<V, kAuxCached>(make_srd(a.actions_in + (int64_t)d * wp + g0), lo, 0);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                uint32_t x = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) x |= rng[4 * k + j].action(T::A) << (8 * j);
                act.d[k] = x;
            }
        }
        {
            Pk<V> m[T::A];
            const Pk<V> bad = action_masks<T, V>(act, m);
            if (any_bad<V>(bad, w0, a.n_walks)) atomicOr(&g_status, RC_STATUS_BAD_ACTION);
            Pk<V> o[T::S];
            apply_move<T, V>(s, m, o);
#pragma unroll
            for (int i = 0; i < T::S; ++i) s[i] = o[i];
        }
        if (d < d_lo) continue;                               // another segment's depth: replayed, nothing emitted (wave-uniform)
        FamilyCodes<T, V> fam;
        ChildFlags<T, V> cf;
        if constexpr (CODE) family_codes<T, V, true>(s, fam);
        // a child can only be solved when its parent is a power of one face turn away from solved, i.e. when every sticker outside
        // that face's ring is home (not_ring == 0).  After a few random moves no walk of the wave is: skip the 12 ring tests.
        bool flags_live = false;
        if (a.child_solved) {
            // cheap necessary test first (round 3: the full preparation below was a quarter of the kernel's VALU work): a parent one
            // face turn from solved has the face OPPOSITE to the turned one untouched, so some face must be entirely home
            Pk<V> some_face_home = splat<V>(0);
            sfor<6>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                Pk<V> off = splat<V>(0);
                sfor<T::FACE>([&](auto kc) {
                    constexpr int i = f * T::FACE + decltype(kc)::value;
                    off = off | (s[i] ^ splat<V>((uint32_t)f * 0x01010101u));
                });
                some_face_home = some_face_home | done_bytes(off);
            });
            if (__any(any(some_face_home))) {
                child_flags_prepare<T, V>(s, cf);
                Pk<V> near = splat<V>(0);
                sfor<T::A / 2>([&](auto fc) { near = near | done_bytes(cf.not_ring[decltype(fc)::value]); });
                flags_live = __any(any(near));
            }
        }
        if (part == 0) {
            if (a.actions_out) bst<V, RC_OUT_AUX>(make_srd(a.actions_out + (int64_t)d * wp + g0), lo, 0, act);
            if (a.parents) {
                const __amdgpu_buffer_rsrc_t r = make_srd(a.parents + (int64_t)d * T::S * wp + st_off);
#pragma unroll
                for (int i = 0; i < T::S; ++i) bst<V, RC_OUT_AUX>(r, lo, i * rs, s[i]);
            }
            if constexpr (FAM) {
                {
                    // the FAMILY record: every used (slot, reading order) look-up as a row of its own -- 51 bytes per state instead of the
                    // 13 x 20 picked codes; rc_onehot_from_family does the per-child pick when it expands to dense
                    constexpr int NF = kFamily<T>.nf;
                    const __amdgpu_buffer_rsrc_t r = make_srd(a.family + (int64_t)d * NF * wp + tile_off(g0, a.pitch, a.shift, NF));
                    sfor<T::NC>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        sfor<6>([&](auto ic) {
                            constexpr int id = decltype(ic)::value;
                            if constexpr (corner_pair_used<T>(q, id)) bst<V, RC_OUT_AUX>(r, lo, (uint32_t)kFamily<T>.cidx[q][id] * rs, fam.c[q][id]);
                        });
                    });
                    sfor<T::NE>([&](auto qc) {
                        constexpr int q = decltype(qc)::value;
                        sfor<2>([&](auto ic) {
                            constexpr int id = decltype(ic)::value;
                            if constexpr (edge_pair_used<T>(q, id)) bst<V, RC_OUT_AUX>(r, lo, (uint32_t)kFamily<T>.eidx[q][id] * rs, fam.e[q][id]);
                        });
                    });
                }
            }
            if constexpr (CODE && !FAM) {
                if (a.parent_code) {
                    Pk<V> pc[T::SLOTS];
                    family_pick<T, V, -1>(fam, pc);
                    const __amdgpu_buffer_rsrc_t r = make_srd(a.parent_code + (int64_t)d * T::SLOTS * wp + code_off);
#pragma unroll
                    for (int p = 0; p < T::SLOTS; ++p) bst<V, RC_OUT_AUX>(r, lo, p * rs, pc[p]);
                }
            }
        }
        const ChildOut co{a.children ? a.children + (int64_t)d * T::A * T::S * wp + st_off : nullptr,
                         a.child_solved ? a.child_solved + (int64_t)d * T::A * wp + g0 : nullptr,
                         a.child_code ? a.child_code + (int64_t)d * T::A * T::SLOTS * wp + code_off : nullptr,
                         rs, a.tiles, flags_live};
        if (CODE && !FAM && co.child_code != nullptr) emit_children<T, V, CODE && !FAM, true>(s, fam, cf, part, a.parts, co, lo);
        else emit_children<T, V, false, true>(s, fam, cf, part, a.parts, co, lo);
    }
}

// -------------------------------------------------------------------------- scramble
struct ScrambleArgs {
    const uint8_t *src;            // start states (== st: in place)
    uint8_t *st;
    int64_t n, pitch;
    int depth, shift;
    uint64_t seed, stream_id;
    int64_t walk_offset;
    const uint8_t *actions_in;
    uint8_t *actions_out;
    int64_t act_pitch;
    uint8_t *done;
    float *reward;
};

template <class T>
__global__ void __launch_bounds__(kWave) k_scramble(ScrambleArgs a) {
    constexpr int V = 1;
    const int64_t g0 = (int64_t)blockIdx.x * (kWave * 4 * V);
    const uint32_t lo = threadIdx.x * (4 * V);
    const int64_t n0 = g0 + lo;
    if (n0 >= a.n) return;
    Pk<V> s[T::S];
    const __amdgpu_buffer_rsrc_t rows = make_srd(a.st + tile_off(g0, a.pitch, a.shift, T::S));
    const uint32_t rs = (uint32_t)a.pitch;
    {
        const __amdgpu_buffer_rsrc_t from = make_srd(a.src + tile_off(g0, a.pitch, a.shift, T::S));
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = bld<V, kAuxCached>(from, lo, i * rs);
    }
    WalkRng rng[4 * V];
    if (a.actions_in == nullptr) {
#pragma unroll
        for (int j = 0; j < 4 * V; ++j) rng[j].seed(a.seed, a.stream_id, (uint64_t)(a.walk_offset + n0 + j));
    }
    for (int d = 0; d < a.depth; ++d) {
        Pk<V> act;
        if (a.actions_in != nullptr) {
            act = ld<V, false>(a.actions_in + (int64_t)d * a.act_pitch + n0);
        } else {
            uint32_t x = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) x |= rng[j].action(T::A) << (8 * j);
            act.d[0] = x;
        }
        if (a.actions_out) st<V, false>(a.actions_out + (int64_t)d * a.act_pitch + n0, act);
        Pk<V> m[T::A];
        const Pk<V> bad = action_masks<T, V>(act, m);
        if (any_bad<V>(bad, n0, a.n)) atomicOr(&g_status, RC_STATUS_BAD_ACTION);
        Pk<V> o[T::S];
        apply_move<T, V>(s, m, o);
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = o[i];
    }
#pragma unroll
    for (int i = 0; i < T::S; ++i) bst<V, kAuxCached>(rows, lo, i * rs, s[i]);
    if (a.done != nullptr || a.reward != nullptr) {
        const Pk<V> dn = done_bytes(unsolved<T, V>(s));
        if (a.done) st_tail<V>(a.done, n0, a.n, dn);
        if (a.reward) store_reward<V>(a.reward, n0, a.n, dn);
    }
}

// ------------------------------------------------- the reference's reset(seed) draws, on the device
// CubeEnv.reset (cube_env.py:62-68) draws its scramble with numpy's LEGACY global generator:
//   np.random.seed(seed); np.random.randint(action_dim, size=k)
// i.e. MT19937 seeded by init_genrand(seed), then for every draw 32-bit outputs masked to the next
// power of two minus one (15 | 7) and rejected while > action_dim-1.  Reproducing that bit for bit for
// millions of envs needs the generator itself on the GPU: one lane = one env, its whole 624-word state
// in LDS (624 x 64 lanes x 4 B = 156 KiB: one wave per CU, column per lane so every access is
// conflict-free).  All lanes consume one output per iteration, so the state index -- and with it the
// twist -- stays wave-uniform; lanes only differ in how many outputs they accept.
constexpr int kMtN = 624, kMtM = 397;
constexpr uint8_t kLegacyRedo = 0xff;                          // row-0 sentinel of a lane the streaming kernel could not finish

__device__ __forceinline__ uint32_t mt_next_seed(uint32_t x, uint32_t i) { return 1812433253u * (x ^ (x >> 30)) + i; }   // init_genrand
__device__ __forceinline__ uint32_t mt_twist(uint32_t cur, uint32_t nxt, uint32_t far) {
    const uint32_t y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}
__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
    return y;
}

// The general form: the whole generator in LDS.  The twist is LAZY -- word k of a generation only needs words k, k + 1 and k + 397 of the
// previous one (k < 227) or word k - 227 of its own, so twisting in blocks of 64 right before they are consumed is the in-place
// algorithm in the same order; a reset(seed, 30) consumes ~45 outputs = one block instead of all 624 words.
// fixup != 0: the launch follows k_legacy_actions_stream and redoes only the waves in which a lane left the kLegacyRedo sentinel in
// row 0 (a grid-stride scan, 256 envs per load).
template <int A_>
__global__ void __launch_bounds__(kWave) k_legacy_actions(const uint32_t *seeds, const int32_t *counts, int count_uniform, int kmax,
                                                          int64_t n, uint8_t *actions_out, int64_t pitch, int fixup) {
    __shared__ uint32_t mt[kMtN * kWave];
    const int lane = threadIdx.x;
    const int64_t n_waves = (n + kWave - 1) / kWave;
    const int64_t step = fixup ? 4 : 1;                        // wave ids per iteration of the outer loop
    for (int64_t w0 = (int64_t)blockIdx.x * step; w0 < n_waves; w0 += (int64_t)gridDim.x * step) {
        uint64_t redo = ~0ull;
        if (fixup) {                                           // one dword per lane = row 0 of 256 envs
            const int64_t e4 = w0 * kWave + 4 * lane;
            uint32_t word = e4 < n ? *reinterpret_cast<const uint32_t *>(actions_out + e4) : 0u;
            bool hit = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) hit |= e4 + j < n && ((word >> (8 * j)) & 0xffu) == kLegacyRedo;
            redo = __ballot(hit);
            if (redo == 0) continue;
        }
        for (int64_t w = w0; w < w0 + step && w < n_waves; ++w) {
            if (fixup && ((redo >> (16 * (w - w0))) & 0xffffull) == 0) continue;
            const int64_t env = w * kWave + lane;
            const bool live = env < n;
            int want = live ? (counts ? counts[env] : count_uniform) : 0;
            want = want < 0 ? 0 : want > kmax ? kmax : want;   // counts live in device memory: never write past [kmax][pitch]
            uint32_t x = live ? seeds[env] : 0u;               // init_genrand
            mt[lane] = x;
            for (int i = 1; i < kMtN; ++i) {
                x = mt_next_seed(x, (uint32_t)i);
                mt[i * kWave + lane] = x;
            }
            constexpr uint32_t rng = A_ - 1;
            constexpr uint32_t mask = rng | rng >> 1 | rng >> 2 | rng >> 3;   // 15 for 12 actions, 7 for 6
            int idx = kMtN, twisted = kMtN, got = 0;           // numpy seeds with mti = 624: the first draw starts a generation
            while (__any(got < want)) {
                if (idx == kMtN) { idx = 0; twisted = 0; }     // new generation: nothing of it is twisted yet
                if (idx == twisted) {                          // genrand's in-place twist, the next block of (up to) 64 words
                    const int hi = twisted + 64 < kMtN ? twisted + 64 : kMtN;
                    for (int k = twisted; k < hi; ++k) {
                        const int k1 = k + 1 == kMtN ? 0 : k + 1, km = k + kMtM >= kMtN ? k + kMtM - kMtN : k + kMtM;
                        mt[k * kWave + lane] = mt_twist(mt[k * kWave + lane], mt[k1 * kWave + lane], mt[km * kWave + lane]);
                    }
                    twisted = hi;
                }
                const uint32_t v = mt_temper(mt[idx * kWave + lane]) & mask;
                ++idx;
                if (got < want && v <= rng) {                  // masked rejection (numpy _bounded_integers, legacy path)
                    actions_out[(int64_t)got * pitch + env] = (uint8_t)v;
                    ++got;
                }
            }
            if (live)
                for (int d = want; d < kmax; ++d) actions_out[(int64_t)d * pitch + env] = (uint8_t)A_;   // pad with the no-op
        }
    }
}

// The common case -- reset(seed, k) with a few dozen moves -- needs no state at all.  With x = the init_genrand chain of the seed and
// T(a, b) = genrand's twist term of (a, b), the first generation's words are
//   new[k] = x[k + 397] ^ T(x[k], x[k + 1])                                 k <  227
//   new[k] = new[k - 227] ^ T(x[k], x[k + 1])                               227 <= k < 623   (new[k - 227] itself by one of these two lines)
// so a few chain iterators (at k, k - 227, k - 454 and 397 / 170 words ahead) produce outputs 0..622 in registers: no LDS, full
// occupancy, 25-45 VALU instructions per output after a 397-step prelude.  (Word 623 needs new[0] as its neighbour and a second
// generation the whole twisted state: those stay with the LDS kernel.)  A lane that still wants draws after `limit` (<= 623) outputs
// leaves kLegacyRedo in row 0 and its wave is redone by the LDS kernel above (launched right behind this one in fixup mode): exact for
// every seed and count.
constexpr int kMtStreamMax = kMtN - 1;                          // outputs 0..622
template <int A_>
__global__ void __launch_bounds__(256) k_legacy_actions_stream(const uint32_t *seeds, const int32_t *counts, int count_uniform, int kmax,
                                                               int64_t n, uint8_t *actions_out, int64_t pitch, int limit) {
    const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool live = env < n;
    int want = live ? (counts ? counts[env] : count_uniform) : 0;
    want = want < 0 ? 0 : want > kmax ? kmax : want;
    const uint32_t x0 = live ? seeds[env] : 0u;
    uint32_t x397 = x0;
    for (uint32_t i = 1; i <= (uint32_t)kMtM; ++i) x397 = mt_next_seed(x397, i);
    constexpr uint32_t rng = A_ - 1;
    constexpr uint32_t mask = rng | rng >> 1 | rng >> 2 | rng >> 3;
    constexpr int kP1 = kMtN - kMtM, kP2 = 2 * kP1;             // 227, 454: where the recursion gains a level
    int got = 0;
    auto emit = [&](uint32_t word) {
        const uint32_t v = mt_temper(word) & mask;
        if (got < want && v <= rng) {                          // masked rejection (numpy _bounded_integers, legacy path)
            actions_out[(int64_t)got * pitch + env] = (uint8_t)v;
            ++got;
        }
    };
    uint32_t xa = x0;                                          // x[k]: runs through all three phases
    int k = 0;
    {
        uint32_t xb = x397;                                    // x[k + 397]
        for (; k < limit && k < kP1 && __any(got < want); ++k) {
            const uint32_t xa1 = mt_next_seed(xa, (uint32_t)k + 1u);
            emit(mt_twist(xa, xa1, xb));
            xa = xa1;
            xb = mt_next_seed(xb, (uint32_t)(k + kMtM) + 1u);  // (the word past x[623] is computed once and never used)
        }
    }
    const uint32_t x227 = xa;                                  // (meaningful when phase 1 ran to its end, which phases 2 and 3 require)
    if (k == kP1) {
        uint32_t xc = x0, xd = x397;                           // x[k - 227], x[k + 170]
        for (; k < limit && k < kP2 && __any(got < want); ++k) {
            const uint32_t xa1 = mt_next_seed(xa, (uint32_t)k + 1u), xc1 = mt_next_seed(xc, (uint32_t)(k - kP1) + 1u);
            emit(mt_twist(xa, xa1, mt_twist(xc, xc1, xd)));    // new[k - 227] = x[k + 170] ^ T(x[k - 227], x[k - 226])
            xa = xa1;
            xc = xc1;
            xd = mt_next_seed(xd, (uint32_t)(k + 170) + 1u);
        }
    }
    if (k == kP2) {
        uint32_t xe = x227, xf = x0, xg = x397;                // x[k - 227], x[k - 454], x[k - 57]
        for (; k < limit && k < kMtStreamMax && __any(got < want); ++k) {
            const uint32_t xa1 = mt_next_seed(xa, (uint32_t)k + 1u), xe1 = mt_next_seed(xe, (uint32_t)(k - kP1) + 1u), xf1 = mt_next_seed(xf, (uint32_t)(k - kP2) + 1u);
            const uint32_t n0 = mt_twist(xf, xf1, xg);         // new[k - 454] = x[k - 57] ^ T(x[k - 454], x[k - 453])
            emit(mt_twist(xa, xa1, mt_twist(xe, xe1, n0)));    // new[k - 227] = new[k - 454] ^ T(x[k - 227], x[k - 226])
            xa = xa1;
            xe = xe1;
            xf = xf1;
            xg = mt_next_seed(xg, (uint32_t)(k - 57) + 1u);
        }
    }
    if (!live) return;
    if (got < want) { actions_out[env] = kLegacyRedo; return; }
    for (int d = want; d < kmax; ++d) actions_out[(int64_t)d * pitch + env] = (uint8_t)A_;
}

// ------------------------------------------------------------------------ ADI targets
// One thread per (walk, depth).  Inputs are addressed by strides so that the SAME kernel serves one depth (rc_adi_targets: child_value
// [A][pitch]) and a whole group of depths straight out of the value net's output (rc_adi_targets_depths: value [G][A + 1][block_stride],
// child_solved [G][A][Wp], one weight per depth); results land walk-major, out[w * out_stride + g], the layout the replay sink keeps.
struct TargetArgs {
    const float *child_value;  int64_t cv_depth, cv_child;     // child_value[g * cv_depth + a * cv_child + w]
    const uint8_t *child_solved; int64_t cs_depth, cs_child;
    const float *parent_value; int64_t pv_depth;                // parent_value[g * pv_depth + w]
    const double *weight;      int weight_per_depth;            // weight[g] (one per depth) or weight[w] (one per walk, G == 1)
    int64_t n;
    float *target_value; int32_t *target_policy; double *error;
    int64_t out_stride;                                         // elements between two walks of an output array
};
template <int A_>
__global__ void __launch_bounds__(256) k_adi_targets(TargetArgs t) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t g = blockIdx.y;
    if (i >= t.n) return;
    const float *cv = t.child_value + g * t.cv_depth + i;
    const uint8_t *cs = t.child_solved + g * t.cs_depth + i;
    float best = 0.f;
    int arg = -1, solved_at = -1;
#pragma unroll
    for (int k = 0; k < A_; ++k) {
        const float v = cv[k * t.cv_child] + (-1.0f);                    // cube_env.py:244  value + reward
        if (cs[k * t.cs_child] && solved_at < 0) solved_at = k;          // cube_env.py:229-232  first solved child wins
        if (arg < 0 || v > best) { best = v; arg = k; }                  // torch.max: first maximal index
    }
    const float tv = solved_at >= 0 ? 1.0f : best;
    const int64_t o = i * t.out_stride + g;
    t.target_value[o] = tv;
    t.target_policy[o] = solved_at >= 0 ? solved_at : arg;
    if (t.error) t.error[o] = fabs((double)t.parent_value[g * t.pv_depth + i] - (double)tv) * t.weight[t.weight_per_depth ? g : i];  // cube_env.py:247-251
}

// ------------------------------------------------------- batch-1 facade step (latency path)
// CubeEnv.step for ONE cube (cube_env.py:71-111): the action arrives by value in the launch arguments, the new
// one-hot (dense uint8, R*C bytes), the done flag and a sequence word are written straight into a
// HOST-MAPPED pinned buffer, so the caller needs no upload, no download and no stream synchronisation:
// it polls the sequence word.  One wave; every lane carries the same (single-cube) pack, lane l then
// writes bytes 8l..8l+7 of the one-hot.  Layout of `host_out` (512 bytes): [0, R*C) one-hot, [496] done,
// [504..507] sequence (written last, after a system-scope fence).
constexpr int kFacadeDone = 496, kFacadeSeq = 504;

constexpr int kFacadeMaxActs = 60;
struct FacadeActs {
    uint32_t n;
    uint8_t a[kFacadeMaxActs];
};

// One cube is latency, not throughput: these two kernels run the reference's algorithm LITERALLY, one lane per sticker
// (doMove_3 = a gather through the permutation table, py333.py:220-222; isSolved_3, :229-233) and one lane per piece slot
// (getOP_3's hash + table look-up, :224-227), with wave shuffles (ds_bpermute) as the gather.  ~150 instructions per lane
// and two dependent memory round trips instead of the packed network's ~1400 serial VALU instructions for a single cube.
template <class T>
struct FacadeTables {
    uint8_t perm[T::A][64];            // perm[a][lane]: source lane of sticker `lane` under action a (lanes >= S: themselves)
    uint8_t slot[32][4];               // sticker indices of piece slot p (corner: 3, edge: 2), [3] = 1 for corners
    uint8_t child[T::A * T::SLOTS][4]; // the same for slot p of child a, as PARENT sticker indices: perm[a][slot[p][k]]
    constexpr FacadeTables() : perm{}, slot{}, child{} {
        for (int a = 0; a < T::A; ++a)
            for (int i = 0; i < 64; ++i) perm[a][i] = (uint8_t)(i < T::S ? kPerm<T>.v[a][i] : i);
        for (int p = 0; p < T::SLOTS; ++p) {
            if (p < T::NC) { slot[p][0] = T::cdef[p][0]; slot[p][1] = T::cdef[p][1]; slot[p][2] = T::cdef[p][2]; slot[p][3] = 1; }
            else { slot[p][0] = T::edef[p - T::NC][0]; slot[p][1] = T::edef[p - T::NC][1]; slot[p][2] = 0; slot[p][3] = 0; }
        }
        for (int a = 0; a < T::A; ++a)
            for (int p = 0; p < T::SLOTS; ++p) {
                for (int k = 0; k < 3; ++k) child[a * T::SLOTS + p][k] = kPerm<T>.v[a][slot[p][k]];
                child[a * T::SLOTS + p][3] = slot[p][3];
            }
    }
};
__constant__ FacadeTables<Cube3> c_facade3{};
__constant__ FacadeTables<Cube2> c_facade2{};
template <class T> __device__ __forceinline__ const FacadeTables<T> &facade_tables();
template <> __device__ __forceinline__ const FacadeTables<Cube3> &facade_tables<Cube3>() { return c_facade3; }
template <> __device__ __forceinline__ const FacadeTables<Cube2> &facade_tables<Cube2>() { return c_facade2; }

// code of one piece slot from its three (two) sticker colours: hash + the 72-byte LUT held in literals (lut72)
template <class T>
__device__ __forceinline__ uint32_t slot_code(uint32_t c0, uint32_t c1, uint32_t c2, bool corner) {
    Pk<1> hc, he;
    hc.d[0] = c0 + 2u * c1 + 10u * c2;                                   // py333.py:167,225
    he.d[0] = c0 + 10u * c1;                                              // py333.py:168,226
    const uint32_t cc = lut72<T, 1, true>(hc).d[0] & 0xffu;
    if constexpr (T::NE == 0) return cc;
    else return corner ? cc : (lut72<T, 1, false>(he).d[0] & 0xffu);
}

// lane l writes bytes 8l .. 8l+7 of one dense uint8 one-hot [R][C]; `code` holds slot p's code in lane p
template <class T>
__device__ __forceinline__ void facade_write_onehot(uint32_t code, uint8_t *out, int lane) {
    constexpr int RC_ = T::R * T::C;
    const int e0 = lane * 8;
    uint32_t w[2] = {0u, 0u};
    if constexpr (T::SIZE == 3) {                                         // row = slot, column = code; 8 | 24: one row per lane
        const int r = e0 < RC_ ? e0 / T::C : 0;
        const uint32_t d = (uint32_t)__shfl((int)code, r) - (uint32_t)(e0 - r * T::C);
        if (d < 8u) w[d >> 2] = 1u << (8u * (d & 3u));
    } else {                                                              // row = piece, column = slot*3 + ori (cube_env.py:143-147)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + j < RC_ ? e0 + j : 0;
            const int r = e / T::C, col = e - r * T::C, sl = col / 3, ori = col - sl * 3;
            const uint32_t cs = (uint32_t)__shfl((int)code, sl);
            if (e0 + j < RC_ && cs == (uint32_t)(r * 3 + ori)) w[j >> 2] |= 1u << (8 * (j & 3));
        }
    }
    if (e0 + 8 <= RC_) {
        u32x2 u = {w[0], w[1]};
        *reinterpret_cast<u32x2 *>(out + e0) = u;
    } else {
        for (int j = 0; e0 + j < RC_; ++j) out[e0 + j] = (uint8_t)(w[j >> 2] >> (8 * (j & 3)));
    }
}

template <class T>
__global__ void __launch_bounds__(kWave) k_facade_step(uint8_t *st, uint32_t pitch, FacadeActs acts, uint8_t *host_out, uint32_t seq) {
    const FacadeTables<T> &tb = facade_tables<T>();
    const int lane = threadIdx.x;
    const int i = lane < T::S ? lane : 0;
    uint32_t v = st[(uint32_t)i * pitch];                                 // lane i = sticker i of cube 0
    for (uint32_t k = 0; k < acts.n; ++k) {                               // the moves of one tree descent, in one launch
        const uint32_t a = acts.a[k];                                     // wave-uniform
        if (a < (uint32_t)T::A) v = (uint32_t)__shfl((int)v, tb.perm[a][lane]);   // new[i] = old[moveDefs[a][i]]
        else if (a != (uint32_t)T::A && lane == 0) atomicOr(&g_status, RC_STATUS_BAD_ACTION);   // A itself is the no-op
    }
    if (lane < T::S) st[(uint32_t)lane * pitch] = (uint8_t)v;
    if (seq == 0) return;                                                 // an intermediate launch of a long path: state only
    const uint32_t first = (uint32_t)__shfl((int)v, (i / T::FACE) * T::FACE);
    const bool solved = __all(lane >= T::S || v == first);               // every face equals its first sticker
    const int p = lane < T::SLOTS ? lane : 0;
    const uint32_t c0 = (uint32_t)__shfl((int)v, tb.slot[p][0]), c1 = (uint32_t)__shfl((int)v, tb.slot[p][1]), c2 = (uint32_t)__shfl((int)v, tb.slot[p][2]);
    const uint32_t code = slot_code<T>(c0, c1, c2, tb.slot[p][3] != 0);
    facade_write_onehot<T>(code, host_out, lane);
    if (lane == 0) host_out[kFacadeDone] = solved ? 1 : 0;
    __threadfence_system();
    __syncthreads();
    if (lane == 0) __hip_atomic_store(reinterpret_cast<uint32_t *>(host_out + kFacadeSeq), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The node expansion of a single-root tree search (mcts.py:83-113: 12 env.step + 13 deepcopy per leaf) for ONE cube,
// results straight into host-mapped memory like k_facade_step: [0, SLOTS) the cube's own compact code (its node key),
// [32 + a * SLOTS ...) the code of child a, [288 + a] solved flag of child a, [504..507] sequence word; with `dense`
// also [512 + a * R*C ...) the dense uint8 one-hot of child a (what get_target_value feeds the value net).
constexpr int kFacadeChildCode = 32, kFacadeChildDone = 288, kFacadeDense = 512;

template <class T>
__global__ void __launch_bounds__(kWave) k_facade_expand(const uint8_t *st, uint32_t pitch, uint8_t *host_out, uint32_t seq, int dense) {
    __shared__ __attribute__((aligned(4))) uint8_t out_lds[320];
    const FacadeTables<T> &tb = facade_tables<T>();
    const int lane = threadIdx.x;
    const int i = lane < T::S ? lane : 0;
    const uint32_t v = st[(uint32_t)i * pitch];
    for (int w = lane; w < 320 / 4; w += kWave) reinterpret_cast<uint32_t *>(out_lds)[w] = 0u;
    __syncthreads();
    {   // own code
        const int p = lane < T::SLOTS ? lane : 0;
        const uint32_t c0 = (uint32_t)__shfl((int)v, tb.slot[p][0]), c1 = (uint32_t)__shfl((int)v, tb.slot[p][1]), c2 = (uint32_t)__shfl((int)v, tb.slot[p][2]);
        const uint32_t code = slot_code<T>(c0, c1, c2, tb.slot[p][3] != 0);
        if (lane < T::SLOTS) out_lds[lane] = (uint8_t)code;
    }
    // child codes: one lane per (child, slot) pair, A * SLOTS pairs in rounds of 64
    for (int q0 = 0; q0 < T::A * T::SLOTS; q0 += kWave) {
        const int q = q0 + lane < T::A * T::SLOTS ? q0 + lane : 0;
        const uint32_t c0 = (uint32_t)__shfl((int)v, tb.child[q][0]), c1 = (uint32_t)__shfl((int)v, tb.child[q][1]), c2 = (uint32_t)__shfl((int)v, tb.child[q][2]);
        const uint32_t code = slot_code<T>(c0, c1, c2, tb.child[q][3] != 0);
        if (q0 + lane < T::A * T::SLOTS) out_lds[kFacadeChildCode + q0 + lane] = (uint8_t)code;
    }
    // child solved flags: child a's sticker i is the parent's sticker perm[a][i]
    const int f0 = (i / T::FACE) * T::FACE;
#pragma unroll
    for (int a = 0; a < T::A; ++a) {
        const uint32_t cv = (uint32_t)__shfl((int)v, tb.perm[a][lane]);
        const uint32_t first = (uint32_t)__shfl((int)cv, f0);
        const bool solved = __all(lane >= T::S || cv == first);
        if (lane == 0) out_lds[kFacadeChildDone + a] = solved ? 1 : 0;
    }
    __syncthreads();
    for (int w = lane; w < 320 / 4; w += kWave) reinterpret_cast<uint32_t *>(host_out)[w] = reinterpret_cast<const uint32_t *>(out_lds)[w];
    if (dense) {
        constexpr int RC_ = T::R * T::C;                                  // 480 | 147 bytes per child
        for (int w = lane; w < (T::A * RC_ + 3) / 4; w += kWave) {
            uint32_t word = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = 4 * w + j;
                if (e < T::A * RC_) {
                    const int a = e / RC_, x = e - a * RC_, r = x / T::C, col = x - r * T::C;
                    const uint8_t *code = out_lds + kFacadeChildCode + a * T::SLOTS;
                    bool one;
                    if constexpr (T::SIZE == 3) one = code[r] == col;
                    else { const int slot = col / 3, ori = col - slot * 3; one = code[slot] == r * 3 + ori; }
                    if (one) word |= 1u << (8 * j);
                }
            }
            reinterpret_cast<uint32_t *>(host_out + kFacadeDense)[w] = word;
        }
    }
    __threadfence_system();
    __syncthreads();
    if (lane == 0) __hip_atomic_store(reinterpret_cast<uint32_t *>(host_out + kFacadeSeq), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------- lockstep search: results per root (N2)
// One expansion launch leaves leaf codes [tile][SLOTS][pitch], child codes [A][tile][SLOTS][pitch] and flags [A][tiles * pitch]; the
// host trees (include/rubiktree.h: rc_tree_update) want one record per root.  One thread per (root, block) with block = child 0..A-1
// or A = the leaf itself: SLOTS strided byte reads (coalesced across the roots of a wave), SLOTS contiguous bytes out.
template <class T>
__global__ void __launch_bounds__(256) k_search_pack(const uint8_t *__restrict__ leaf_code, const uint8_t *__restrict__ child_code,
                                                     const uint8_t *__restrict__ child_solved, int64_t n, int64_t pitch, int shift, int64_t tiles,
                                                     uint8_t *__restrict__ leaf_out, uint8_t *__restrict__ child_out, uint8_t *__restrict__ solved_out) {
    const int64_t cube = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int blk = blockIdx.y;                                            // 0..A-1 children, A = the leaf
    if (cube >= n) return;
    const int64_t tile = shift >= 63 ? 0 : cube >> shift, col = cube - tile * pitch;
    const uint8_t *src = (blk < T::A ? child_code + ((int64_t)blk * tiles + tile) * T::SLOTS * pitch : leaf_code + tile * T::SLOTS * pitch) + col;
    uint8_t *dst = blk < T::A ? child_out + (cube * T::A + blk) * T::SLOTS : leaf_out + cube * T::SLOTS;
#pragma unroll
    for (int p = 0; p < T::SLOTS; ++p) dst[p] = src[(int64_t)p * pitch];
    if (blk < T::A) solved_out[cube * T::A + blk] = child_solved[(int64_t)blk * tiles * pitch + cube];
}

// read-and-clear of the status word in one atomic, result into host-mapped memory
__global__ void k_read_status(uint32_t *host_word) { *host_word = atomicExch(&g_status, 0u); }

// ------------------------------------------------------------------------- host side
thread_local char t_err[256] = "";

int fail(int code, const char *fmt, const char *detail = "") {
    snprintf(t_err, sizeof t_err, fmt, detail);
    return code;
}
#define RC_HIP(call)                                                              \
    do {                                                                          \
        hipError_t e_ = (call);                                                   \
        if (e_ != hipSuccess) return fail(RC_EHIP, #call ": %s", hipGetErrorString(e_)); \
    } while (0)

inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline bool bad_pitch(int64_t pitch, int64_t n) { return pitch < n || (pitch & 15) != 0; }  // single-tile buffers
// state / code buffers may be tiled: one tile (pitch >= n, pitch % 16 == 0) or several
// (pitch a power of two >= kMinTile = 512, the widest wave span).  Returns the shift for tile_off, or -1 if the pitch is bad.
// Row offsets inside a tile are 32-bit scalar offsets of buffer instructions: rows * pitch < 2^32
// (a single tile of more than 79 M 3x3x3 cubes has to be split into tiles).
constexpr int64_t kMinTile = 512;
inline int tile_shift(int64_t pitch, int64_t n, int rows = 54) {
    if (pitch <= 0 || (pitch & 15) != 0 || pitch * rows >= ((int64_t)1 << 32)) return -1;
    if (n <= pitch) return 63;
    if (pitch < kMinTile || (pitch & (pitch - 1)) != 0) return -1;
    int sh = 0;
    while (((int64_t)1 << sh) < pitch) ++sh;
    return sh;
}
inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }
inline bool grid_ok(int64_t blocks) { return blocks > 0 && blocks <= 0x7fffffff; }
#define RC_GRID(blocks) \
    do { if (!grid_ok(blocks)) return fail(RC_EINVAL, "too many cubes for one launch%s"); } while (0)

// rc_init bookkeeping: every launching entry point returns RC_ENODEV until rc_init(device) has succeeded for the CURRENT device
// (a plain-C caller that skips it would otherwise see a raw HIP launch error, or run on a device that is not gfx950).
std::atomic<uint64_t> g_inited[4];   // one bit per device ordinal (256 devices)
int need_init() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return fail(RC_ENODEV, "no current HIP device%s");
    if (dev < 0 || dev >= 256 || !((g_inited[dev >> 6].load(std::memory_order_acquire) >> (dev & 63)) & 1))
        return fail(RC_ENODEV, "rc_init was not called for the current device%s");
    return RC_OK;
}
#define RC_NEED_INIT() do { if (int rc_ = need_init()) return rc_; } while (0)

template <class F>
int by_size(int cube_size, F &&f) {
    if (cube_size == 3) return f(Cube3{});
    if (cube_size == 2) return f(Cube2{});
    return fail(RC_EINVAL, "cube_size must be 2 or 3%s");  // NotImplementedError, cube_env.py:44
}

// Per-call tuning override of the *_ex entry points (0 = the measured defaults).  Decimal digits:
//   units      pack width: 1,2 -> V = 1,2 (4, 8 cubes per lane)
//   tens       row-traffic policy of the step kernel: 1 -> POL 2 (stream), 2 -> POL 0 (cached), 3 -> POL 1 (keep)
//   thousands  (2 digits) parts per walk group for expansion / ADI (1..A)
//   100000s    dense one-hot tile: 1 -> 64, 2 -> 256 cubes per workgroup
// Measured on MI355X at 4M cubes (tools/exp/exp_step2.hip; round 1's exp_step.hip is in the git history): V = 2 (8 cubes per lane, dwordx2 rows)
// beats V = 1; V = 4 (16 cubes per lane, 345 VGPRs) never beat V = 2 and is no longer instantiated.  The row-traffic
// policy follows the working set (RowPolicy).
// Round-3 sweep over 2^18 .. 2^24 cubes x {done, reward, code, in place} (profiles/r03_ab.json): with every side store a
// contiguous 1-KiB instruction V = 2 wins or ties everywhere from 2^18 cubes up (V = 1 is 2 % ahead only around 2^21).
// Which decimal fields of `variant` an entry point defines (include/rubikhip.h RC_VARIANT_*): anything else is RC_EINVAL, in the
// *_ex launchers and in rc_describe_dispatch alike, so a value composed for one entry point cannot silently mean something else in another.
int check_variant(int op, int A, int variant) {
    if (variant == 0) return RC_OK;
    const int units = variant % 10, tens = (variant / 10) % 10, hundreds = (variant / 100) % 10, field = (variant / 1000) % 100,
              form = (variant / 100000) % 10, segs = (variant / 1000000) % 100, rest = variant / 100000000;
    bool ok = variant > 0 && rest == 0;
    if (op == RC_OP_STEP) ok = ok && units <= 2 && tens <= 4 && hundreds == 0 && field == 0 && form <= 2 && segs == 0;
    else if (op == RC_OP_EXPAND) ok = ok && units <= 2 && tens == 0 && hundreds <= 8 && field <= A && form == 0 && segs == 0;
    else if (op == RC_OP_ADI) ok = ok && units <= 2 && tens == 0 && hundreds == 0 && field <= A && form == 0 && segs <= 16;
    else if (op == RC_OP_CODE_TO_DENSE) {
        ok = ok && hundreds == 0 && segs == 0 && form <= 4;
        if (form == 3) ok = ok && units == 0;                                               // wide: tens = skew, field = groups / 16
        else if (form == 0 || form == 4) ok = ok && (units <= 2 || units == 4) && (tens == 0 || (tens >= 2 && tens <= 4)) && field == 0;   // front
        else ok = ok && units == 0 && tens == 0 && field == 0;
    } else ok = false;                                                                       // RC_OP_FAMILY_TO_DENSE: no tuning fields
    return ok ? RC_OK : fail(RC_EINVAL, "variant: a field this entry point does not define is set (include/rubikhip.h RC_VARIANT_*)%s");
}

int pick_v(int64_t n, int variant) {
    const int v = variant % 10;
    if (v == 1 || v == 2) return v;
    return n >= (int64_t)1 << 18 ? 2 : 1;
}
constexpr int64_t kMallBytes = (int64_t)240 << 20;   // what we count on of the 256 MiB Infinity Cache
int pick_policy(int64_t in_bytes, int64_t out_bytes, bool in_place, int64_t side_bytes, int variant) {
    const int p = (variant / 10) % 10;
    if (p == 1) return 2;
    if (p == 2) return 0;
    if (p == 3) return 1;
    if (p == 4) return 3;
    const int64_t touched = in_place ? in_bytes : in_bytes + out_bytes;
    if (touched + side_bytes <= kMallBytes) return 0;   // resident: default-cached (1M cubes run out of the Infinity Cache)
    if (touched <= kMallBytes) return 3;                // the state is resident, its side outputs (code, reward, done) stream past it
    if (!in_place && out_bytes > 0 && out_bytes <= kMallBytes) return 1;   // stream the input, keep the output for the next launch
    return 2;
}
// bytes of the outputs no later launch of the env loop reads back: compact code, done flags, reward
template <class T>
int64_t side_bytes(int64_t n, bool code, bool done, bool reward) { return n * ((code ? T::SLOTS : 0) + (done ? 1 : 0) + (reward ? 4 : 0)); }

struct StepPlan { int v, pol; };
// One place decides pack width and row-traffic policy of a step launch (the launcher and rc_describe_dispatch both call it).
template <class T>
StepPlan plan_step(int64_t n, bool writes, bool in_place, bool code, bool done, bool reward, int variant) {
    const int pol = pick_policy(n * T::S, writes ? n * T::S : 0, writes && in_place, side_bytes<T>(n, code, done, reward), variant);
    return {pick_v(n, variant), pol};
}

template <class T, int V, bool MOVE, bool STORE, bool CODE>
int launch_step(const StepArgs &a, hipStream_t st, int pol) {
    constexpr int BLOCK = 64;
    const int64_t lanes = (a.n + 4 * V - 1) / (4 * V);
    const int64_t blocks = (lanes + BLOCK - 1) / BLOCK;
    RC_GRID(blocks);
    const dim3 g((unsigned)blocks), b(BLOCK);
    if (pol == 4) {
        if constexpr (CODE) hipLaunchKernelGGL((k_step<T, V, MOVE, STORE, CODE, 4, BLOCK>), g, b, 0, st, a);
        else return fail(RC_EINVAL, "row policy 4 belongs to the workspace route%s");
    } else if (pol == 3) hipLaunchKernelGGL((k_step<T, V, MOVE, STORE, CODE, 3, BLOCK>), g, b, 0, st, a);
    else if (pol == 2) hipLaunchKernelGGL((k_step<T, V, MOVE, STORE, CODE, 2, BLOCK>), g, b, 0, st, a);
    else if (pol == 1) hipLaunchKernelGGL((k_step<T, V, MOVE, STORE, CODE, 1, BLOCK>), g, b, 0, st, a);
    else hipLaunchKernelGGL((k_step<T, V, MOVE, STORE, CODE, 0, BLOCK>), g, b, 0, st, a);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

// keep_code: the compact code is read back by the next launch (workspace route): beyond the resident case it is written with POL 4
template <class T, bool MOVE, bool STORE, bool CODE>
int dispatch_step(const StepArgs &a, hipStream_t st, int variant, bool keep_code = false) {
    const bool writes = STORE && a.out != nullptr;
    StepPlan p = plan_step<T>(a.n, writes, writes && a.out == a.in, CODE, a.done != nullptr, a.reward != nullptr, variant);
    if (keep_code && p.pol != 0) p.pol = 4;
    return p.v == 2 ? launch_step<T, 2, MOVE, STORE, CODE>(a, st, p.pol) : launch_step<T, 1, MOVE, STORE, CODE>(a, st, p.pol);
}

// Dense writer form.  100000s digit of `variant`: 1 -> 64-cube tiles, 2 -> 256-cube tiles (256-thread workgroups), 3 -> the wide
// form (3x3x3 only).  Default: wide from 2^17 cubes (3x3x3), 256-cube tiles for 2x2x2 batches of that size, else 64-cube tiles
// (small batches -- MCTS leaves -- stay spread over the chip: 4096 cubes are 64 workgroups).
// Measured at 1M cubes (tools/microbench.py densetile): 256-cube tiles 5.4 TB/s (f32) / 6.3 TB/s (u8) against
// 4.9 / 5.4 with 1024-cube tiles -- shorter private write streams per workgroup; the wide form: see k_code_to_dense_wide.
enum DenseForm { kDense64 = 64, kDense256 = 256, kDenseWide = 960, kDenseFront = 1 };
template <class T>
inline DenseForm dense_form(int64_t n, int variant, bool fused, int fmt) {
    const bool wide_ok = T::SIZE == 3 && !fused;   // the wide form exists for code -> dense only: a FUSED wide kernel (every wave
    //   produces a tile, then all sweep 15 tiles) was built and measured in round 3 and lost to the 256-thread form for every
    //   format and group count (bf16 0.62-0.74 against 0.74-0.76: its ~110 workgroups funnel the state traffic and stall the
    //   dense stream while they produce; profiles/r03_ab.json "dense_wide_groups")
    const int forced = (variant / 100000) % 10;
    if (forced == 1) return kDense64;
    if (forced == 2) return kDense256;
    if (forced == 3) return wide_ok ? kDenseWide : kDense256;
    if (forced == 4) return wide_ok ? kDenseFront : kDense256;
    // Measured over 2^15 .. 2^22 cubes, two buffers each (profiles/r04_dense_sizes.json, fraction of the 8 TB/s peak):
    //   code -> dense: the front writer from 2^15 (f32: 0.73 against 0.63), 2^16 (16-bit: 0.69 against 0.65), 2^18 (u8: 0.75 against
    //                  0.72) cubes -- 0.83-0.95 / 0.77-0.89 / 0.75-0.82 beyond, on every allocation; 64-cube tiles below
    //   fused step   : (without a workspace; with one see step_common) 64-cube tiles below 2^17 cubes; from there 16-bit formats
    //                  64-cube tiles from 2^19 (0.78-0.80 against 0.73-0.76), 256-cube tiles otherwise
    if (!fused && wide_ok && n >= ((int64_t)1 << (fmt == RC_FMT_F32 ? 15 : fmt == RC_FMT_U8 ? 18 : 16))) return kDenseFront;
    if (n < ((int64_t)1 << 17) || (!fused && wide_ok)) return kDense64;
    if (!fused) return T::SIZE == 2 && fmt != RC_FMT_U8 ? kDense64 : kDense256;    // 2x2x2 code -> dense (profiles/r05_dense222.json): 64-cube tiles for
    //   f32 (0.66 against 0.58 at 1M cubes) and the 16-bit formats (0.53 against 0.48), 256-cube tiles for u8 (0.52 against 0.34)
    return T::SIZE == 3 && (fmt == RC_FMT_F16 || fmt == RC_FMT_BF16) && n >= ((int64_t)1 << 19) ? kDense64 : kDense256;
}
struct WideGrid { int64_t groups, per; };
// `variant` thousands field (2 digits, otherwise the expansion's parts): wanted workgroups / 16, for tuning sweeps
inline WideGrid wide_grid(int64_t n, int variant) {
    const int64_t tiles = (n + kWideTile - 1) / kWideTile;
    const int f = (variant / 1000) % 100;
    const int64_t want = f ? f * 16 : kWideGroups;
    const int64_t per = (tiles + want - 1) / want;
    return {(tiles + per - 1) / per, per};
}

// The dense kernels loop over tiles (grid-stride).  f32 rows (1920 B per cube) run 5 % faster at 1M cubes when the grid is
// capped at the 2048 workgroups the chip holds at once (8 per CU), each taking tiles b, b + 2048, ...: 330 us against 347
// (tools/exp/dense_exp.py); the 1- and 2-byte formats show no gain and keep one workgroup per tile.
inline int64_t dense_grid(int64_t blocks, int fmt) { return fmt == RC_FMT_F32 && blocks > 2048 ? 2048 : blocks; }

template <class T, bool MOVE, bool STORE, int TILE>
int launch_dense_t(const StepArgs &a, void *onehot, int fmt, hipStream_t st) {
    int64_t blocks = (a.n + TILE - 1) / TILE;
    RC_GRID(blocks);
    blocks = dense_grid(blocks, fmt);
    const dim3 g((unsigned)blocks);
    if (fmt == RC_FMT_U8) hipLaunchKernelGGL((k_step_dense<T, uint8_t, MOVE, STORE, TILE>), g, dim3(kDenseThreads<T, uint8_t>), 0, st, a, static_cast<uint8_t *>(onehot));
    else if (fmt == RC_FMT_F16) hipLaunchKernelGGL((k_step_dense<T, uint16_t, MOVE, STORE, TILE>), g, dim3(kDenseThreads<T, uint16_t>), 0, st, a, static_cast<uint16_t *>(onehot));
    else if (fmt == RC_FMT_BF16) hipLaunchKernelGGL((k_step_dense<T, Bf16, MOVE, STORE, TILE>), g, dim3(kDenseThreads<T, Bf16>), 0, st, a, static_cast<Bf16 *>(onehot));
    else hipLaunchKernelGGL((k_step_dense<T, float, MOVE, STORE, TILE>), g, dim3(kDenseThreads<T, float>), 0, st, a, static_cast<float *>(onehot));
    RC_HIP(hipGetLastError());
    return RC_OK;
}

template <class T, bool MOVE, bool STORE>
int launch_dense(const StepArgs &a, void *onehot, int fmt, hipStream_t st, int variant) {
    switch (dense_form<T>(a.n, variant, true, fmt)) {
        case kDense64: return launch_dense_t<T, MOVE, STORE, 64>(a, onehot, fmt, st);
        default: return launch_dense_t<T, MOVE, STORE, 256>(a, onehot, fmt, st);
    }
}

template <class T, int TILE>
int launch_code_to_dense(const uint8_t *code, int64_t n, int64_t code_pitch, int sh, void *onehot, int fmt, hipStream_t st,
                         int n_blocks = 1, int64_t src_bs = 0, int64_t dst_bs = 0) {
    int64_t blocks = (n + TILE - 1) / TILE;
    RC_GRID(blocks);
    blocks = dense_grid(blocks, fmt);
    const dim3 g((unsigned)blocks, (unsigned)n_blocks);
    if (fmt == RC_FMT_U8) hipLaunchKernelGGL((k_code_to_dense<T, uint8_t, TILE>), g, dim3(kDenseThreads<T, uint8_t>), 0, st, code, n, code_pitch, sh, static_cast<uint8_t *>(onehot), src_bs, dst_bs);
    else if (fmt == RC_FMT_F16) hipLaunchKernelGGL((k_code_to_dense<T, uint16_t, TILE>), g, dim3(kDenseThreads<T, uint16_t>), 0, st, code, n, code_pitch, sh, static_cast<uint16_t *>(onehot), src_bs, dst_bs);
    else if (fmt == RC_FMT_BF16) hipLaunchKernelGGL((k_code_to_dense<T, Bf16, TILE>), g, dim3(kDenseThreads<T, Bf16>), 0, st, code, n, code_pitch, sh, static_cast<Bf16 *>(onehot), src_bs, dst_bs);
    else hipLaunchKernelGGL((k_code_to_dense<T, float, TILE>), g, dim3(kDenseThreads<T, float>), 0, st, code, n, code_pitch, sh, static_cast<float *>(onehot), src_bs, dst_bs);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

// Shape of the front writer per format (kernel comment; measured side by side at 2^20 cubes, four buffers each): f32 a byte gather
// per lane, one front per XCD (0.95); 16-bit wave 0's load + LDS, one front (0.87); u8 LDS, two fronts per XCD per workgroup
// (0.82-0.86; one front 0.76).  `variant`: units digit 1, 2, 4 force F; tens digit 2 = one linear front, 3 = gather, 4 = LDS.
struct FrontShape { int f; bool lds, linear; };
inline FrontShape front_shape(int fmt, int variant) {
    const int v = variant % 10, t = (variant / 10) % 10;
    FrontShape s{fmt == RC_FMT_U8 ? 2 : 1, fmt != RC_FMT_F32, t == 2};
    if (v == 1 || v == 2 || v == 4) s.f = v;
    if (t == 3) s.lds = false;
    if (t == 4) s.lds = true;
    if (s.linear) s.f = 1;
    return s;
}
template <class T, class E, int F, bool LDS>
int launch_front_e(const uint8_t *code, int64_t n, int64_t code_pitch, int sh, E *onehot, hipStream_t st, bool linear) {
    constexpr int cpp = 240 / (480 / (16 / (int)sizeof(E)));                        // cubes per 3840-byte pass: 2 / 4 / 8
    const int64_t passes = (n + cpp - 1) / cpp;
    int64_t per_xcd = 0, per_front = 0, blocks = passes;
    if (!linear) {
        per_front = (passes + 8 * F - 1) / (8 * F);                                  // passes of one front
        per_xcd = per_front * F;
        blocks = per_front * 8;
    }
    RC_GRID(blocks);
    hipLaunchKernelGGL((k_code_to_dense_front<T, E, F, LDS>), dim3((unsigned)blocks), dim3(256), 0, st, code, n, code_pitch, sh, onehot, per_xcd, per_front);
    RC_HIP(hipGetLastError());
    return RC_OK;
}
template <class T, class E>
int launch_front_shape(const uint8_t *code, int64_t n, int64_t code_pitch, int sh, E *onehot, hipStream_t st, FrontShape s) {
    if (s.lds) {
        if (s.f == 4) return launch_front_e<T, E, 4, true>(code, n, code_pitch, sh, onehot, st, s.linear);
        if (s.f == 2) return launch_front_e<T, E, 2, true>(code, n, code_pitch, sh, onehot, st, s.linear);
        return launch_front_e<T, E, 1, true>(code, n, code_pitch, sh, onehot, st, s.linear);
    }
    if (s.f == 4) return launch_front_e<T, E, 4, false>(code, n, code_pitch, sh, onehot, st, s.linear);
    if (s.f == 2) return launch_front_e<T, E, 2, false>(code, n, code_pitch, sh, onehot, st, s.linear);
    return launch_front_e<T, E, 1, false>(code, n, code_pitch, sh, onehot, st, s.linear);
}
// family rows -> the dense one-hots of all A children and the parent (block a at onehot + a * block_stride cubes): the front writer
// over (A + 1) * ceil(n / cubes per pass) passes, same shapes per format as the code -> dense launch
struct FamilyDepths { int n_depths; int64_t src_depth_stride, dst_depth_stride; };   // strides: bytes of one depth's record / cubes of one depth's blocks
template <class T, class E, int F, bool LDS>
int launch_family_e(const uint8_t *fam, int64_t n, int64_t pitch, int sh, E *onehot, int64_t block_stride, FamilyDepths dp, hipStream_t st) {
    constexpr int cpp = 240 / (480 / (16 / (int)sizeof(E)));
    const int64_t ppb = (n + cpp - 1) / cpp;                                      // passes per block
    const int64_t per_front = (ppb + 8 * F - 1) / (8 * F), per_xcd = per_front * F, blocks = per_front * 8;   // gridDim.x % 8 == 0: x % 8 is the XCD in every row
    RC_GRID(blocks);
    const int64_t nblk = (int64_t)(T::A + 1) * dp.n_depths;
    if (nblk > 65535) return fail(RC_EINVAL, "rc_onehot_from_family: at most 5041 depths per launch%s");
    const FamilyBlocks fb{block_stride, dp.src_depth_stride, dp.dst_depth_stride};
    hipLaunchKernelGGL((k_code_to_dense_front<T, E, F, LDS, true>), dim3((unsigned)blocks, (unsigned)nblk), dim3(256), 0, st, fam, n, pitch, sh, onehot, per_xcd, per_front, fb);
    RC_HIP(hipGetLastError());
    return RC_OK;
}
template <class T>
int launch_family_to_dense(const uint8_t *fam, int64_t n, int64_t pitch, int sh, void *onehot, int fmt, int64_t block_stride, FamilyDepths dp, hipStream_t st) {
    if (fmt == RC_FMT_U8) return launch_family_e<T, uint8_t, 2, true>(fam, n, pitch, sh, static_cast<uint8_t *>(onehot), block_stride, dp, st);
    if (fmt == RC_FMT_F16) return launch_family_e<T, uint16_t, 1, true>(fam, n, pitch, sh, static_cast<uint16_t *>(onehot), block_stride, dp, st);
    if (fmt == RC_FMT_BF16) return launch_family_e<T, Bf16, 1, true>(fam, n, pitch, sh, static_cast<Bf16 *>(onehot), block_stride, dp, st);
    return launch_family_e<T, float, 1, false>(fam, n, pitch, sh, static_cast<float *>(onehot), block_stride, dp, st);
}

template <class T>
int launch_code_to_dense_front(const uint8_t *code, int64_t n, int64_t code_pitch, int sh, void *onehot, int fmt, hipStream_t st, int variant) {
    if constexpr (T::SIZE == 3) {
        const FrontShape s = front_shape(fmt, variant);
        if (fmt == RC_FMT_U8) return launch_front_shape<T>(code, n, code_pitch, sh, static_cast<uint8_t *>(onehot), st, s);
        if (fmt == RC_FMT_F16) return launch_front_shape<T>(code, n, code_pitch, sh, static_cast<uint16_t *>(onehot), st, s);
        if (fmt == RC_FMT_BF16) return launch_front_shape<T>(code, n, code_pitch, sh, static_cast<Bf16 *>(onehot), st, s);
        return launch_front_shape<T>(code, n, code_pitch, sh, static_cast<float *>(onehot), st, s);
    } else {
        return launch_code_to_dense<T, 256>(code, n, code_pitch, sh, onehot, fmt, st);
    }
}

template <class T>
int launch_code_to_dense_wide(const uint8_t *code, int64_t n, int64_t code_pitch, int sh, void *onehot, int fmt, hipStream_t st, int variant) {
    if constexpr (T::SIZE == 3) {
        const WideGrid w = wide_grid(n, variant);
        RC_GRID(w.groups);
        const int skew = (variant / 10) % 10 ? (variant / 10) % 10 : kWideSkew;
        const dim3 g((unsigned)w.groups), b(kWideBlock);
        if (fmt == RC_FMT_U8) hipLaunchKernelGGL((k_code_to_dense_wide<T, uint8_t>), g, b, 0, st, code, n, code_pitch, sh, static_cast<uint8_t *>(onehot), w.per, skew);
        else if (fmt == RC_FMT_F16) hipLaunchKernelGGL((k_code_to_dense_wide<T, uint16_t>), g, b, 0, st, code, n, code_pitch, sh, static_cast<uint16_t *>(onehot), w.per, skew);
        else if (fmt == RC_FMT_BF16) hipLaunchKernelGGL((k_code_to_dense_wide<T, Bf16>), g, b, 0, st, code, n, code_pitch, sh, static_cast<Bf16 *>(onehot), w.per, skew);
        else hipLaunchKernelGGL((k_code_to_dense_wide<T, float>), g, b, 0, st, code, n, code_pitch, sh, static_cast<float *>(onehot), w.per, skew);
        RC_HIP(hipGetLastError());
        return RC_OK;
    } else {
        return launch_code_to_dense<T, 256>(code, n, code_pitch, sh, onehot, fmt, st);
    }
}

int check_fmt(void *onehot, int fmt, int64_t code_pitch, int64_t n, int *sh_code) {
    if (fmt < RC_FMT_NONE || fmt > RC_FMT_BF16) return fail(RC_EINVAL, "unknown one-hot format%s");
    if ((fmt == RC_FMT_NONE) != (onehot == nullptr)) return fail(RC_EINVAL, "onehot pointer and fmt disagree%s");
    if (onehot && !aligned16(onehot)) return fail(RC_EINVAL, "onehot must be 16-byte aligned%s");
    *sh_code = 63;
    if (fmt == RC_FMT_CODE && (*sh_code = tile_shift(code_pitch, n, 20)) < 0) return fail(RC_EINVAL, "code_pitch: need pitch %% 16 == 0 and pitch >= n_cubes, or a power-of-two tile >= 512%s");
    return RC_OK;
}

// Expansion / ADI launch geometry (measured on MI355X, profiles/r02_design_ab.json).  A large write-once output
// stream is fastest with FEW waves issuing WIDE stores: 100k walks x 30 run 0.32 ms with 196 waves of 8 walks per lane
// against 0.41-0.43 ms with 2346 waves of 4 walks per lane; 16 walks per lane would be better still for the stores alone
// (7.2 TB/s in the store-only harness) but one wave per 1024 walks cannot hide the walk's VALU work (0.44 ms).  Small
// batches are latency-bound instead: spread them over the chip.  Code-only expansion is VALU-bound: narrow packs.
struct Geometry { int v, parts, segs; };
constexpr int64_t kExpandStreamGrid = 512;   // 1M parents: 113 us (0.83) with 512 waves, 114 (256), 115 (1024), 116.4 us for 2048 short-lived waves
constexpr double kReplayCostCodes = 0.2, kReplayCostStickers = 0.35;   // replayed depth / emitted depth (RNG + move against everything)
Geometry pick_geometry(int64_t n, int A, int variant, bool stickers_out, int64_t out_bytes) {
    const int fv = variant % 10, fp = (variant / 1000) % 100;
    const bool stream = stickers_out && out_bytes >= ((int64_t)256 << 20);
    const int64_t want = stream ? 96 : stickers_out ? 2048 : 700;
    int v = 1;
    if (fv == 1 || fv == 2) v = fv;
    else if (stream && n >= want * kWave * 8) v = 2;
    const int64_t groups = (n + kWave * 4 * v - 1) / (kWave * 4 * v);
    int parts = 1;
    if (fp >= 1 && fp <= A) parts = fp;
    else while (parts < A && groups * parts < want) ++parts;
    while (A % parts) ++parts;
    return {v, parts, 1};
}
// Code-only ADI (no child stickers): one wave per walk group leaves most SIMDs idle or alone with a long serial chain, and
// `parts` duplicates the code look-ups.  DEPTH SEGMENTS split the work without duplicating them; measured at 100k walks x 30
// (three repeats, profiles/r03_ab.json): 8 walks per lane x 3 segments (588 waves) 148-149 us every time, 4 walks per lane x
// 2 segments (782 waves) 146-160 us, round 2's 4 walks per lane x 2 parts 162-166 us; more segments lose to the replayed
// moves.  Rule: about 600 waves; wide packs once that still leaves >= 2 segments' worth of groups.
// FAMILY records (118 B per state instead of 327): with a third of the stores the launch is VALU-bound and wants MORE, narrower
// waves -- 4 walks per lane and about 1560 waves (100k x 30: 82 us against 109 us with 8 walks per lane x 4 segments and 89 us x 5;
// 1M x 4: 86 us against 103 us; profiles/r04_adi_family.json); small batches keep the 700-wave rule (20k x 30: 36 us).
Geometry pick_geometry_adi(int64_t n, int A, int variant, bool stickers_out, int64_t out_bytes, bool codes, bool family = false) {
    Geometry g = pick_geometry(n, A, variant, stickers_out, out_bytes);
    if (codes && !stickers_out) {
        const int fv = variant % 10, fp = (variant / 1000) % 100;
        g.v = fv == 1 || fv == 2 ? fv : (n >= 64 * kWave * 8 && !family ? 2 : 1);
        g.parts = fp >= 1 && fp <= A ? g.parts : 1;
        const int64_t waves = (n + kWave * 4 * g.v - 1) / (kWave * 4 * g.v) * g.parts;
        const int64_t target = !family ? 600 : n >= 64 * kWave * 8 ? 1560 : 700;
        const int64_t segs = (target + waves / 2) / (waves > 0 ? waves : 1);
        g.segs = segs < 1 ? 1 : segs > kMaxSegs ? kMaxSegs : (int)segs;
    }
    return g;
}

// Depth segments of the ADI kernel (k_adi): boundaries that equalise the work of the segments when a replayed depth
// costs `replay` of an emitted one (segment s replays seg_lo[s] depths and emits seg_lo[s+1] - seg_lo[s]).
void fill_segments(AdiArgs &a, int segs, double replay) {
    if (segs > a.depth) segs = a.depth;
    if (segs > kMaxSegs) segs = kMaxSegs;
    if (segs < 1) segs = 1;
    a.segs = segs;
    double lo_c = 0, hi_c = a.depth;                        // bisection on the per-segment cost
    for (int it = 0; it < 60; ++it) {
        const double c = 0.5 * (lo_c + hi_c);
        double lo = 0;
        for (int k = 0; k < segs; ++k) lo += c - replay * lo > 0 ? c - replay * lo : 0;
        (lo < a.depth ? lo_c : hi_c) = c;
    }
    double lo = 0;
    int prev = 0;
    a.seg_lo[0] = 0;
    for (int k = 1; k <= segs; ++k) {
        lo += hi_c - replay * lo;
        int b = k == segs ? a.depth : (int)(lo + 0.5);
        if (b < prev + 1) b = prev + 1;                      // every segment emits at least one depth ...
        if (b > a.depth - (segs - k)) b = a.depth - (segs - k);   // ... and leaves one for each later segment
        a.seg_lo[k] = (uint16_t)b;
        prev = b;
    }
}

// grid of the streaming expansion: hundreds digit of `variant` 1..7 -> 128, 192, 256, 384, 512, 768, 1024 waves; 8 -> not streamed
// out_bytes = n * S * A of the cube size at hand (1M 2x2x2 parents write 151 MB: below the threshold, they keep k_expand -- the
// streaming form was only measured on 3x3x3).  A forced `parts` value (thousands field, 1 included) or pack width 1 selects k_expand.
inline int64_t expand_stream_grid(int64_t out_bytes, bool stickers, bool codes, int variant) {
    static const int table[8] = {0, 128, 192, 256, 384, 512, 768, 1024};
    const int h = (variant / 100) % 10;
    if (!stickers || codes || h == 8 || (variant % 10) == 1 || (variant / 1000) % 100 >= 1) return 0;
    if (h >= 1 && h <= 7) return table[h];                         // forced (tests reach the kernel with small batches too)
    return out_bytes >= ((int64_t)256 << 20) ? kExpandStreamGrid : 0;   // large write-once streams only
}

template <class T>
int launch_expand_stream(ExpandArgs a, hipStream_t st, int64_t grid) {
    const int64_t groups = (a.n + kWave * 8 - 1) / (kWave * 8);
    if (grid > groups) grid = groups;
    hipLaunchKernelGGL((k_expand_stream<T>), dim3((unsigned)grid), dim3(kWave), 0, st, a);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

template <class T, int V>
int launch_expand(ExpandArgs a, hipStream_t st) {
    const int64_t groups = (a.n + kWave * 4 * V - 1) / (kWave * 4 * V);
    RC_GRID(groups * a.parts);
    const dim3 g((unsigned)(groups * a.parts)), b(kWave);
    if (a.child_code) hipLaunchKernelGGL((k_expand<T, V, true>), g, b, 0, st, a);
    else hipLaunchKernelGGL((k_expand<T, V, false>), g, b, 0, st, a);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

template <class T, int V>
int launch_adi(AdiArgs a, hipStream_t st) {
    const int64_t groups = (a.n_walks + kWave * 4 * V - 1) / (kWave * 4 * V);
    RC_GRID(groups * a.parts * a.segs);
    const dim3 g((unsigned)(groups * a.parts * a.segs)), b(kWave);
    if (a.family) hipLaunchKernelGGL((k_adi<T, V, true, true>), g, b, 0, st, a);      // (adi_common: no codes beside the family record)
    else if (a.parent_code || a.child_code) hipLaunchKernelGGL((k_adi<T, V, true>), g, b, 0, st, a);
    else hipLaunchKernelGGL((k_adi<T, V, false>), g, b, 0, st, a);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

int rc_version(void) { return 600; }

// the hash of the sources this binary was compiled from (__graft_entry__.build passes -DRC_SRC_HASH=<16 hex digits>); a plain
// `hipcc -c` of this file still compiles and reports "unhashed", which the Python binding refuses as stale
#ifndef RC_SRC_HASH
#define RC_SRC_HASH unhashed
#endif
#define RC_STR2(x) #x
#define RC_STR(x) RC_STR2(x)
static const char k_build_id[] = "rc-build-id:" RC_STR(RC_SRC_HASH);
const char *rc_build_id(void) { return k_build_id + 12; }

const char *rc_last_error(void) { return t_err; }

int rc_init(int device) {
    int count = 0, prev = 0;
    RC_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return fail(RC_ENODEV, "no such HIP device%s");
    hipDeviceProp_t prop;
    RC_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(RC_ENODEV, "librubikhip is built for gfx950 only, device is %s", prop.gcnArchName);
    // clear the device's status word; the caller's current device is left as it was
    RC_HIP(hipGetDevice(&prev));
    RC_HIP(hipSetDevice(device));
    uint32_t zero = 0;
    const hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(g_status), &zero, sizeof zero);
    RC_HIP(hipSetDevice(prev));
    RC_HIP(e);
    if (device < 256) g_inited[device >> 6].fetch_or((uint64_t)1 << (device & 63), std::memory_order_release);
    return RC_OK;
}

int rc_get_tables(int cube_size, uint8_t *perm, uint8_t *solved, uint8_t *corner_defs, uint8_t *edge_defs,
                  uint8_t *corner_code, uint8_t *edge_code, int32_t dims[6]) {
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        if (perm) memcpy(perm, kPerm<T>.v, sizeof kPerm<T>.v);
        if (solved) for (int i = 0; i < T::S; ++i) solved[i] = (uint8_t)(i / T::FACE);
        if (corner_defs) memcpy(corner_defs, T::cdef, (size_t)T::NC * 3);
        if (edge_defs && T::NE) memcpy(edge_defs, T::edef, (size_t)T::NE * 2);
        if (corner_code) memcpy(corner_code, T::ccode, 72);
        if (edge_code) memcpy(edge_code, T::ecode, 72);
        if (dims) { dims[0] = T::S; dims[1] = T::A; dims[2] = T::NC; dims[3] = T::NE; dims[4] = T::R; dims[5] = T::C; }
        return RC_OK;
    });
}

int rc_fill_solved(uint8_t *stp, int64_t n, int64_t pitch, int cube_size, void *stream) {
    RC_NEED_INIT();
    const int sh = tile_shift(pitch, n);
    if (!stp || !aligned16(stp) || n < 0 || sh < 0) return fail(RC_EINVAL, "rc_fill_solved: bad buffer / pitch%s");
    if (n == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        const int64_t blocks = (n + kWave * 8 - 1) / (kWave * 8);
        RC_GRID(blocks);
        hipLaunchKernelGGL((k_fill_solved<T>), dim3((unsigned)blocks), dim3(kWave), 0, S(stream), stp, n, pitch, sh);
        RC_HIP(hipGetLastError());
        return RC_OK;
    });
}

// Two-launch dense route (rc_apply_moves_ws): the step kernel writes the compact code into the caller's workspace (one tile,
// [SLOTS][ws_pitch]), the front writer expands it.  3x3x3 only, from kFrontMin cubes; 0 = not applicable.
// Measured over 2^17 .. 2^22 cubes (profiles/r04_dense_sizes.json): f32 0.78-0.92 of peak on every allocation against 0.63-0.86
// (placement dependent) for the one-launch kernel, 16-bit formats 0.65-0.85 against 0.62-0.80.  u8 gains nothing while its state
// ping-pong fits the Infinity Cache (0.80 against 0.81 at 2^20) and takes the route from 2^22 cubes only (0.79 against 0.65).
constexpr int64_t kFrontMin = (int64_t)1 << 17, kFrontMinU8 = (int64_t)1 << 22, kWsTile = 32768;     // workspace = a tiled code buffer [tile][SLOTS][32768]
inline int64_t dense_workspace_bytes(int cube_size, int64_t n, int fmt) {
    if (cube_size != 3 || fmt < RC_FMT_U8 || fmt > RC_FMT_BF16 || n < (fmt == RC_FMT_U8 ? kFrontMinU8 : kFrontMin)) return 0;
    return (n + kWsTile - 1) / kWsTile * kWsTile * 20;
}

static int step_common(const uint8_t *in, uint8_t *out, const uint8_t *actions, int64_t n, int64_t pitch_in, int64_t pitch_out,
                       int cube_size, float *reward, uint8_t *done, void *onehot, int fmt, int64_t code_pitch, void *stream,
                       bool move, bool store, int variant, void *workspace = nullptr, int64_t workspace_bytes = 0) {
    RC_NEED_INIT();
    if (int rc = check_variant(RC_OP_STEP, 12, variant)) return rc;
    const int sh_in = tile_shift(pitch_in, n), sh_out = store ? tile_shift(pitch_out, n) : 63;
    int sh_code = 63;
    if (!in || !aligned16(in) || n < 0 || sh_in < 0) return fail(RC_EINVAL, "bad input state buffer / pitch%s");
    if (store && (!out || !aligned16(out) || sh_out < 0)) return fail(RC_EINVAL, "bad output state buffer / pitch%s");
    if (move && !actions) return fail(RC_EINVAL, "actions is NULL%s");
    if (reward && (reinterpret_cast<uintptr_t>(reward) & 15u)) return fail(RC_EINVAL, "reward must be 16-byte aligned%s");
    if (int rc = check_fmt(onehot, fmt, code_pitch, n, &sh_code)) return rc;
    if (n == 0) return RC_OK;
    StepArgs a{in, out, actions, n, pitch_in, pitch_out, reward, done, fmt == RC_FMT_CODE ? static_cast<uint8_t *>(onehot) : nullptr, code_pitch,
               sh_in, sh_out, sh_code};
    hipStream_t st = S(stream);
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        if (fmt >= RC_FMT_U8) {
            if constexpr (T::SIZE == 3) {
                const int64_t need = dense_workspace_bytes(3, n, fmt);
                if (workspace && need > 0 && workspace_bytes >= need && (variant / 100000) % 10 == 0) {
                    // the workspace is written by the first launch and read by the second while `onehot` is being written: it must not
                    // share a byte with any operand (a caller carving it out of the one-hot allocation would get silently wrong rows)
                    const auto hits = [&](const void *p, int64_t bytes) {
                        const uintptr_t w0 = reinterpret_cast<uintptr_t>(workspace), p0 = reinterpret_cast<uintptr_t>(p);
                        return p != nullptr && bytes > 0 && p0 < w0 + (uintptr_t)need && w0 < p0 + (uintptr_t)bytes;
                    };
                    const auto state_bytes = [&](int64_t pitch) { return (n <= pitch ? 1 : (n + pitch - 1) / pitch) * T::S * pitch; };
                    const int64_t esize = fmt == RC_FMT_U8 ? 1 : fmt == RC_FMT_F32 ? 4 : 2;
                    if (hits(in, state_bytes(pitch_in)) || (store && hits(out, state_bytes(pitch_out))) || hits(actions, n) || hits(reward, 4 * n) ||
                        hits(done, n) || hits(onehot, n * T::R * T::C * esize))
                        return fail(RC_EINVAL, "workspace overlaps an input or output buffer%s");
                    // step (or encode) + reward + done + compact code (into the workspace), then the front writer: the dense stream
                    // leaves as one sweeping window instead of thousands of private 240-KiB streams (k_code_to_dense_front)
                    StepArgs a2 = a;
                    a2.code = static_cast<uint8_t *>(workspace);
                    a2.code_pitch = kWsTile;
                    a2.sh_code = 15;                                                 // log2(kWsTile)
                    if (int rc = move ? dispatch_step<T, true, true, true>(a2, st, variant, true) : dispatch_step<T, false, false, true>(a2, st, variant, true)) return rc;
                    return launch_code_to_dense_front<T>(a2.code, n, a2.code_pitch, a2.sh_code, onehot, fmt, st, 0);
                }
            }
            if (move) return launch_dense<T, true, true>(a, onehot, fmt, st, variant);
            return launch_dense<T, false, false>(a, onehot, fmt, st, variant);
        }
        if (move) {
            if (fmt == RC_FMT_CODE) return dispatch_step<T, true, true, true>(a, st, variant);
            return dispatch_step<T, true, true, false>(a, st, variant);
        }
        if (fmt == RC_FMT_CODE) return dispatch_step<T, false, false, true>(a, st, variant);
        return dispatch_step<T, false, false, false>(a, st, variant);
    });
}

int rc_apply_moves_ex(const uint8_t *in, uint8_t *out, const uint8_t *actions, int64_t n, int64_t pitch_in, int64_t pitch_out,
                      int cube_size, float *reward, uint8_t *done, void *onehot, int fmt, int64_t code_pitch, void *stream, int variant) {
    return step_common(in, out, actions, n, pitch_in, pitch_out, cube_size, reward, done, onehot, fmt, code_pitch, stream, true, true, variant);
}

int rc_apply_moves(const uint8_t *in, uint8_t *out, const uint8_t *actions, int64_t n, int64_t pitch_in, int64_t pitch_out,
                   int cube_size, float *reward, uint8_t *done, void *onehot, int fmt, int64_t code_pitch, void *stream) {
    return step_common(in, out, actions, n, pitch_in, pitch_out, cube_size, reward, done, onehot, fmt, code_pitch, stream, true, true, 0);
}

int64_t rc_workspace_bytes(int op, int cube_size, int64_t n, int fmt) {
    if (op != RC_OP_STEP || n <= 0) return 0;
    return dense_workspace_bytes(cube_size, n, fmt);
}

int rc_apply_moves_ws(const uint8_t *in, uint8_t *out, const uint8_t *actions, int64_t n, int64_t pitch_in, int64_t pitch_out,
                      int cube_size, float *reward, uint8_t *done, void *onehot, int fmt, int64_t code_pitch, void *workspace,
                      int64_t workspace_bytes, void *stream) {
    if (workspace && !aligned16(workspace)) return fail(RC_EINVAL, "workspace must be 16-byte aligned%s");
    if (workspace_bytes < 0) return fail(RC_EINVAL, "workspace_bytes is negative%s");
    return step_common(in, out, actions, n, pitch_in, pitch_out, cube_size, reward, done, onehot, fmt, code_pitch, stream, true, true, 0,
                       workspace, workspace_bytes);
}

int rc_scramble_from(const uint8_t *src, uint8_t *stp, int64_t n, int64_t pitch, int cube_size, int depth, uint64_t seed, uint64_t stream_id,
                     int64_t walk_offset, const uint8_t *actions_in, uint8_t *actions_out, int64_t act_pitch, uint8_t *done,
                     float *reward, void *stream) {
    RC_NEED_INIT();
    const int sh = tile_shift(pitch, n);
    if (!stp || !aligned16(stp) || !src || !aligned16(src) || n < 0 || depth < 0 || sh < 0) return fail(RC_EINVAL, "rc_scramble: bad state buffer / pitch%s");
    if ((actions_in || actions_out) && bad_pitch(act_pitch, n)) return fail(RC_EINVAL, "rc_scramble: bad act_pitch%s");
    if ((actions_in && !aligned16(actions_in)) || (actions_out && !aligned16(actions_out))) return fail(RC_EINVAL, "rc_scramble: action buffers must be 16-byte aligned%s");
    if (reward && (reinterpret_cast<uintptr_t>(reward) & 15u)) return fail(RC_EINVAL, "reward must be 16-byte aligned%s");
    if (n == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        ScrambleArgs a{src, stp, n, pitch, depth, sh, seed, stream_id, walk_offset, actions_in, actions_out, act_pitch, done, reward};
        const int64_t blocks = (n + kWave * 4 - 1) / (kWave * 4);
        RC_GRID(blocks);
        hipLaunchKernelGGL((k_scramble<T>), dim3((unsigned)blocks), dim3(kWave), 0, S(stream), a);
        RC_HIP(hipGetLastError());
        return RC_OK;
    });
}

int rc_scramble(uint8_t *stp, int64_t n, int64_t pitch, int cube_size, int depth, uint64_t seed, uint64_t stream_id,
                int64_t walk_offset, const uint8_t *actions_in, uint8_t *actions_out, int64_t act_pitch, uint8_t *done,
                float *reward, void *stream) {
    return rc_scramble_from(stp, stp, n, pitch, cube_size, depth, seed, stream_id, walk_offset, actions_in, actions_out, act_pitch, done, reward, stream);
}

int rc_search_pack(const uint8_t *leaf_code, const uint8_t *child_code, const uint8_t *child_solved, int64_t n, int64_t pitch, int cube_size,
                   uint8_t *leaf_out, uint8_t *child_out, uint8_t *solved_out, void *stream) {
    RC_NEED_INIT();
    const int sh = tile_shift(pitch, n, 20);
    if (!leaf_code || !child_code || !child_solved || !leaf_out || !child_out || !solved_out || n < 0 || sh < 0) return fail(RC_EINVAL, "rc_search_pack: bad arguments%s");
    if (n == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        const int64_t blocks = (n + 255) / 256, tiles = n <= pitch ? 1 : (n + pitch - 1) / pitch;
        RC_GRID(blocks);
        hipLaunchKernelGGL((k_search_pack<T>), dim3((unsigned)blocks, T::A + 1), dim3(256), 0, S(stream), leaf_code, child_code, child_solved, n, pitch, sh, tiles,
                           leaf_out, child_out, solved_out);
        RC_HIP(hipGetLastError());
        return RC_OK;
    });
}

// Which generator form runs (`variant` of rc_legacy_scramble_actions_ex; include/rubikhip.h RC_VARIANT_LEGACY_*): 0 = by kmax,
// 1 = the LDS kernel alone, 2 + 16 * limit = the streaming kernel with `limit` outputs per lane (1..623; 0 = 623) + the fixup launch.
// Default rule: a draw is accepted with probability 3/4, so k draws consume k / 0.75 outputs on average with a standard deviation of
// sqrt(k) * 0.67: up to kmax = 400 (533 +- 13) the 623 streamed outputs are never short in practice, and when they are the fixup
// launch redoes that wave -- the rule only decides which kernel does the bulk.
constexpr int kLegacyStreamMax = 400;
int rc_legacy_scramble_actions_ex(const uint32_t *seeds, const int32_t *counts, int count_uniform, int kmax, int64_t n, int cube_size,
                                   uint8_t *actions_out, int64_t pitch, void *stream, int variant) {
    RC_NEED_INIT();
    if (!seeds || !actions_out || n < 0 || kmax < 0 || bad_pitch(pitch, n)) return fail(RC_EINVAL, "rc_legacy_scramble_actions: bad arguments%s");
    if (!aligned16(actions_out)) return fail(RC_EINVAL, "rc_legacy_scramble_actions: actions_out must be 16-byte aligned%s");
    if (!counts && (count_uniform < 0 || count_uniform > kmax)) return fail(RC_EINVAL, "rc_legacy_scramble_actions: count_uniform must be in 0..kmax%s");
    if (cube_size != 2 && cube_size != 3) return fail(RC_EINVAL, "cube_size must be 2 or 3%s");
    const int mode = variant & 15, lim = variant >> 4;
    if (variant < 0 || mode > 2 || lim > kMtStreamMax || (mode != 2 && lim != 0)) return fail(RC_EINVAL, "rc_legacy_scramble_actions: unknown variant%s");
    if (n == 0 || kmax == 0) return RC_OK;
    const int64_t waves = (n + kWave - 1) / kWave;
    RC_GRID(waves);
    const bool streamed = mode == 2 || (mode == 0 && kmax <= kLegacyStreamMax);
    if (streamed) {
        const dim3 g((unsigned)((n + 255) / 256)), b(256);
        const int limit = lim ? lim : kMtStreamMax;
        if (cube_size == 3) hipLaunchKernelGGL((k_legacy_actions_stream<12>), g, b, 0, S(stream), seeds, counts, count_uniform, kmax, n, actions_out, pitch, limit);
        else hipLaunchKernelGGL((k_legacy_actions_stream<6>), g, b, 0, S(stream), seeds, counts, count_uniform, kmax, n, actions_out, pitch, limit);
        RC_HIP(hipGetLastError());
    }
    const int64_t chunks = (waves + 3) / 4;
    const dim3 g((unsigned)(streamed ? (chunks < 1024 ? chunks : 1024) : waves)), b(kWave);
    if (cube_size == 3) hipLaunchKernelGGL((k_legacy_actions<12>), g, b, 0, S(stream), seeds, counts, count_uniform, kmax, n, actions_out, pitch, streamed ? 1 : 0);
    else hipLaunchKernelGGL((k_legacy_actions<6>), g, b, 0, S(stream), seeds, counts, count_uniform, kmax, n, actions_out, pitch, streamed ? 1 : 0);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

int rc_legacy_scramble_actions(const uint32_t *seeds, const int32_t *counts, int count_uniform, int kmax, int64_t n, int cube_size,
                                uint8_t *actions_out, int64_t pitch, void *stream) {
    return rc_legacy_scramble_actions_ex(seeds, counts, count_uniform, kmax, n, cube_size, actions_out, pitch, stream, 0);
}

int rc_is_solved(const uint8_t *stp, int64_t n, int64_t pitch, int cube_size, uint8_t *done, float *reward, void *stream) {
    if (!done && !reward) return fail(RC_EINVAL, "rc_is_solved: nothing to write%s");
    return step_common(stp, nullptr, nullptr, n, pitch, 0, cube_size, reward, done, nullptr, RC_FMT_NONE, 0, stream, false, false, 0);
}

int rc_encode(const uint8_t *stp, int64_t n, int64_t pitch, int cube_size, void *onehot, int fmt, int64_t code_pitch, void *stream) {
    if (fmt == RC_FMT_NONE) return fail(RC_EINVAL, "rc_encode: fmt must not be RC_FMT_NONE%s");
    return step_common(stp, nullptr, nullptr, n, pitch, 0, cube_size, nullptr, nullptr, onehot, fmt, code_pitch, stream, false, false, 0);
}

int rc_encode_ws(const uint8_t *stp, int64_t n, int64_t pitch, int cube_size, void *onehot, int fmt, int64_t code_pitch, void *workspace,
                 int64_t workspace_bytes, void *stream) {
    if (fmt == RC_FMT_NONE) return fail(RC_EINVAL, "rc_encode: fmt must not be RC_FMT_NONE%s");
    if (workspace && !aligned16(workspace)) return fail(RC_EINVAL, "workspace must be 16-byte aligned%s");
    if (workspace_bytes < 0) return fail(RC_EINVAL, "workspace_bytes is negative%s");
    return step_common(stp, nullptr, nullptr, n, pitch, 0, cube_size, nullptr, nullptr, onehot, fmt, code_pitch, stream, false, false, 0, workspace, workspace_bytes);
}

int rc_onehot_from_code_ex(const uint8_t *code, int64_t n, int64_t code_pitch, int cube_size, void *onehot, int fmt, void *stream, int variant) {
    RC_NEED_INIT();
    if (int rc = check_variant(RC_OP_CODE_TO_DENSE, 12, variant)) return rc;
    const int sh = tile_shift(code_pitch, n, 20);
    if (!code || !aligned16(code) || n < 0 || sh < 0) return fail(RC_EINVAL, "bad code buffer / pitch%s");
    if (fmt < RC_FMT_U8 || fmt > RC_FMT_BF16 || !onehot || !aligned16(onehot)) return fail(RC_EINVAL, "rc_onehot_from_code: dense fmt and aligned buffer required%s");
    if (n == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        switch (dense_form<T>(n, variant, false, fmt)) {
            case kDenseFront: return launch_code_to_dense_front<T>(code, n, code_pitch, sh, onehot, fmt, S(stream), variant);
            case kDenseWide: return launch_code_to_dense_wide<T>(code, n, code_pitch, sh, onehot, fmt, S(stream), variant);
            case kDense256: return launch_code_to_dense<T, 256>(code, n, code_pitch, sh, onehot, fmt, S(stream));
            default: return launch_code_to_dense<T, 64>(code, n, code_pitch, sh, onehot, fmt, S(stream));
        }
    });
}

int rc_onehot_from_code(const uint8_t *code, int64_t n, int64_t code_pitch, int cube_size, void *onehot, int fmt, void *stream) {
    return rc_onehot_from_code_ex(code, n, code_pitch, cube_size, onehot, fmt, stream, 0);
}

int rc_onehot_from_code_blocks(const uint8_t *code, int64_t n, int64_t code_pitch, int cube_size, void *onehot, int fmt, int n_blocks,
                               int64_t src_block_stride, int64_t dst_block_stride, void *stream) {
    RC_NEED_INIT();
    const int sh = tile_shift(code_pitch, n, 20);
    if (!code || !aligned16(code) || n < 0 || sh < 0) return fail(RC_EINVAL, "rc_onehot_from_code_blocks: bad code buffer / pitch%s");
    if (fmt < RC_FMT_U8 || fmt > RC_FMT_BF16 || !onehot || !aligned16(onehot)) return fail(RC_EINVAL, "rc_onehot_from_code_blocks: dense fmt and aligned buffer required%s");
    if (n_blocks < 1 || n_blocks > 65535 || src_block_stride < 0 || src_block_stride % 16 || dst_block_stride < n)
        return fail(RC_EINVAL, "rc_onehot_from_code_blocks: 1..65535 blocks, source stride a multiple of 16 bytes, destination stride >= n cubes%s");
    const int64_t esz = fmt == RC_FMT_U8 ? 1 : fmt == RC_FMT_F32 ? 4 : 2;
    if (n == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        if ((dst_block_stride * T::R * T::C * esz) % 16) return fail(RC_EINVAL, "rc_onehot_from_code_blocks: every block of the output must start 16-byte aligned%s");
        // the tile kernels only (blocks of a small batch: the front / wide forms pay off from 2^15 cubes per launch, and a caller with
        // blocks that large loses nothing by launching them one by one)
        if (n >= ((int64_t)1 << 17) && !(T::SIZE == 2 && fmt != RC_FMT_U8)) return launch_code_to_dense<T, 256>(code, n, code_pitch, sh, onehot, fmt, S(stream), n_blocks, src_block_stride, dst_block_stride);
        return launch_code_to_dense<T, 64>(code, n, code_pitch, sh, onehot, fmt, S(stream), n_blocks, src_block_stride, dst_block_stride);
    });
}

int rc_expand_children_ex(const uint8_t *in, int64_t n, int64_t pitch_in, int cube_size, uint8_t *children, uint8_t *child_solved,
                          uint8_t *child_code, int64_t pitch_out, void *stream, int variant) {
    RC_NEED_INIT();
    if (int rc = check_variant(RC_OP_EXPAND, cube_size == 2 ? 6 : 12, variant)) return rc;
    const int sh_in = tile_shift(pitch_in, n), sh_out = tile_shift(pitch_out, n);
    if (!in || !aligned16(in) || n < 0 || sh_in < 0 || sh_out < 0) return fail(RC_EINVAL, "rc_expand_children: bad buffer / pitch%s");
    if (!children && !child_solved && !child_code) return fail(RC_EINVAL, "rc_expand_children: nothing to write%s");
    if ((children && !aligned16(children)) || (child_solved && !aligned16(child_solved)) || (child_code && !aligned16(child_code)))
        return fail(RC_EINVAL, "rc_expand_children: outputs must be 16-byte aligned%s");
    if (n == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        const Geometry geo = pick_geometry(n, T::A, variant, children != nullptr, n * T::S * T::A);
        ExpandArgs a{in, n, pitch_in, children, child_solved, child_code, pitch_out, n <= pitch_out ? 1 : (n + pitch_out - 1) / pitch_out,
                     geo.parts, sh_in, sh_out};
        hipStream_t st = S(stream);
        if (const int64_t grid = expand_stream_grid(n * T::S * T::A, children != nullptr, child_code != nullptr, variant)) return launch_expand_stream<T>(a, st, grid);
        return geo.v == 2 ? launch_expand<T, 2>(a, st) : launch_expand<T, 1>(a, st);
    });
}

int rc_expand_children(const uint8_t *in, int64_t n, int64_t pitch_in, int cube_size, uint8_t *children, uint8_t *child_solved,
                       uint8_t *child_code, int64_t pitch_out, void *stream) {
    return rc_expand_children_ex(in, n, pitch_in, cube_size, children, child_solved, child_code, pitch_out, stream, 0);
}

static int adi_common(uint64_t seed, uint64_t stream_id, int64_t walk_offset, int64_t n_walks, int depth, int cube_size, int64_t pitch,
                      const uint8_t *actions_in, uint8_t *actions_out, uint8_t *parents, uint8_t *parent_code, uint8_t *children,
                      uint8_t *child_code, uint8_t *child_solved, uint8_t *family, void *stream, int variant) {
    RC_NEED_INIT();
    if (int rc = check_variant(RC_OP_ADI, cube_size == 2 ? 6 : 12, variant)) return rc;
    if (family && !aligned16(family)) return fail(RC_EINVAL, "rc_adi_generate: buffers must be 16-byte aligned%s");
    const int sh = tile_shift(pitch, n_walks);
    if (n_walks < 0 || depth < 0 || sh < 0) return fail(RC_EINVAL, "rc_adi_generate: bad sizes / pitch%s");
    const void *ptrs[] = {actions_in, actions_out, parents, parent_code, children, child_code, child_solved};
    for (const void *p : ptrs)
        if (p && !aligned16(p)) return fail(RC_EINVAL, "rc_adi_generate: buffers must be 16-byte aligned%s");
    if (n_walks == 0 || depth == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        const bool any_child = children || child_code || child_solved;
        Geometry geo = pick_geometry_adi(n_walks, T::A, variant, children != nullptr, n_walks * depth * T::S * T::A, parent_code || child_code || family, family != nullptr);
        if (!any_child && (variant / 1000) % 100 == 0) geo.parts = 1;
        AdiArgs a{seed, stream_id, walk_offset, n_walks, pitch, n_walks <= pitch ? 1 : (n_walks + pitch - 1) / pitch, depth,
                  geo.parts, sh, actions_in, actions_out, parents, parent_code, children, child_code, child_solved, family, 1, {}};
        if (depth > 0xffff) return fail(RC_EINVAL, "rc_adi_generate: depth must be below 65536%s");
        const int fsegs = (variant / 1000000) % 100;
        const bool codes = parent_code || child_code || family;
        // a replayed depth is RNG + move; an emitted one adds the flags, the code look-ups and the stores
        fill_segments(a, fsegs ? fsegs : geo.segs, codes ? kReplayCostCodes : kReplayCostStickers);
        hipStream_t st = S(stream);
        return geo.v == 2 ? launch_adi<T, 2>(a, st) : launch_adi<T, 1>(a, st);
    });
}

int rc_adi_generate_ex(uint64_t seed, uint64_t stream_id, int64_t walk_offset, int64_t n_walks, int depth, int cube_size, int64_t pitch,
                       const uint8_t *actions_in, uint8_t *actions_out, uint8_t *parents, uint8_t *parent_code, uint8_t *children,
                       uint8_t *child_code, uint8_t *child_solved, void *stream, int variant) {
    return adi_common(seed, stream_id, walk_offset, n_walks, depth, cube_size, pitch, actions_in, actions_out, parents, parent_code, children, child_code,
                      child_solved, nullptr, stream, variant);
}

int rc_adi_generate(uint64_t seed, uint64_t stream_id, int64_t walk_offset, int64_t n_walks, int depth, int cube_size, int64_t pitch,
                    const uint8_t *actions_in, uint8_t *actions_out, uint8_t *parents, uint8_t *parent_code, uint8_t *children,
                    uint8_t *child_code, uint8_t *child_solved, void *stream) {
    return rc_adi_generate_ex(seed, stream_id, walk_offset, n_walks, depth, cube_size, pitch, actions_in, actions_out, parents, parent_code,
                              children, child_code, child_solved, stream, 0);
}

int rc_adi_generate_family(uint64_t seed, uint64_t stream_id, int64_t walk_offset, int64_t n_walks, int depth, int cube_size, int64_t pitch,
                           const uint8_t *actions_in, uint8_t *actions_out, uint8_t *parents, uint8_t *family, uint8_t *child_solved, void *stream,
                           int variant) {
    if (!family) return fail(RC_EINVAL, "rc_adi_generate_family: family is NULL%s");
    return adi_common(seed, stream_id, walk_offset, n_walks, depth, cube_size, pitch, actions_in, actions_out, parents, nullptr, nullptr, nullptr, child_solved,
                      family, stream, variant);
}

int rc_family_layout(int cube_size, uint8_t *rows, int32_t *n_rows) {
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        if (rows) memcpy(rows, kFamily<T>.row, sizeof kFamily<T>.row);
        if (n_rows) *n_rows = kFamily<T>.nf;
        return RC_OK;
    });
}

int rc_onehot_from_family_depths(const uint8_t *family, int64_t n, int64_t pitch, int cube_size, void *onehot, int fmt, int64_t block_stride,
                                 int n_depths, void *stream) {
    RC_NEED_INIT();
    if (cube_size != 3) return fail(RC_EINVAL, "rc_onehot_from_family: 3x3x3 only%s");
    const int sh = tile_shift(pitch, n, kFamily<Cube3>.nf);
    if (!family || !aligned16(family) || n < 0 || sh < 0) return fail(RC_EINVAL, "bad family buffer / pitch%s");
    if (fmt < RC_FMT_U8 || fmt > RC_FMT_BF16 || !onehot || !aligned16(onehot)) return fail(RC_EINVAL, "rc_onehot_from_family: dense fmt and aligned buffer required%s");
    if (block_stride < n) return fail(RC_EINVAL, "rc_onehot_from_family: block_stride must be >= n_cubes%s");
    if (n_depths < 0 || n_depths > 5041) return fail(RC_EINVAL, "rc_onehot_from_family: n_depths must be in 0..5041%s");
    if (n == 0 || n_depths == 0) return RC_OK;
    const int64_t tiles = n <= pitch ? 1 : (n + pitch - 1) / pitch;
    const FamilyDepths dp{n_depths, tiles * kFamily<Cube3>.nf * pitch, (int64_t)(Cube3::A + 1) * block_stride};
    return launch_family_to_dense<Cube3>(family, n, pitch, sh, onehot, fmt, block_stride, dp, S(stream));
}

int rc_onehot_from_family(const uint8_t *family, int64_t n, int64_t pitch, int cube_size, void *onehot, int fmt, int64_t block_stride, void *stream) {
    return rc_onehot_from_family_depths(family, n, pitch, cube_size, onehot, fmt, block_stride, 1, stream);
}

static int launch_targets(const TargetArgs &t, int n_depths, int cube_size, void *stream) {
    const int64_t blocks = (t.n + 255) / 256;
    RC_GRID(blocks);
    if (n_depths > 65535) return fail(RC_EINVAL, "rc_adi_targets: at most 65535 depths per launch%s");
    const dim3 g((unsigned)blocks, (unsigned)n_depths), b(256);
    if (cube_size == 3) hipLaunchKernelGGL((k_adi_targets<12>), g, b, 0, S(stream), t);
    else if (cube_size == 2) hipLaunchKernelGGL((k_adi_targets<6>), g, b, 0, S(stream), t);
    else return fail(RC_EINVAL, "cube_size must be 2 or 3%s");
    RC_HIP(hipGetLastError());
    return RC_OK;
}

int rc_adi_targets(const float *child_value, const uint8_t *child_solved, const float *parent_value, const double *weight, int64_t n,
                   int64_t pitch, int cube_size, float *target_value, int32_t *target_policy, double *error, void *stream) {
    RC_NEED_INIT();
    if (n == 0) return RC_OK;                                  // nothing to assemble: empty outputs may be null pointers
    if (!child_value || !child_solved || !target_value || !target_policy || n < 0 || pitch < n) return fail(RC_EINVAL, "rc_adi_targets: bad arguments%s");
    if (error && (!parent_value || !weight)) return fail(RC_EINVAL, "rc_adi_targets: error needs parent_value and weight%s");
    const TargetArgs t{child_value, 0, pitch, child_solved, 0, pitch, parent_value, 0, weight, 0, n, target_value, target_policy, error, 1};
    return launch_targets(t, 1, cube_size, stream);
}

int rc_adi_targets_depths(const float *child_value, int64_t cv_depth_stride, int64_t cv_child_stride, const uint8_t *child_solved,
                          int64_t solved_pitch, const float *parent_value, int64_t pv_depth_stride, const double *weight, int64_t n,
                          int n_depths, int cube_size, float *target_value, int32_t *target_policy, double *error, int64_t out_stride,
                          void *stream) {
    RC_NEED_INIT();
    if (n == 0 || n_depths == 0) return RC_OK;
    if (!child_value || !child_solved || !target_value || !target_policy || n < 0 || n_depths < 0 || solved_pitch < n || cv_child_stride < n ||
        out_stride < n_depths)
        return fail(RC_EINVAL, "rc_adi_targets_depths: bad arguments%s");
    if (error && (!parent_value || !weight)) return fail(RC_EINVAL, "rc_adi_targets_depths: error needs parent_value and weight%s");
    const int A = cube_size == 2 ? 6 : 12;
    const TargetArgs t{child_value, cv_depth_stride, cv_child_stride, child_solved, (int64_t)A * solved_pitch, solved_pitch, parent_value, pv_depth_stride,
                       weight, 1, n, target_value, target_policy, error, out_stride};
    return launch_targets(t, n_depths, cube_size, stream);
}

static int facade_wait(uint8_t *host_out, uint32_t seq, void *stream, const char *who);

// host_out is dereferenced by the HOST while polling and written by the KERNEL through its device alias: for hipHostMalloc'ed
// memory the two addresses are equal, for hipHostRegister'ed memory (e.g. torch with pinned_use_cuda_host_register) they may
// differ, so the kernel always gets attr.devicePointer.  Validated (host, device ordinal) -> alias pairs are cached PROCESS-WIDE
// (a short table behind a mutex: ~20 ns on a 10 us path), so rc_facade_release works from any thread -- Python runs __del__ wherever
// the garbage collector happens to run.  A sequence number of 1 (a caller's first use of a buffer: every CubeEnv starts its sequence
// there) always re-validates, so an address that was freed and handed out again -- pinned or not -- is looked up afresh by its new owner.
struct FacadeAlias { const uint8_t *host; uint8_t *dev; int device; };
static std::mutex g_alias_lock;
static std::vector<FacadeAlias> g_alias;
static void alias_drop(const uint8_t *host_out) {                            // caller holds the lock; NULL = everything
    size_t k = 0;
    for (size_t i = 0; i < g_alias.size(); ++i)
        if (host_out != nullptr && g_alias[i].host != host_out) g_alias[k++] = g_alias[i];
    g_alias.resize(k);
}
static int facade_check_host(const uint8_t *host_out, uint32_t seq, const char *who, uint8_t **dev_alias) {
    int device = -1;
    if (hipGetDevice(&device) != hipSuccess) return fail(RC_ENODEV, "no current HIP device%s");
    std::lock_guard<std::mutex> hold(g_alias_lock);
    if (seq != 1)
        for (const FacadeAlias &a : g_alias)
            if (a.host == host_out && a.device == device) { *dev_alias = a.dev; return RC_OK; }
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, host_out) != hipSuccess || attr.type != hipMemoryTypeHost || attr.devicePointer == nullptr) {
        (void)hipGetLastError();
        alias_drop(host_out);
        return fail(RC_EINVAL, "%s: host_out must be host-mapped pinned memory (hipHostMalloc / torch pin_memory)", who);
    }
    for (FacadeAlias &a : g_alias)
        if (a.host == host_out && a.device == device) { a.dev = static_cast<uint8_t *>(attr.devicePointer); *dev_alias = a.dev; return RC_OK; }
    if (g_alias.size() >= 256) g_alias.erase(g_alias.begin());             // callers that never release: forget the oldest
    g_alias.push_back({host_out, static_cast<uint8_t *>(attr.devicePointer), device});
    *dev_alias = g_alias.back().dev;
    return RC_OK;
}

int rc_host_alias(const void *host, void **device_alias) {
    if (!host || !device_alias) return fail(RC_EINVAL, "rc_host_alias: NULL argument%s");
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, host) != hipSuccess || attr.type != hipMemoryTypeHost || attr.devicePointer == nullptr) {
        (void)hipGetLastError();
        return fail(RC_EINVAL, "rc_host_alias: not host-mapped pinned memory (hipHostMalloc / hipHostRegister / torch pin_memory)%s");
    }
    *device_alias = attr.devicePointer;
    return RC_OK;
}

int rc_facade_steps(uint8_t *stp, int64_t pitch, int cube_size, const uint8_t *actions, int n_actions, uint8_t *host_out, uint32_t seq,
                    int wait, void *stream) {
    RC_NEED_INIT();
    if (!stp || !host_out || pitch <= 0 || pitch * 54 >= ((int64_t)1 << 32)) return fail(RC_EINVAL, "rc_facade_steps: bad arguments%s");
    if (n_actions < 0 || (n_actions > 0 && !actions)) return fail(RC_EINVAL, "rc_facade_steps: bad action list%s");
    if (seq == 0) return fail(RC_EINVAL, "rc_facade_steps: seq must be non-zero%s");
    uint8_t *dev_out = nullptr;
    if (int rc = facade_check_host(host_out, seq, "rc_facade_steps", &dev_out)) return rc;
    const int rc = by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        int done = 0;
        do {                                                       // up to 60 moves travel in the launch arguments
            FacadeActs fa{};
            const int m = n_actions - done < kFacadeMaxActs ? n_actions - done : kFacadeMaxActs;
            fa.n = (uint32_t)m;
            memcpy(fa.a, actions + done, (size_t)m);
            done += m;
            hipLaunchKernelGGL((k_facade_step<T>), dim3(1), dim3(kWave), 0, S(stream), stp, (uint32_t)pitch, fa, dev_out, done == n_actions ? seq : 0u);
            RC_HIP(hipGetLastError());
        } while (done < n_actions);
        return RC_OK;
    });
    if (rc != RC_OK || !wait) return rc;
    return facade_wait(host_out, seq, stream, "rc_facade_steps");
}

int rc_facade_step(uint8_t *stp, int64_t pitch, int cube_size, int action, uint8_t *host_out, uint32_t seq, int wait, void *stream) {
    if (action < 0 || action > 255) return fail(RC_EINVAL, "rc_facade_step: action out of the byte range%s");
    const uint8_t a = (uint8_t)action;
    return rc_facade_steps(stp, pitch, cube_size, &a, 1, host_out, seq, wait, stream);
}

static int facade_wait(uint8_t *host_out, uint32_t seq, void *stream, const char *who) {
    // poll the sequence word the kernel writes last; fall back to a stream sync if it does not show up (e.g. host_out is not
    // host-visible memory), so a wrong buffer becomes an error instead of a hang
    volatile uint32_t *flag = reinterpret_cast<volatile uint32_t *>(host_out + kFacadeSeq);
    for (int spin = 0; spin < 20000000; ++spin) {
        if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return RC_OK;
        __builtin_ia32_pause();
    }
    RC_HIP(hipStreamSynchronize(S(stream)));
    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) == seq) return RC_OK;
    return fail(RC_EHIP, "%s: the result never reached host_out (is it host-mapped pinned memory?)", who);
}

int rc_facade_release(const uint8_t *host_out) {
    std::lock_guard<std::mutex> hold(g_alias_lock);
    alias_drop(host_out);
    return RC_OK;
}

int rc_facade_expand(const uint8_t *stp, int64_t pitch, int cube_size, uint8_t *host_out, uint32_t seq, int dense, int wait, void *stream) {
    RC_NEED_INIT();
    if (!stp || !host_out || pitch <= 0 || pitch * 54 >= ((int64_t)1 << 32)) return fail(RC_EINVAL, "rc_facade_expand: bad arguments%s");
    if (seq == 0) return fail(RC_EINVAL, "rc_facade_expand: seq must be non-zero%s");
    uint8_t *dev_out = nullptr;
    if (int rc = facade_check_host(host_out, seq, "rc_facade_expand", &dev_out)) return rc;
    const int rc = by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        hipLaunchKernelGGL((k_facade_expand<T>), dim3(1), dim3(kWave), 0, S(stream), stp, (uint32_t)pitch, dev_out, seq, dense);
        RC_HIP(hipGetLastError());
        return RC_OK;
    });
    if (rc != RC_OK || !wait) return rc;
    return facade_wait(host_out, seq, stream, "rc_facade_expand");
}

// What a call WOULD launch, from the same pick_* functions the launchers use (benchmarks label their records with this, so a
// change of the dispatch policy cannot leave a stale kernel name behind).
int rc_describe_dispatch(int op, int cube_size, int64_t n, int depth, unsigned outputs, int fmt, int variant, char *buf, int buflen) {
    if (!buf || buflen < 16) return fail(RC_EINVAL, "rc_describe_dispatch: buffer too small%s");
    if (n <= 0) return fail(RC_EINVAL, "rc_describe_dispatch: n must be positive%s");
    if (op != RC_OP_STEP && op != RC_OP_EXPAND && op != RC_OP_ADI && op != RC_OP_CODE_TO_DENSE && op != RC_OP_FAMILY_TO_DENSE)
        return fail(RC_EINVAL, "rc_describe_dispatch: unknown op%s");
    if (int rc = check_variant(op, cube_size == 2 ? 6 : 12, variant)) return rc;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        const char *cube = T::SIZE == 3 ? "Cube3" : "Cube2";
        const bool states = outputs & RC_OUT_STATES, code = outputs & RC_OUT_CODE;
        if (op == RC_OP_STEP) {
            if (fmt >= RC_FMT_U8 && fmt <= RC_FMT_BF16) {
                static const char *const names[] = {"", "", "u8", "f16", "f32", "bf16"};
                if ((outputs & RC_OUT_WORKSPACE) && dense_workspace_bytes(T::SIZE, n, fmt) > 0 && (variant / 100000) % 10 == 0) {
                    StepPlan p = plan_step<T>(n, states, states && (outputs & RC_OUT_INPLACE), true, outputs & RC_OUT_DONE, outputs & RC_OUT_REWARD, variant);
                    if (p.pol != 0) p.pol = 4;
                    const int cpp = fmt == RC_FMT_F32 ? 2 : fmt == RC_FMT_U8 ? 8 : 4;
                    const FrontShape fs = front_shape(fmt, 0);
                    snprintf(buf, buflen, "k_step<%s,V=%d%s,code,POL=%d> grid=%lld block=64 + k_code_to_dense_front<%s,%s,F=%d,%s> xcd grid=%lld block=256", cube, p.v,
                             states ? ",move,store" : "", p.pol, (long long)((n + 256 * p.v - 1) / (256 * p.v)), cube, names[fmt], fs.f, fs.lds ? "lds" : "gather",
                             (long long)(((n + cpp - 1) / cpp + 8 * fs.f - 1) / (8 * fs.f) * 8));
                    return RC_OK;
                }
                const int form = (int)dense_form<T>(n, variant, true, fmt);
                snprintf(buf, buflen, "k_step_dense<%s,%s,%s,TILE=%d> grid=%lld block=%d", cube, names[fmt], states ? "move,store" : "encode",
                         form, (long long)dense_grid((n + form - 1) / form, fmt), fmt == RC_FMT_F32 ? kDenseThreads<T, float> : kDenseThreads<T, uint8_t>);
                return RC_OK;
            }
            const bool with_code = code || fmt == RC_FMT_CODE;
            const StepPlan p = plan_step<T>(n, states, states && (outputs & RC_OUT_INPLACE), with_code, outputs & RC_OUT_DONE, outputs & RC_OUT_REWARD, variant);
            snprintf(buf, buflen, "k_step<%s,V=%d%s%s,POL=%d> grid=%lld block=64", cube, p.v, states ? ",move,store" : "", with_code ? ",code" : "", p.pol,
                     (long long)((n + 256 * p.v - 1) / (256 * p.v)));
            return RC_OK;
        }
        if (op == RC_OP_CODE_TO_DENSE) {
            static const char *const names[] = {"", "", "u8", "f16", "f32", "bf16"};
            if (fmt < RC_FMT_U8 || fmt > RC_FMT_BF16) return fail(RC_EINVAL, "rc_describe_dispatch: dense fmt required%s");
            const DenseForm form = dense_form<T>(n, variant, false, fmt);
            if (form == kDenseFront) {
                const int cpp = fmt == RC_FMT_F32 ? 2 : fmt == RC_FMT_U8 ? 8 : 4;
                const FrontShape fs = front_shape(fmt, variant);
                snprintf(buf, buflen, "k_code_to_dense_front<%s,%s,F=%d,%s> cubes_per_pass=%d%s grid=%lld block=256", cube, names[fmt], fs.f, fs.lds ? "lds" : "gather", cpp,
                         fs.linear ? "" : " xcd", (long long)(fs.linear ? (n + cpp - 1) / cpp : ((n + cpp - 1) / cpp + 8 * fs.f - 1) / (8 * fs.f) * 8));
            } else if (form == kDenseWide) {
                const WideGrid w = wide_grid(n, variant);
                snprintf(buf, buflen, "k_code_to_dense_wide<%s,%s> tiles_per_group=%lld grid=%lld block=%d", cube, names[fmt], (long long)w.per, (long long)w.groups, kWideBlock);
            } else {
                snprintf(buf, buflen, "k_code_to_dense<%s,%s,TILE=%d> grid=%lld block=%d", cube, names[fmt], (int)form, (long long)dense_grid((n + (int)form - 1) / (int)form, fmt), fmt == RC_FMT_F32 ? kDenseThreads<T, float> : kDenseThreads<T, uint8_t>);
            }
            return RC_OK;
        }
        if (op == RC_OP_FAMILY_TO_DENSE) {
            static const char *const names[] = {"", "", "u8", "f16", "f32", "bf16"};
            if (T::SIZE != 3 || fmt < RC_FMT_U8 || fmt > RC_FMT_BF16) return fail(RC_EINVAL, "rc_describe_dispatch: 3x3x3 and a dense fmt required%s");
            if (depth <= 0) return fail(RC_EINVAL, "rc_describe_dispatch: depth (the number of depths per launch) must be positive%s");
            const int cpp = fmt == RC_FMT_F32 ? 2 : fmt == RC_FMT_U8 ? 8 : 4, f = fmt == RC_FMT_U8 ? 2 : 1;   // launch_family_to_dense's shapes
            const int64_t ppb = (n + cpp - 1) / cpp;
            snprintf(buf, buflen, "k_code_to_dense_front<%s,%s,F=%d,%s,family> depths=%d cubes_per_pass=%d xcd grid=%lldx%d block=256", cube, names[fmt], f,
                     fmt == RC_FMT_F32 ? "gather" : "lds", depth, cpp, (long long)((ppb + 8 * f - 1) / (8 * f) * 8), (T::A + 1) * depth);
            return RC_OK;
        }
        if (op == RC_OP_EXPAND) {
            const Geometry geo = pick_geometry(n, T::A, variant, states, n * T::S * T::A);
            if (const int64_t grid = expand_stream_grid(n * T::S * T::A, states, code, variant)) {
                const int64_t groups = (n + 511) / 512;
                snprintf(buf, buflen, "k_expand_stream<%s> grid=%lld block=64", cube, (long long)(grid < groups ? grid : groups));
                return RC_OK;
            }
            snprintf(buf, buflen, "k_expand<%s,V=%d%s> parts=%d grid=%lld block=64", cube, geo.v, code ? ",code" : "", geo.parts,
                     (long long)((n + 256 * geo.v - 1) / (256 * geo.v) * geo.parts));
            return RC_OK;
        }
        if (op == RC_OP_ADI) {
            if (depth <= 0) return fail(RC_EINVAL, "rc_describe_dispatch: depth must be positive%s");
            const bool family = outputs & RC_OUT_FAMILY;
            Geometry geo = pick_geometry_adi(n, T::A, variant, states, n * depth * T::S * T::A, code || family, family);
            const bool any_child = states || code || (outputs & RC_OUT_FLAGS);
            if (!any_child && (variant / 1000) % 100 == 0) geo.parts = 1;
            AdiArgs a{};
            a.depth = depth;
            const int fsegs = (variant / 1000000) % 100;
            fill_segments(a, fsegs ? fsegs : geo.segs, code || family ? kReplayCostCodes : kReplayCostStickers);
            snprintf(buf, buflen, "k_adi<%s,V=%d%s> parts=%d segs=%d grid=%lld block=64", cube, geo.v, family ? ",code,family" : code ? ",code" : "", geo.parts, a.segs,
                     (long long)((n + 256 * geo.v - 1) / (256 * geo.v) * geo.parts * a.segs));
            return RC_OK;
        }
        return fail(RC_EINVAL, "rc_describe_dispatch: unknown op%s");
    });
}

int rc_read_status(uint32_t *status, void *stream) {
    RC_NEED_INIT();
    if (!status) return fail(RC_EINVAL, "status is NULL%s");
    // one atomic read-and-clear on the device (bits set by kernels still running on OTHER streams are neither lost nor
    // reported early: they show in a later read); the word travels through a pinned, host-mapped scratch word
    static thread_local uint32_t *host_word = nullptr;
    if (!host_word) RC_HIP(hipHostMalloc(reinterpret_cast<void **>(&host_word), sizeof(uint32_t), hipHostMallocMapped | hipHostMallocPortable));
    *host_word = 0xffffffffu;
    hipLaunchKernelGGL(k_read_status, dim3(1), dim3(1), 0, S(stream), host_word);
    RC_HIP(hipGetLastError());
    RC_HIP(hipStreamSynchronize(S(stream)));
    *status = *host_word;
    return RC_OK;
}

}  // extern "C"
