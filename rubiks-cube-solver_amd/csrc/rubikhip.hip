// rubikhip.hip -- kernels and C ABI of librubikhip.so (see include/rubikhip.h).
// gfx950 (MI355X, CDNA4) only; no other back end, no compatibility layer.
//
// Kernel map (DESIGN.md has the roofline of each):
//   k_fill_solved     initState_3                      py333.py:211-218
//   k_step            CubeEnv.step for N cubes          cube_env.py:71-111
//                     (move + solved/reward + compact code, all in registers)
//   k_step_dense      same + dense one-hot, the [slot][cube] code tile staged in LDS so the
//                     [N][R][C] rows leave as coalesced 16-byte stores
//   k_code_to_dense   compact code -> dense one-hot (same LDS stage)
//   k_scramble        reset()'s scramble loop, in place    cube_env.py:65-67
//   k_expand          12 children of every cube         cube_env.py:212-236, mcts.py:96-101
//   k_adi             ADI walks + expansion, persistent over depth, per-walk xoroshiro128+
//                                                       cube_env.py:177-194,212-236
//   k_adi_targets     target value/policy/error         cube_env.py:229-232,239-251
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include "../../include/rubikhip.h"
#include "rc_device.h"

using namespace rc;

#ifndef RC_OUT_NT
#define RC_OUT_NT false  // expansion / ADI output stores: default-cached measured 1.5-3 % faster than non-temporal (tools/exp/adi_ab.py)
#endif

namespace {

constexpr int kWave = 64;
__device__ uint32_t g_status;  // RC_STATUS_* bits, per device (one code object instance per device)

// Tiled structure-of-arrays addressing (include/rubikhip.h "State layout"): cube n of a buffer
// with `rows` rows lives in tile n / pitch; tiles follow each other, [tile][row][pitch].
// `shift` = log2(pitch) when the buffer has several tiles, 63 when it has one (then tile = 0 and
// any pitch % 16 == 0 works).  g0 is a wave-uniform cube index that is a multiple of the wave's
// span (<= 1024 cubes), so a wave never straddles tiles (multi-tile pitch is a multiple of 1024).
__device__ __forceinline__ int64_t tile_off(int64_t g0, int64_t pitch, int shift, int rows) {
    return g0 + (g0 >> shift) * (rows - 1) * pitch;
}

// ------------------------------------------------------------------------------ fill
template <class T>
__global__ void __launch_bounds__(kWave) k_fill_solved(uint8_t *base, int64_t n, int64_t pitch, int shift) {
    const int64_t g0 = (int64_t)blockIdx.x * (kWave * 16);
    const uint32_t lo = threadIdx.x * 16;
    if (g0 + lo >= n) return;
    uint8_t *row = base + tile_off(g0, pitch, shift, T::S);
#pragma unroll
    for (int s = 0; s < T::S; ++s) { st<4, false>(row + lo, splat<4>((uint32_t)(s / T::FACE) * 0x01010101u)); row += pitch; }
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

// One lane = 4*V consecutive cubes.  MOVE: apply actions; STORE: write the state rows;
// CODE: write the compact code rows; NT: non-temporal row traffic.  FULL: every pack of the wave lies
// inside the batch (all but the last wave): no per-byte tail paths in the instruction stream.
template <class T, int V, bool MOVE, bool STORE, bool CODE, bool NT, bool FULL>
__device__ __forceinline__ void step_body(const StepArgs &a, int64_t g0, uint32_t lo) {
    const int64_t n0 = g0 + lo;
    const int64_t n = FULL ? n0 + 4 * V : a.n;               // FULL: the tail helpers take their vector path
    Pk<V> s[T::S];
    {
        const uint8_t *row = a.in + tile_off(g0, a.pitch_in, a.sh_in, T::S);
#pragma unroll
        for (int i = 0; i < T::S; ++i) { s[i] = ld<V, NT>(row + lo); row += a.pitch_in; }
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
        uint8_t *row = a.out + tile_off(g0, a.pitch_out, a.sh_out, T::S);
#pragma unroll
        for (int i = 0; i < T::S; ++i) { st<V, NT>(row + lo, s[i]); row += a.pitch_out; }
    }
    if (a.done != nullptr || a.reward != nullptr) {
        const Pk<V> dn = done_bytes(unsolved<T, V>(s));
        if (a.done) st_tail<V>(a.done, n0, n, dn);
        if (a.reward) store_reward<V>(a.reward, n0, n, dn);
    }
    if constexpr (CODE) {
        Pk<V> c[T::SLOTS];
        encode<T, V>(s, c);
        uint8_t *row = a.code + tile_off(g0, a.code_pitch, a.sh_code, T::SLOTS);
#pragma unroll
        for (int p = 0; p < T::SLOTS; ++p) { st<V, NT>(row + lo, c[p]); row += a.code_pitch; }
    }
}

template <class T, int V, bool MOVE, bool STORE, bool CODE, bool NT, int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_step(StepArgs a) {
    // wave-uniform part of the cube index in SGPRs, 32-bit lane offset in one VGPR
    const int64_t g0 = (int64_t)blockIdx.x * (BLOCK * 4 * V);
    const uint32_t lo = threadIdx.x * (4 * V);
    if (g0 + BLOCK * 4 * V <= a.n) {                           // uniform: the whole workgroup is inside the batch
        step_body<T, V, MOVE, STORE, CODE, NT, true>(a, g0, lo);
    } else if (g0 + lo < a.n) {
        step_body<T, V, MOVE, STORE, CODE, NT, false>(a, g0, lo);
    }
}

// ----------------------------------------------------------- step + dense one-hot (LDS stage)
constexpr int kDenseBlock = 256;
// TILE cubes per workgroup (64 | 256): the first TILE/4 lanes compute the pack's codes, then
// all 256 threads stream the dense rows.  Small tiles keep small batches (MCTS leaves) spread over
// the chip: 4096 cubes are 64 workgroups at TILE = 64 but only 4 at TILE = 1024.
// LDS row pitch TILE + 4 bytes: rows fall on different banks for the byte reads of dense_write.

template <class T, class E, bool MOVE, bool STORE, int TILE>
__global__ void __launch_bounds__(kDenseBlock) k_step_dense(StepArgs a, E *dense) {
    constexpr int TP = TILE + 4;
    __shared__ __attribute__((aligned(16))) uint8_t lds_code[T::SLOTS * TP];
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const uint32_t lo = threadIdx.x * 4;
    const int64_t n0 = tile0 + lo;
    if (lo < TILE && n0 < a.n) {
        Pk<1> s[T::S];
        {
            const uint8_t *row = a.in + tile_off(tile0, a.pitch_in, a.sh_in, T::S);
#pragma unroll
            for (int i = 0; i < T::S; ++i) { s[i] = ld<1, false>(row + lo); row += a.pitch_in; }
        }
        if constexpr (MOVE) {
            const Pk<1> act = ld_tail<1>(a.actions, n0, a.n, 0);
            Pk<1> m[T::A];
            const Pk<1> bad = action_masks<T, 1>(act, m);
            if (any_bad<1>(bad, n0, a.n)) atomicOr(&g_status, RC_STATUS_BAD_ACTION);
            Pk<1> o[T::S];
            apply_move<T, 1>(s, m, o);
#pragma unroll
            for (int i = 0; i < T::S; ++i) s[i] = o[i];
        }
        if constexpr (STORE) {
            uint8_t *row = a.out + tile_off(tile0, a.pitch_out, a.sh_out, T::S);
#pragma unroll
            for (int i = 0; i < T::S; ++i) { st<1, false>(row + lo, s[i]); row += a.pitch_out; }
        }
        if (a.done != nullptr || a.reward != nullptr) {
            const Pk<1> dn = done_bytes(unsolved<T, 1>(s));
            if (a.done) st_tail<1>(a.done, n0, a.n, dn);
            if (a.reward) store_reward<1>(a.reward, n0, a.n, dn);
        }
        Pk<1> c[T::SLOTS];
        encode<T, 1>(s, c);
#pragma unroll
        for (int p = 0; p < T::SLOTS; ++p) *reinterpret_cast<uint32_t *>(lds_code + p * TP + lo) = c[p].d[0];
    }
    __syncthreads();
    const int64_t left = a.n - tile0;
    const int ncubes = left < TILE ? (int)left : TILE;
    dense_write<T, E>(lds_code, TP, dense + tile0 * (T::R * T::C), ncubes, threadIdx.x, kDenseBlock);
}

template <class T, class E, int TILE>
__global__ void __launch_bounds__(kDenseBlock) k_code_to_dense(const uint8_t *code, int64_t n, int64_t code_pitch, int shift, E *dense) {
    constexpr int TP = TILE + 4;
    __shared__ __attribute__((aligned(16))) uint8_t lds_code[T::SLOTS * TP];
    const int64_t tile0 = (int64_t)blockIdx.x * TILE;
    const uint32_t lo = threadIdx.x * 4;
    if (lo < TILE && tile0 + lo < n) {
        const uint8_t *row = code + tile_off(tile0, code_pitch, shift, T::SLOTS) + lo;
#pragma unroll
        for (int p = 0; p < T::SLOTS; ++p) { *reinterpret_cast<uint32_t *>(lds_code + p * TP + lo) = ld<1, false>(row).d[0]; row += code_pitch; }
    }
    __syncthreads();
    const int64_t left = n - tile0;
    const int ncubes = left < TILE ? (int)left : TILE;
    dense_write<T, E>(lds_code, TP, dense + tile0 * (T::R * T::C), ncubes, threadIdx.x, kDenseBlock);
}

// ---------------------------------------------------------------------------- expand
struct ExpandArgs {
    const uint8_t *in;
    int64_t n, pitch_in;
    uint8_t *children, *child_solved, *child_code;
    int64_t pitch_out, tiles_out;
    int parts, sh_in, sh_out;
};

// Row addressing: every row pointer handed to these helpers is WAVE-UNIFORM (kernel argument +
// block-derived offset, lives in SGPRs); the lane adds a 32-bit offset `lo`.  Keeps the 648
// child-row addresses of one expansion out of the vector registers.
__device__ __forceinline__ uint8_t *opaque(uint8_t *p) {
    // defined inside the loop body on purpose: stops the optimiser from turning every row of
    // every child into its own loop-carried address register
    asm volatile("" : "+s"(p));
    return p;
}

// Where one expansion's outputs go.  Child a of the wave's cubes: a tiled state buffer of its own,
// children + a * tiles * S * pitch (layout [A][tile][S][pitch]); codes likewise with SLOTS rows;
// flags are [A][tiles * pitch].  The three pointers already include the wave's own offset
// (tile_off / g0), so only the per-child strides are left.
struct ChildOut {
    uint8_t *children, *child_solved, *child_code;
    int64_t pitch, tiles;
};

template <class T, int V, int A_, bool CODE>
__device__ __forceinline__ void emit_child(const Pk<V> (&s)[T::S], const FamilyCodes<T, V> &fam, const ChildOut &o, uint32_t lo) {
    Pk<V> c[T::S];
    fixed_move<T, V, A_>(s, c);
    if (o.children) {
        uint8_t *row = opaque(o.children + (int64_t)A_ * T::S * o.tiles * o.pitch);
#pragma unroll
        for (int i = 0; i < T::S; ++i) { st<V, RC_OUT_NT>(row + lo, c[i]); row += o.pitch; }
    }
    if (o.child_solved) st<V, RC_OUT_NT>(o.child_solved + (int64_t)A_ * o.tiles * o.pitch + lo, done_bytes(unsolved<T, V>(c)));
    if constexpr (CODE) {
        Pk<V> cc[T::SLOTS];
        family_pick<T, V, A_>(fam, cc);
        uint8_t *row = opaque(o.child_code + (int64_t)A_ * T::SLOTS * o.tiles * o.pitch);
#pragma unroll
        for (int p = 0; p < T::SLOTS; ++p) { st<V, RC_OUT_NT>(row + lo, cc[p]); row += o.pitch; }
    }
}

// children part, part+parts, ... of one parent pack; pointers are wave-uniform; `fam` (the family's shared
// code look-ups) is only read when CODE
template <class T, int V, bool CODE>
__device__ __forceinline__ void emit_children(const Pk<V> (&s)[T::S], const FamilyCodes<T, V> &fam, int part, int parts,
                                              const ChildOut &o, uint32_t lo) {
    sfor<T::A>([&](auto ac) {
        constexpr int a = decltype(ac)::value;
        if ((a - part) % parts == 0 && a >= part) emit_child<T, V, a, CODE>(s, fam, o, lo);
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
        const uint8_t *row = a.in + tile_off(g0, a.pitch_in, a.sh_in, T::S);
#pragma unroll
        for (int i = 0; i < T::S; ++i) { s[i] = ld<V, false>(row + lo); row += a.pitch_in; }
    }
    const ChildOut o{a.children ? a.children + tile_off(g0, a.pitch_out, a.sh_out, T::S) : nullptr,
                     a.child_solved ? a.child_solved + g0 : nullptr,
                     a.child_code ? a.child_code + tile_off(g0, a.pitch_out, a.sh_out, T::SLOTS) : nullptr, a.pitch_out, a.tiles_out};
    FamilyCodes<T, V> fam;
    if constexpr (CODE) family_codes<T, V>(s, fam);
    emit_children<T, V, CODE>(s, fam, part, a.parts, o, lo);
}

// ------------------------------------------------------------------------------- ADI
struct AdiArgs {
    uint64_t seed, stream_id;
    int64_t walk_offset, n_walks, pitch, tiles;   // tiles * pitch = padded walk count of one [.] row
    int depth, parts, shift;
    const uint8_t *actions_in;
    uint8_t *actions_out, *parents, *parent_code, *children, *child_code, *child_solved;
};

// One wave = 256*V walks, kept in registers for all `depth` steps (persistent over depth).
// `parts` waves share a walk group: each recomputes the (cheap) walk and writes its share of
// the children, which is where all the bytes go; part 0 also writes actions and parents.
template <class T, int V, bool CODE>
__global__ void __launch_bounds__(kWave) k_adi(AdiArgs a) {
    const int64_t item = blockIdx.x;
    const int64_t g = item / a.parts;
    const int part = (int)(item - g * a.parts);
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
    for (int d = 0; d < a.depth; ++d) {
        Pk<V> act;
        if (a.actions_in != nullptr) {
            act = ld<V, false>(a.actions_in + (int64_t)d * a.tiles * a.pitch + w0);
        } else {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                uint32_t x = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) x |= rng[4 * k + j].action(T::A) << (8 * j);
                act.d[k] = x;
            }
        }
        Pk<V> m[T::A];
        const Pk<V> bad = action_masks<T, V>(act, m);
        if (any_bad<V>(bad, w0, a.n_walks)) atomicOr(&g_status, RC_STATUS_BAD_ACTION);
        Pk<V> o[T::S];
        apply_move<T, V>(s, m, o);
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = o[i];
        FamilyCodes<T, V> fam;
        if constexpr (CODE) family_codes<T, V>(s, fam);
        if (part == 0) {
            if (a.actions_out) st<V, RC_OUT_NT>(a.actions_out + (int64_t)d * a.tiles * a.pitch + w0, act);
            if (a.parents) {
                uint8_t *row = opaque(a.parents + (int64_t)d * T::S * a.tiles * a.pitch + tile_off(g0, a.pitch, a.shift, T::S));
#pragma unroll
                for (int i = 0; i < T::S; ++i) { st<V, RC_OUT_NT>(row + lo, s[i]); row += a.pitch; }
            }
            if constexpr (CODE) {
                if (a.parent_code) {
                    Pk<V> pc[T::SLOTS];
                    family_pick<T, V, -1>(fam, pc);
                    uint8_t *row = opaque(a.parent_code + (int64_t)d * T::SLOTS * a.tiles * a.pitch + tile_off(g0, a.pitch, a.shift, T::SLOTS));
#pragma unroll
                    for (int p = 0; p < T::SLOTS; ++p) { st<V, RC_OUT_NT>(row + lo, pc[p]); row += a.pitch; }
                }
            }
        }
        const int64_t wp = a.tiles * a.pitch;
        const ChildOut co{a.children ? a.children + (int64_t)d * T::A * T::S * wp + tile_off(g0, a.pitch, a.shift, T::S) : nullptr,
                         a.child_solved ? a.child_solved + (int64_t)d * T::A * wp + g0 : nullptr,
                         a.child_code ? a.child_code + (int64_t)d * T::A * T::SLOTS * wp + tile_off(g0, a.pitch, a.shift, T::SLOTS) : nullptr,
                         a.pitch, a.tiles};
        if (CODE && co.child_code != nullptr) emit_children<T, V, CODE>(s, fam, part, a.parts, co, lo);
        else emit_children<T, V, false>(s, fam, part, a.parts, co, lo);
    }
}

// -------------------------------------------------------------------------- scramble
struct ScrambleArgs {
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
    {
        const uint8_t *row = a.st + tile_off(g0, a.pitch, a.shift, T::S);
#pragma unroll
        for (int i = 0; i < T::S; ++i) { s[i] = ld<V, false>(row + lo); row += a.pitch; }
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
    {
        uint8_t *row = a.st + tile_off(g0, a.pitch, a.shift, T::S);
#pragma unroll
        for (int i = 0; i < T::S; ++i) { st<V, false>(row + lo, s[i]); row += a.pitch; }
    }
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

template <int A_>
__global__ void __launch_bounds__(kWave) k_legacy_actions(const uint32_t *seeds, const int32_t *counts, int count_uniform, int kmax,
                                                          int64_t n, uint8_t *actions_out, int64_t pitch) {
    __shared__ uint32_t mt[kMtN * kWave];
    const int lane = threadIdx.x;
    const int64_t env = (int64_t)blockIdx.x * kWave + lane;
    const bool live = env < n;
    const int want = live ? (counts ? counts[env] : count_uniform) : 0;
    uint32_t x = live ? seeds[env] : 0u;                       // init_genrand
    mt[lane] = x;
    for (int i = 1; i < kMtN; ++i) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)i;
        mt[i * kWave + lane] = x;
    }
    constexpr uint32_t rng = A_ - 1;
    constexpr uint32_t mask = rng | rng >> 1 | rng >> 2 | rng >> 3;   // 15 for 12 actions, 7 for 6
    int idx = kMtN, got = 0;                                   // numpy seeds with mti = 624: the first draw twists
    while (__any(got < want)) {
        if (idx == kMtN) {                                     // genrand's in-place twist of the whole state
            for (int k = 0; k < kMtN; ++k) {
                const int k1 = k + 1 == kMtN ? 0 : k + 1, km = k + kMtM >= kMtN ? k + kMtM - kMtN : k + kMtM;
                const uint32_t y = (mt[k * kWave + lane] & 0x80000000u) | (mt[k1 * kWave + lane] & 0x7fffffffu);
                mt[k * kWave + lane] = mt[km * kWave + lane] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx * kWave + lane];
        ++idx;
        y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
        const uint32_t v = y & mask;
        if (got < want && v <= rng) {                          // masked rejection (numpy _bounded_integers, legacy path)
            actions_out[(int64_t)got * pitch + env] = (uint8_t)v;
            ++got;
        }
    }
    if (live)
        for (int d = want; d < kmax; ++d) actions_out[(int64_t)d * pitch + env] = (uint8_t)A_;   // pad with the no-op
}

// ------------------------------------------------------------------------ ADI targets
template <int A_>
__global__ void __launch_bounds__(256) k_adi_targets(const float *child_value, const uint8_t *child_solved, const float *parent_value,
                                                     const double *weight, int64_t n, int64_t pitch,
                                                     float *target_value, int32_t *target_policy, double *error) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float best = 0.f;
    int arg = -1, solved_at = -1;
#pragma unroll
    for (int k = 0; k < A_; ++k) {
        const float v = child_value[k * pitch + i] + (-1.0f);       // cube_env.py:244  value + reward
        if (child_solved[k * pitch + i] && solved_at < 0) solved_at = k;  // cube_env.py:229-232  first solved child wins
        if (arg < 0 || v > best) { best = v; arg = k; }               // torch.max: first maximal index
    }
    const float tv = solved_at >= 0 ? 1.0f : best;
    target_value[i] = tv;
    target_policy[i] = solved_at >= 0 ? solved_at : arg;
    if (error) error[i] = fabs((double)parent_value[i] - (double)tv) * weight[i];  // cube_env.py:247-251
}

// ------------------------------------------------------------------------- host side
thread_local char t_err[256] = "";
int g_variant = 0;

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
// (pitch a power of two >= 1024).  Returns the shift for tile_off, or -1 if the pitch is bad.
inline int tile_shift(int64_t pitch, int64_t n) {
    if (pitch <= 0 || (pitch & 15) != 0) return -1;
    if (n <= pitch) return 63;
    if (pitch < 1024 || (pitch & (pitch - 1)) != 0) return -1;
    int sh = 0;
    while (((int64_t)1 << sh) < pitch) ++sh;
    return sh;
}
inline hipStream_t S(void *s) { return static_cast<hipStream_t>(s); }

template <class F>
int by_size(int cube_size, F &&f) {
    if (cube_size == 3) return f(Cube3{});
    if (cube_size == 2) return f(Cube2{});
    return fail(RC_EINVAL, "cube_size must be 2 or 3%s");  // NotImplementedError, cube_env.py:44
}

// rc_set_variant(v): v % 10 = pack width (1,2,3 -> V = 1,2,4; 0 = auto), (v / 10) % 10 = row
// traffic policy (1 = non-temporal, 2 = default-cached, 0 = auto).
// Measured on MI355X at 4M cubes (tools/exp/exp_step.hip): V = 2 (8 cubes per lane, dwordx2 rows)
// beats V = 1 and V = 4; non-temporal row traffic wins once the working set no longer fits the
// 256 MiB Infinity Cache and loses when it does (1M cubes).
int pick_v(int64_t n) {
    const int v = g_variant % 10;
    if (v >= 1 && v <= 3) return v == 1 ? 1 : v == 2 ? 2 : 4;
    return n >= (int64_t)1 << 18 ? 2 : 1;
}
bool pick_nt(int64_t touched_bytes) {
    const int p = (g_variant / 10) % 10;
    if (p == 1) return true;
    if (p == 2) return false;
    return touched_bytes > ((int64_t)240 << 20);  // beyond the Infinity Cache: stream past it
}

template <class T, int V, bool MOVE, bool STORE, bool CODE>
int launch_step(const StepArgs &a, hipStream_t st) {
    constexpr int BLOCK = 64;
    const int64_t lanes = (a.n + 4 * V - 1) / (4 * V);
    const int64_t blocks = (lanes + BLOCK - 1) / BLOCK;
    if (blocks > 0x7fffffff) return fail(RC_EINVAL, "too many cubes for one launch%s");
    const bool nt = pick_nt(a.n * T::S * ((STORE && a.out != a.in) ? 2 : 1));
    if (nt) hipLaunchKernelGGL((k_step<T, V, MOVE, STORE, CODE, true, BLOCK>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, a);
    else hipLaunchKernelGGL((k_step<T, V, MOVE, STORE, CODE, false, BLOCK>), dim3((unsigned)blocks), dim3(BLOCK), 0, st, a);
    RC_HIP(hipGetLastError());
    return RC_OK;
}

template <class T, bool MOVE, bool STORE, bool CODE>
int dispatch_step(const StepArgs &a, hipStream_t st) {
    switch (pick_v(a.n)) {
        case 4: return launch_step<T, 4, MOVE, STORE, CODE>(a, st);
        case 2: return launch_step<T, 2, MOVE, STORE, CODE>(a, st);
        default: return launch_step<T, 1, MOVE, STORE, CODE>(a, st);
    }
}

// Measured at 1M cubes (tools/microbench.py densetile): 256-cube tiles 5.4 TB/s (f32) / 6.3 TB/s (u8) against
// 4.9 / 5.4 with 1024-cube tiles -- shorter private write streams per workgroup (DESIGN.md "ADI write ceiling").
inline int dense_tile(int64_t n) {
    const int forced = (g_variant / 100000) % 10;  // rc_set_variant: 100000 / 200000 force 64 / 256
    if (forced) return forced == 1 ? 64 : 256;
    return n >= ((int64_t)1 << 17) ? 256 : 64;
}

template <class T, bool MOVE, bool STORE, int TILE>
int launch_dense_t(const StepArgs &a, void *onehot, int fmt, hipStream_t st) {
    const int64_t blocks = (a.n + TILE - 1) / TILE;
    if (blocks > 0x7fffffff) return fail(RC_EINVAL, "too many cubes for one launch%s");
    const dim3 g((unsigned)blocks), b(kDenseBlock);
    if (fmt == RC_FMT_U8) hipLaunchKernelGGL((k_step_dense<T, uint8_t, MOVE, STORE, TILE>), g, b, 0, st, a, static_cast<uint8_t *>(onehot));
    else if (fmt == RC_FMT_F16) hipLaunchKernelGGL((k_step_dense<T, uint16_t, MOVE, STORE, TILE>), g, b, 0, st, a, static_cast<uint16_t *>(onehot));
    else if (fmt == RC_FMT_BF16) hipLaunchKernelGGL((k_step_dense<T, Bf16, MOVE, STORE, TILE>), g, b, 0, st, a, static_cast<Bf16 *>(onehot));
    else hipLaunchKernelGGL((k_step_dense<T, float, MOVE, STORE, TILE>), g, b, 0, st, a, static_cast<float *>(onehot));
    RC_HIP(hipGetLastError());
    return RC_OK;
}

template <class T, bool MOVE, bool STORE>
int launch_dense(const StepArgs &a, void *onehot, int fmt, hipStream_t st) {
    switch (dense_tile(a.n)) {
        case 256: return launch_dense_t<T, MOVE, STORE, 256>(a, onehot, fmt, st);
        default: return launch_dense_t<T, MOVE, STORE, 64>(a, onehot, fmt, st);
    }
}

template <class T, int TILE>
int launch_code_to_dense(const uint8_t *code, int64_t n, int64_t code_pitch, int sh, void *onehot, int fmt, hipStream_t st) {
    const dim3 g((unsigned)((n + TILE - 1) / TILE)), b(kDenseBlock);
    if (fmt == RC_FMT_U8) hipLaunchKernelGGL((k_code_to_dense<T, uint8_t, TILE>), g, b, 0, st, code, n, code_pitch, sh, static_cast<uint8_t *>(onehot));
    else if (fmt == RC_FMT_F16) hipLaunchKernelGGL((k_code_to_dense<T, uint16_t, TILE>), g, b, 0, st, code, n, code_pitch, sh, static_cast<uint16_t *>(onehot));
    else if (fmt == RC_FMT_BF16) hipLaunchKernelGGL((k_code_to_dense<T, Bf16, TILE>), g, b, 0, st, code, n, code_pitch, sh, static_cast<Bf16 *>(onehot));
    else hipLaunchKernelGGL((k_code_to_dense<T, float, TILE>), g, b, 0, st, code, n, code_pitch, sh, static_cast<float *>(onehot));
    RC_HIP(hipGetLastError());
    return RC_OK;
}

int check_fmt(void *onehot, int fmt, int64_t code_pitch, int64_t n, int *sh_code) {
    if (fmt < RC_FMT_NONE || fmt > RC_FMT_BF16) return fail(RC_EINVAL, "unknown one-hot format%s");
    if ((fmt == RC_FMT_NONE) != (onehot == nullptr)) return fail(RC_EINVAL, "onehot pointer and fmt disagree%s");
    if (onehot && !aligned16(onehot)) return fail(RC_EINVAL, "onehot must be 16-byte aligned%s");
    *sh_code = 63;
    if (fmt == RC_FMT_CODE && (*sh_code = tile_shift(code_pitch, n)) < 0) return fail(RC_EINVAL, "code_pitch: need pitch %% 16 == 0 and pitch >= n_cubes, or a power-of-two tile >= 1024%s");
    return RC_OK;
}

int parts_for(int64_t groups, int A, int64_t want = 2048) {
    // enough wave-items to cover 256 CUs a few times over; parts must divide A (1,2,3,4,6,12 | 1,2,3,6).
    // rc_set_variant: (v / 1000) % 100 forces the value (benchmarks).
    const int forced = (g_variant / 1000) % 100;
    int parts = 1;
    if (forced >= 1 && forced <= A) parts = forced;
    else while (parts < A && groups * parts < want) ++parts;
    while (A % parts) ++parts;
    return parts;
}

}  // namespace

// =============================================================================== C ABI
extern "C" {

int rc_version(void) { return 100; }

const char *rc_last_error(void) { return t_err; }

int rc_set_variant(int variant) {
    g_variant = variant;
    return RC_OK;
}

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
    const int sh = tile_shift(pitch, n);
    if (!stp || !aligned16(stp) || n < 0 || sh < 0) return fail(RC_EINVAL, "rc_fill_solved: bad buffer / pitch%s");
    if (n == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        const int64_t blocks = (n + kWave * 16 - 1) / (kWave * 16);
        hipLaunchKernelGGL((k_fill_solved<T>), dim3((unsigned)blocks), dim3(kWave), 0, S(stream), stp, n, pitch, sh);
        RC_HIP(hipGetLastError());
        return RC_OK;
    });
}

static int step_common(const uint8_t *in, uint8_t *out, const uint8_t *actions, int64_t n, int64_t pitch_in, int64_t pitch_out,
                       int cube_size, float *reward, uint8_t *done, void *onehot, int fmt, int64_t code_pitch, void *stream,
                       bool move, bool store) {
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
            if (move) return launch_dense<T, true, true>(a, onehot, fmt, st);
            return launch_dense<T, false, false>(a, onehot, fmt, st);
        }
        if (move) {
            if (fmt == RC_FMT_CODE) return dispatch_step<T, true, true, true>(a, st);
            return dispatch_step<T, true, true, false>(a, st);
        }
        if (fmt == RC_FMT_CODE) return dispatch_step<T, false, false, true>(a, st);
        return dispatch_step<T, false, false, false>(a, st);
    });
}

int rc_apply_moves(const uint8_t *in, uint8_t *out, const uint8_t *actions, int64_t n, int64_t pitch_in, int64_t pitch_out,
                   int cube_size, float *reward, uint8_t *done, void *onehot, int fmt, int64_t code_pitch, void *stream) {
    return step_common(in, out, actions, n, pitch_in, pitch_out, cube_size, reward, done, onehot, fmt, code_pitch, stream, true, true);
}

int rc_scramble(uint8_t *stp, int64_t n, int64_t pitch, int cube_size, int depth, uint64_t seed, uint64_t stream_id,
                int64_t walk_offset, const uint8_t *actions_in, uint8_t *actions_out, int64_t act_pitch, uint8_t *done,
                float *reward, void *stream) {
    const int sh = tile_shift(pitch, n);
    if (!stp || !aligned16(stp) || n < 0 || depth < 0 || sh < 0) return fail(RC_EINVAL, "rc_scramble: bad state buffer / pitch%s");
    if ((actions_in || actions_out) && bad_pitch(act_pitch, n)) return fail(RC_EINVAL, "rc_scramble: bad act_pitch%s");
    if ((actions_in && !aligned16(actions_in)) || (actions_out && !aligned16(actions_out))) return fail(RC_EINVAL, "rc_scramble: action buffers must be 16-byte aligned%s");
    if (reward && (reinterpret_cast<uintptr_t>(reward) & 15u)) return fail(RC_EINVAL, "reward must be 16-byte aligned%s");
    if (n == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        ScrambleArgs a{stp, n, pitch, depth, sh, seed, stream_id, walk_offset, actions_in, actions_out, act_pitch, done, reward};
        const dim3 g((unsigned)((n + kWave * 4 - 1) / (kWave * 4))), b(kWave);
        hipLaunchKernelGGL((k_scramble<T>), g, b, 0, S(stream), a);
        RC_HIP(hipGetLastError());
        return RC_OK;
    });
}

int rc_legacy_scramble_actions(const uint32_t *seeds, const int32_t *counts, int count_uniform, int kmax, int64_t n, int cube_size,
                                uint8_t *actions_out, int64_t pitch, void *stream) {
    if (!seeds || !actions_out || n < 0 || kmax < 0 || bad_pitch(pitch, n)) return fail(RC_EINVAL, "rc_legacy_scramble_actions: bad arguments%s");
    if (!counts && (count_uniform < 0 || count_uniform > kmax)) return fail(RC_EINVAL, "rc_legacy_scramble_actions: count_uniform must be in 0..kmax%s");
    if (n == 0 || kmax == 0) return RC_OK;
    const dim3 g((unsigned)((n + kWave - 1) / kWave)), b(kWave);
    if (cube_size == 3) hipLaunchKernelGGL((k_legacy_actions<12>), g, b, 0, S(stream), seeds, counts, count_uniform, kmax, n, actions_out, pitch);
    else if (cube_size == 2) hipLaunchKernelGGL((k_legacy_actions<6>), g, b, 0, S(stream), seeds, counts, count_uniform, kmax, n, actions_out, pitch);
    else return fail(RC_EINVAL, "cube_size must be 2 or 3%s");
    RC_HIP(hipGetLastError());
    return RC_OK;
}

int rc_is_solved(const uint8_t *stp, int64_t n, int64_t pitch, int cube_size, uint8_t *done, float *reward, void *stream) {
    if (!done && !reward) return fail(RC_EINVAL, "rc_is_solved: nothing to write%s");
    return step_common(stp, nullptr, nullptr, n, pitch, 0, cube_size, reward, done, nullptr, RC_FMT_NONE, 0, stream, false, false);
}

int rc_encode(const uint8_t *stp, int64_t n, int64_t pitch, int cube_size, void *onehot, int fmt, int64_t code_pitch, void *stream) {
    if (fmt == RC_FMT_NONE) return fail(RC_EINVAL, "rc_encode: fmt must not be RC_FMT_NONE%s");
    return step_common(stp, nullptr, nullptr, n, pitch, 0, cube_size, nullptr, nullptr, onehot, fmt, code_pitch, stream, false, false);
}

int rc_onehot_from_code(const uint8_t *code, int64_t n, int64_t code_pitch, int cube_size, void *onehot, int fmt, void *stream) {
    const int sh = tile_shift(code_pitch, n);
    if (!code || !aligned16(code) || n < 0 || sh < 0) return fail(RC_EINVAL, "bad code buffer / pitch%s");
    if (fmt < RC_FMT_U8 || fmt > RC_FMT_BF16 || !onehot || !aligned16(onehot)) return fail(RC_EINVAL, "rc_onehot_from_code: dense fmt and aligned buffer required%s");
    if (n == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        switch (dense_tile(n)) {
            case 256: return launch_code_to_dense<T, 256>(code, n, code_pitch, sh, onehot, fmt, S(stream));
            default: return launch_code_to_dense<T, 64>(code, n, code_pitch, sh, onehot, fmt, S(stream));
        }
    });
}

int rc_expand_children(const uint8_t *in, int64_t n, int64_t pitch_in, int cube_size, uint8_t *children, uint8_t *child_solved,
                       uint8_t *child_code, int64_t pitch_out, void *stream) {
    const int sh_in = tile_shift(pitch_in, n), sh_out = tile_shift(pitch_out, n);
    if (!in || !aligned16(in) || n < 0 || sh_in < 0 || sh_out < 0) return fail(RC_EINVAL, "rc_expand_children: bad buffer / pitch%s");
    if (!children && !child_solved && !child_code) return fail(RC_EINVAL, "rc_expand_children: nothing to write%s");
    if ((children && !aligned16(children)) || (child_solved && !aligned16(child_solved)) || (child_code && !aligned16(child_code)))
        return fail(RC_EINVAL, "rc_expand_children: outputs must be 16-byte aligned%s");
    if (n == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        const int V = n >= ((int64_t)1 << 20) ? 2 : 1;
        const int64_t groups = (n + kWave * 4 * V - 1) / (kWave * 4 * V);
        ExpandArgs a{in, n, pitch_in, children, child_solved, child_code, pitch_out, n <= pitch_out ? 1 : (n + pitch_out - 1) / pitch_out,
                     parts_for(groups, T::A), sh_in, sh_out};
        const dim3 g((unsigned)(groups * a.parts)), b(kWave);
        hipStream_t st = S(stream);
        if (V == 2) {
            if (child_code) hipLaunchKernelGGL((k_expand<T, 2, true>), g, b, 0, st, a);
            else hipLaunchKernelGGL((k_expand<T, 2, false>), g, b, 0, st, a);
        } else {
            if (child_code) hipLaunchKernelGGL((k_expand<T, 1, true>), g, b, 0, st, a);
            else hipLaunchKernelGGL((k_expand<T, 1, false>), g, b, 0, st, a);
        }
        RC_HIP(hipGetLastError());
        return RC_OK;
    });
}

int rc_adi_generate(uint64_t seed, uint64_t stream_id, int64_t walk_offset, int64_t n_walks, int depth, int cube_size, int64_t pitch,
                    const uint8_t *actions_in, uint8_t *actions_out, uint8_t *parents, uint8_t *parent_code, uint8_t *children,
                    uint8_t *child_code, uint8_t *child_solved, void *stream) {
    const int sh = tile_shift(pitch, n_walks);
    if (n_walks < 0 || depth < 0 || sh < 0) return fail(RC_EINVAL, "rc_adi_generate: bad sizes / pitch%s");
    const void *ptrs[] = {actions_in, actions_out, parents, parent_code, children, child_code, child_solved};
    for (const void *p : ptrs)
        if (p && !aligned16(p)) return fail(RC_EINVAL, "rc_adi_generate: buffers must be 16-byte aligned%s");
    if (n_walks == 0 || depth == 0) return RC_OK;
    return by_size(cube_size, [&](auto t) {
        using T = decltype(t);
        const int V = 1;
        const int64_t groups = (n_walks + kWave * 4 * V - 1) / (kWave * 4 * V);
        const bool any_child = children || child_code || child_solved;
        // sticker children are byte-bound: split them over many waves; code-only expansion is VALU-bound
        // (every part re-encodes the parent), measured best at 2 parts for 100k walks
        AdiArgs a{seed, stream_id, walk_offset, n_walks, pitch, n_walks <= pitch ? 1 : (n_walks + pitch - 1) / pitch, depth,
                  any_child ? parts_for(groups, T::A, children ? 2048 : 700) : 1, sh, actions_in, actions_out, parents, parent_code, children, child_code, child_solved};
        const dim3 g((unsigned)(groups * a.parts)), b(kWave);
        if (parent_code || child_code) hipLaunchKernelGGL((k_adi<T, 1, true>), g, b, 0, S(stream), a);
        else hipLaunchKernelGGL((k_adi<T, 1, false>), g, b, 0, S(stream), a);
        RC_HIP(hipGetLastError());
        return RC_OK;
    });
}

int rc_adi_targets(const float *child_value, const uint8_t *child_solved, const float *parent_value, const double *weight, int64_t n,
                   int64_t pitch, int cube_size, float *target_value, int32_t *target_policy, double *error, void *stream) {
    if (!child_value || !child_solved || !target_value || !target_policy || n < 0 || pitch < n) return fail(RC_EINVAL, "rc_adi_targets: bad arguments%s");
    if (error && (!parent_value || !weight)) return fail(RC_EINVAL, "rc_adi_targets: error needs parent_value and weight%s");
    if (n == 0) return RC_OK;
    const dim3 g((unsigned)((n + 255) / 256)), b(256);
    if (cube_size == 3) hipLaunchKernelGGL((k_adi_targets<12>), g, b, 0, S(stream), child_value, child_solved, parent_value, weight, n, pitch, target_value, target_policy, error);
    else if (cube_size == 2) hipLaunchKernelGGL((k_adi_targets<6>), g, b, 0, S(stream), child_value, child_solved, parent_value, weight, n, pitch, target_value, target_policy, error);
    else return fail(RC_EINVAL, "cube_size must be 2 or 3%s");
    RC_HIP(hipGetLastError());
    return RC_OK;
}

int rc_read_status(uint32_t *status, void *stream) {
    if (!status) return fail(RC_EINVAL, "status is NULL%s");
    uint32_t zero = 0;
    RC_HIP(hipStreamSynchronize(S(stream)));
    RC_HIP(hipMemcpyFromSymbol(status, HIP_SYMBOL(g_status), sizeof *status));
    if (*status) RC_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_status), &zero, sizeof zero));
    return RC_OK;
}

}  // extern "C"
