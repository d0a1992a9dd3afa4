// rc_device.h -- device-side building blocks of the cube hot path (gfx950 / CDNA4 only).
//
// Everything works on PACKED bytes: one 32-bit register holds the same sticker (or action,
// or code) of 4 consecutive cubes, a lane carries V such registers per row (4*V cubes), so a
// wavefront moves 256*V cubes at once and every global access is a coalesced row segment
// of the structure-of-arrays state.  A face turn is then a network of v_bfi_b32 selects
// driven by per-byte action masks (one v_perm_b32 each); no LDS, no per-byte gathers.
//
// Reference semantics restated here (paths relative to /root/reference/gym-cube/gym_cube/envs):
//   apply_move  : doMove_3,  assets/py333.py:220-222   new[i] = old[moveDefs[a][i]]
//   unsolved    : isSolved_3, assets/py333.py:229-233  every face equals its first sticker
//   encode      : getOP_3 + pos_to_state_3, assets/py333.py:224-246 (hash, LUT, column)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>

#include "rc_tables.h"

namespace rc {

// ---------------------------------------------------------------- compile-time loops
template <class F, int... I>
__device__ __forceinline__ void sfor_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void sfor(F &&f) {
    sfor_impl(f, std::make_integer_sequence<int, N>{});
}

// perm[a][i] expanded at compile time from the 4-cycles of rc_tables.h
template <class T>
struct PermTable {
    uint8_t v[T::A][T::S];
    constexpr PermTable() : v{} {
        for (int a = 0; a < T::A; ++a)
            for (int i = 0; i < T::S; ++i) v[a][i] = (uint8_t)i;
        for (int f = 0; f < T::A / 2; ++f)
            for (int c = 0; c < T::NCYC; ++c)
                for (int k = 0; k < 4; ++k) {
                    const uint8_t src = T::turn[f][c][k], dst = T::turn[f][c][(k + 1) & 3];
                    v[2 * f][dst] = src;      // clockwise: content of src arrives at dst
                    v[2 * f + 1][src] = dst;  // counter-clockwise: the inverse
                }
    }
};
template <class T>
inline constexpr PermTable<T> kPerm{};

// ------------------------------------------------------------------ packed registers
template <int V>
struct Pk {
    uint32_t d[V];
};

#define RC_V _Pragma("unroll") for (int k = 0; k < V; ++k)

template <int V> __device__ __forceinline__ Pk<V> splat(uint32_t x) { Pk<V> r; RC_V r.d[k] = x; return r; }
template <int V> __device__ __forceinline__ Pk<V> operator^(Pk<V> a, Pk<V> b) { Pk<V> r; RC_V r.d[k] = a.d[k] ^ b.d[k]; return r; }
template <int V> __device__ __forceinline__ Pk<V> operator|(Pk<V> a, Pk<V> b) { Pk<V> r; RC_V r.d[k] = a.d[k] | b.d[k]; return r; }
template <int V> __device__ __forceinline__ Pk<V> operator&(Pk<V> a, uint32_t c) { Pk<V> r; RC_V r.d[k] = a.d[k] & c; return r; }
template <int V> __device__ __forceinline__ Pk<V> shl(Pk<V> a, int s) { Pk<V> r; RC_V r.d[k] = a.d[k] << s; return r; }
// (a << s) + b  -> v_lshl_add_u32
template <int V> __device__ __forceinline__ Pk<V> shl_add(Pk<V> a, int s, Pk<V> b) { Pk<V> r; RC_V r.d[k] = (a.d[k] << s) + b.d[k]; return r; }
// per bit: m ? a : b  -> v_bfi_b32
template <int V> __device__ __forceinline__ Pk<V> sel(Pk<V> m, Pk<V> a, Pk<V> b) { Pk<V> r; RC_V r.d[k] = (a.d[k] & m.d[k]) | (b.d[k] & ~m.d[k]); return r; }
template <int V> __device__ __forceinline__ Pk<V> andn(Pk<V> a, Pk<V> m) { Pk<V> r; RC_V r.d[k] = a.d[k] & ~m.d[k]; return r; }
// v_perm_b32: byte j of the result = bytes {hi:lo}[sel.byte j] for sel 0..7, 0x00 for 12, 0xff for >= 13,
// 0xff * sign of byte 1/3/5/7 for sel 8/9/10/11
template <int V> __device__ __forceinline__ Pk<V> perm(uint32_t hi, uint32_t lo, Pk<V> s) { Pk<V> r; RC_V r.d[k] = __builtin_amdgcn_perm(hi, lo, s.d[k]); return r; }
// 0xff in every byte whose bit 7 is set
template <int V> __device__ __forceinline__ Pk<V> signmask(Pk<V> x) { Pk<V> r; RC_V r.d[k] = __builtin_amdgcn_perm(x.d[k] << 8, x.d[k], 0x090B080Au); return r; }
template <int V> __device__ __forceinline__ bool any(Pk<V> a) { uint32_t o = 0; RC_V o |= a.d[k]; return o != 0; }

// ------------------------------------------------------------------ global row access
template <int V> struct VecT;
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <> struct VecT<1> { using type = uint32_t; };
template <> struct VecT<2> { using type = u32x2; };
template <> struct VecT<4> { using type = u32x4; };

template <int V, bool NT>
__device__ __forceinline__ Pk<V> ld(const uint8_t *p) {
    using U = typename VecT<V>::type;
    U u;
    if constexpr (NT) u = __builtin_nontemporal_load(reinterpret_cast<const U *>(p));
    else u = *reinterpret_cast<const U *>(p);
    Pk<V> r;
    __builtin_memcpy(&r, &u, sizeof(U));
    return r;
}
template <int V, bool NT>
__device__ __forceinline__ void st(uint8_t *p, Pk<V> v) {
    using U = typename VecT<V>::type;
    U u;
    __builtin_memcpy(&u, &v, sizeof(U));
    if constexpr (NT) __builtin_nontemporal_store(u, reinterpret_cast<U *>(p));
    else *reinterpret_cast<U *>(p) = u;
}
// [n] arrays without padding (actions, done): vector access when the whole pack is inside, bytes otherwise.
// The byte paths are fully unrolled with static register indices (a run-time index would push the pack into
// scratch / LDS and slow the common path down as well).
template <int V>
__device__ __forceinline__ Pk<V> ld_tail(const uint8_t *base, int64_t n0, int64_t n, uint32_t fill) {
    if (n0 + 4 * V <= n) return ld<V, false>(base + n0);
    Pk<V> r = splat<V>(fill * 0x01010101u);
#pragma unroll
    for (int j = 0; j < 4 * V; ++j)
        if (n0 + j < n) r.d[j >> 2] = (r.d[j >> 2] & ~(0xffu << (8 * (j & 3)))) | ((uint32_t)base[n0 + j] << (8 * (j & 3)));
    return r;
}
template <int V>
__device__ __forceinline__ void st_tail(uint8_t *base, int64_t n0, int64_t n, Pk<V> v) {
    if (n0 + 4 * V <= n) { st<V, false>(base + n0, v); return; }
#pragma unroll
    for (int j = 0; j < 4 * V; ++j)
        if (n0 + j < n) base[n0 + j] = (uint8_t)(v.d[j >> 2] >> (8 * (j & 3)));
}

// ------------------------------------------------------------- buffer (SRD) row access
// Row traffic goes through raw BUFFER instructions: a 4-SGPR resource descriptor built from a
// WAVE-UNIFORM base pointer, the lane's 32-bit byte offset in one VGPR (voffset) and the row offset in an
// SGPR (soffset).  No per-access 64-bit VALU address arithmetic, no address VGPR pairs, and the cache
// policy is an immediate of the instruction (aux bits on gfx950: 1 = sc0, 2 = nt, 16 = sc1).
// Measured on MI355X (tools/exp/exp_write3.hip, exp_step2.hip; profiles/r02_design_ab.json):
//   write-once output streams  : sc0 sc1 nt (19) -- the ADI child stream 7.2 TB/s against 5.3 default-cached
//   inputs read once           : nt (2)
//   outputs the next launch reads (state ping-pong up to the Infinity Cache size): sc0 sc1 (17)
constexpr int kAuxCached = 0, kAuxStreamLoad = 2, kAuxKeepStore = 17, kAuxStreamStore = 19;

// soffset is 32 bits: callers keep (rows * pitch) below 2^32 (checked on the host).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void *wave_uniform_base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(wave_uniform_base), 0, 0xffffffff, 0x00020000);
}
template <int V, int AUX>
__device__ __forceinline__ Pk<V> bld(__amdgpu_buffer_rsrc_t r, uint32_t lo, uint32_t so) {
    Pk<V> p;
    if constexpr (V == 1) p.d[0] = __builtin_amdgcn_raw_buffer_load_b32(r, lo, so, AUX);
    else if constexpr (V == 2) { const u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, lo, so, AUX); p.d[0] = u[0]; p.d[1] = u[1]; }
    else { const u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, lo, so, AUX); p.d[0] = u[0]; p.d[1] = u[1]; p.d[2] = u[2]; p.d[3] = u[3]; }
    return p;
}
template <int V, int AUX>
__device__ __forceinline__ void bst(__amdgpu_buffer_rsrc_t r, uint32_t lo, uint32_t so, Pk<V> p) {
    if constexpr (V == 1) __builtin_amdgcn_raw_buffer_store_b32(p.d[0], r, lo, so, AUX);
    else if constexpr (V == 2) { const u32x2 u = {p.d[0], p.d[1]}; __builtin_amdgcn_raw_buffer_store_b64(u, r, lo, so, AUX); }
    else { const u32x4 u = {p.d[0], p.d[1], p.d[2], p.d[3]}; __builtin_amdgcn_raw_buffer_store_b128(u, r, lo, so, AUX); }
}

// ---------------------------------------------------------------------- action masks
// m[a] selects (per byte) the cubes whose action is a.  Only the bits a sticker can use
// (0..2) are guaranteed: mask bytes are 0x07 (a < 8) or 0xff (a >= 8); a stray 0x80 can show
// for a >= 8 on cubes with an odd action < 8 -- stickers never carry bit 7, so it selects 0 from 0.
// The action value A (12 | 6) is the NO-OP: every mask reads 0x00 for it, the cube stays put.
// Returns, per byte, non-zero where the action is neither 0..A-1 nor the no-op.
constexpr uint64_t mask_data(int a) {
    // a < 8: selector a reads byte a = 0x07; a >= 8: selector a reads the sign of byte 2*(a-8)+1 = 0x80
    return a < 8 ? (0x07ull << (8 * (a & 7))) : (0x80ull << (8 * (2 * ((a - 8) & 3) + 1)));
}
template <class T, int V>
__device__ __forceinline__ Pk<V> action_masks(Pk<V> act, Pk<V> (&m)[T::A]) {
    Pk<V> seen = splat<V>(0);
    sfor<T::A>([&](auto ac) {
        constexpr int a = decltype(ac)::value;
        constexpr uint64_t data = mask_data(a);
        m[a] = perm<V>((uint32_t)(data >> 32), (uint32_t)data, act);
        seen = seen | m[a];
    });
    Pk<V> big = perm<V>(0u, 0u, act);  // 0xff where the byte is >= 13: every mask above read 0xff
    Pk<V> bad;
    RC_V {
        const uint32_t x = act.d[k] ^ ((uint32_t)T::A * 0x01010101u);                       // 0 where action == A
        const uint32_t not_noop = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) >> 7 & 0x01010101u;  // 1 where action != A
        bad.d[k] = ((~seen.d[k] & 0x07070707u) | big.d[k]) & (not_noop * 0xffu);
    }
    return bad;
}

// any bad action among the pack's cubes that really exist (cube index < n)?  Pad columns may hold anything.
template <int V>
__device__ __forceinline__ bool any_bad(Pk<V> bad, int64_t n0, int64_t n) {
    const int64_t r = n - n0;                                  // cubes of this pack inside the batch (> 0)
    if (r >= 4 * V) return any(bad);
    uint32_t o = 0;
    RC_V {
        const int rk = (int)r - 4 * k;                         // valid bytes of dword k
        const uint32_t mk = rk >= 4 ? 0xffffffffu : rk <= 0 ? 0u : ((1u << (8 * rk)) - 1u);
        o |= bad.d[k] & mk;
    }
    return o != 0;
}

// ------------------------------------------------------------------------- the move
template <class T, int V>
__device__ __forceinline__ void apply_move(const Pk<V> (&in)[T::S], const Pk<V> (&m)[T::A], Pk<V> (&out)[T::S]) {
    sfor<T::S>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        Pk<V> o = in[i];
        sfor<T::A>([&](auto ac) {
            constexpr int a = decltype(ac)::value;
            constexpr int src = kPerm<T>.v[a][i];
            if constexpr (src != i) o = sel(m[a], in[src], o);
        });
        out[i] = o;
    });
}

// The same move IN PLACE, orbit by orbit: corner stickers only ever receive corner stickers and edge stickers edge stickers
// (centres stay), so the new values of one orbit (24 registers per pack) are built and written back before the other orbit is
// touched -- peak live state 54 + 24 instead of 54 + 54 registers per pack.  Used where the register budget is tight (the
// 960-thread dense kernels: 128 VGPRs per lane).
template <class T, int I>
constexpr bool in_orbit(int orbit) {   // orbit 0: corner stickers, 1: edge stickers (3x3x3); the 2x2x2 has corners only
    if (T::SIZE == 2) return orbit == 0;
    const int k = I % 9;
    return orbit == 0 ? (k == 0 || k == 2 || k == 6 || k == 8) : (k == 1 || k == 3 || k == 5 || k == 7);
}
template <class T, int V>
__device__ __forceinline__ void apply_move_inplace(Pk<V> (&s)[T::S], const Pk<V> (&m)[T::A]) {
    sfor<2>([&](auto oc) {
        constexpr int orbit = decltype(oc)::value;
        Pk<V> t[T::S];                                             // only this orbit's entries are ever touched
        sfor<T::S>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (in_orbit<T, i>(orbit)) {
                Pk<V> o = s[i];
                sfor<T::A>([&](auto ac) {
                    constexpr int a = decltype(ac)::value;
                    constexpr int src = kPerm<T>.v[a][i];
                    if constexpr (src != i) o = sel(m[a], s[src], o);
                });
                t[i] = o;
            }
        });
        sfor<T::S>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if constexpr (in_orbit<T, i>(orbit)) s[i] = t[i];
        });
    });
}

// child by a FIXED action: pure register renaming
template <class T, int V, int A_>
__device__ __forceinline__ void fixed_move(const Pk<V> (&in)[T::S], Pk<V> (&out)[T::S]) {
    sfor<T::S>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        constexpr int src = kPerm<T>.v[A_][i];
        out[i] = in[src];
    });
}

// per byte: non-zero iff the cube is NOT solved (py333.py:229-233: compare with the face's first sticker)
template <class T, int V>
__device__ __forceinline__ Pk<V> unsolved(const Pk<V> (&s)[T::S]) {
    Pk<V> acc = splat<V>(0);
    sfor<6>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        sfor<T::FACE - 1>([&](auto kc) {
            constexpr int k = decltype(kc)::value + 1;
            acc = acc | (s[f * T::FACE + k] ^ s[f * T::FACE]);
        });
    });
    return acc;
}
// 0x01 per solved cube, 0x00 otherwise
template <int V>
__device__ __forceinline__ Pk<V> done_bytes(Pk<V> uns) {
    Pk<V> r;
    RC_V {
        const uint32_t x = uns.d[k];
        const uint32_t nz7 = (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;  // bit 7 <=> byte != 0
        r.d[k] = (nz7 >> 7) ^ 0x01010101u;
    }
    return r;
}
// reward of cube j of a pack: +1.0f if done else -1.0f  (cube_env.py:99-104)
__device__ __forceinline__ float reward_of(uint32_t done_dword, int byte) {
    return __uint_as_float(0xBF800000u ^ (((done_dword >> (8 * byte)) & 1u) << 31));
}

// ---- solved flags of all A children of a REACHABLE parent (ADI walks start from the solved cube) ------
// child a = move_a(parent) is solved  <=>  parent == K_a := move_a^-1(solved), a constant state that differs
// from the solved cube only on the ring of face a/2 (the 4 strips around the turned face).  On reachable
// states "every face equals its first sticker" (py333.py:229-233) is the same as "equals the solved cube":
// 3x3x3 centres never move, the 2x2x2 DLB cubie is fixed.  So
//   unsolved(child a) = OR_{j not in ring} (s[j] ^ solved[j])  |  OR_{j in ring} (s[j] ^ K_a[j])
// and the first term is shared between children: stickers are grouped by the set of rings they belong to
// (their signature), one OR per class, one OR of classes per face.  ~370 VALU ops per pack for the 12 flags
// of a 3x3x3 parent instead of 12 x 96.  rc_expand_children (arbitrary colourings allowed) keeps the
// literal per-child test.
template <class T>
struct RingInfo {
    uint8_t K[T::A][T::S];
    uint8_t sig[T::S];      // bit f set: sticker j lies in the ring of face f
    constexpr RingInfo() : K{}, sig{} {
        for (int a = 0; a < T::A; ++a)
            for (int i = 0; i < T::S; ++i) K[a][kPerm<T>.v[a][i]] = (uint8_t)(i / T::FACE);   // child[i] = parent[perm[a][i]] must be solved[i]
        for (int j = 0; j < T::S; ++j)
            for (int a = 0; a < T::A; ++a)
                if (K[a][j] != j / T::FACE) sig[j] |= (uint8_t)(1u << (a / 2));
    }
    constexpr bool sig_used(int c) const {
        for (int j = 0; j < T::S; ++j)
            if (sig[j] == c) return true;
        return false;
    }
};
template <class T>
inline constexpr RingInfo<T> kRing{};

template <class T, int V>
struct ChildFlags {
    Pk<V> not_ring[T::A / 2];   // per face: OR of (s ^ solved) over the stickers outside its ring
};
template <class T, int V>
__device__ __forceinline__ void child_flags_prepare(const Pk<V> (&s)[T::S], ChildFlags<T, V> &cf) {
    constexpr int NSIG = 1 << (T::A / 2);
    Pk<V> cls[NSIG];
    sfor<NSIG>([&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (kRing<T>.sig_used(c)) {
            Pk<V> acc = splat<V>(0);
            sfor<T::S>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                if constexpr (kRing<T>.sig[j] == c) acc = acc | (s[j] ^ splat<V>((uint32_t)(j / T::FACE) * 0x01010101u));
            });
            cls[c] = acc;
        }
    });
    sfor<T::A / 2>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        Pk<V> acc = splat<V>(0);
        sfor<NSIG>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (kRing<T>.sig_used(c) && !((c >> f) & 1)) acc = acc | cls[c];
        });
        cf.not_ring[f] = acc;
    });
}
// per byte: non-zero iff child A_ of the (reachable) parent s is NOT solved
template <class T, int V, int A_>
__device__ __forceinline__ Pk<V> child_unsolved(const Pk<V> (&s)[T::S], const ChildFlags<T, V> &cf) {
    Pk<V> acc = cf.not_ring[A_ / 2];
    sfor<T::S>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        if constexpr ((kRing<T>.sig[j] >> (A_ / 2)) & 1) acc = acc | (s[j] ^ splat<V>((uint32_t)kRing<T>.K[A_][j] * 0x01010101u));
    });
    return acc;
}

// ------------------------------------------------------------------------- one-hot code
// 72-entry byte LUT in 18 literal dwords, index = packed byte h (< 72): 8-entry v_perm per group,
// then a select tree on bits 3..5 (and bit 6 when the hash can reach 64).
template <class T, int V, bool CORNER>
__device__ __forceinline__ Pk<V> lut72(Pk<V> h) {
    constexpr bool BIT6 = CORNER;  // only the corner hash (max 65) can reach 64
    const Pk<V> lo3 = h & 0x07070707u;
    Pk<V> g[8];
    sfor<8>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        constexpr uint32_t hi = CORNER ? T::ccode_dw[2 * j + 1] : T::ecode_dw[2 * j + 1];
        constexpr uint32_t lo = CORNER ? T::ccode_dw[2 * j] : T::ecode_dw[2 * j];
        if constexpr (hi == 0 && lo == 0) g[j] = splat<V>(0);
        else g[j] = perm<V>(hi, lo, lo3);
    });
    const Pk<V> m3 = signmask(shl(h, 4)), m4 = signmask(shl(h, 3)), m5 = signmask(shl(h, 2));
    const Pk<V> a0 = sel(m3, g[1], g[0]), a1 = sel(m3, g[3], g[2]), a2 = sel(m3, g[5], g[4]), a3 = sel(m3, g[7], g[6]);
    const Pk<V> b0 = sel(m4, a1, a0), b1 = sel(m4, a3, a2);
    Pk<V> r = sel(m5, b1, b0);
    if constexpr (BIT6) {
        constexpr uint32_t hi = CORNER ? T::ccode_dw[17] : T::ecode_dw[17];
        constexpr uint32_t lo = CORNER ? T::ccode_dw[16] : T::ecode_dw[16];
        const Pk<V> m6 = signmask(shl(h, 1));
        if constexpr (hi == 0 && lo == 0) r = andn(r, m6);
        else r = sel(m6, perm<V>(hi, lo, lo3), r);
    }
    return r;
}

// Code look-up indexed by TWO colours (rc_tables.h epair_dw / cpair_dw, tables.py pair_tables): row = second colour c1,
// byte inside the row = first colour c0.  Colours are 0..5, so c0 is a v_perm selector as it stands (no hash, no AND), the
// six rows are six 8-entry v_perm look-ups, and the row is chosen by a select tree on the bits of c1 whose three masks are
// v_perm look-ups of c1 as well: 6 + 3 v_perm + 5 v_bfi instead of hash + lut72's ~25-32 operations.
//   TABLE < 0 : edges, epair[c1][c0] = ecode[c0 + 10 c1] -- the hash is injective in (c0, c1): exact for ANY colouring;
//   TABLE >= 0: corners of states REACHABLE from solved, where (c0, c1) of a slot read in a fixed order determines the third
//               colour: cpair[TABLE][c1][c0] = ccode[c0 + 2 c1 + 10 c2(c0, c1)], zeros where the reference's table is silent.
template <class T, int V, int TABLE>
__device__ __forceinline__ Pk<V> lut_pair(Pk<V> c0, Pk<V> c1) {
    Pk<V> g[6];
    sfor<6>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        constexpr uint32_t lo = TABLE < 0 ? T::epair_dw[k][0] : T::cpair_dw[TABLE < 0 ? 0 : TABLE][k][0];
        constexpr uint32_t hi = TABLE < 0 ? T::epair_dw[k][1] : T::cpair_dw[TABLE < 0 ? 0 : TABLE][k][1];
        if constexpr (hi == 0 && lo == 0) g[k] = splat<V>(0);
        else g[k] = perm<V>(hi, lo, c0);
    });
    const Pk<V> b0 = perm<V>(0xFF00FF00u, 0xFF00FF00u, c1);    // 0xff where bit 0 of c1 is set
    const Pk<V> b1 = perm<V>(0xFFFF0000u, 0xFFFF0000u, c1);    //                bit 1
    const Pk<V> b2 = perm<V>(0xFFFFFFFFu, 0x00000000u, c1);    //                bit 2
    const Pk<V> a0 = sel(b0, g[1], g[0]), a1 = sel(b0, g[3], g[2]), a2 = sel(b0, g[5], g[4]);
    return sel(b2, a2, sel(b1, a1, a0));
}

template <class T, int V, int SLOT>
__device__ __forceinline__ Pk<V> encode_slot(const Pk<V> (&s)[T::S]) {
    if constexpr (SLOT < T::NC) {
        // h = c0 + 2*c1 + 10*c2   (py333.py:167,225); max 65 -> bit 6 can be set
        constexpr int i0 = T::cdef[SLOT][0], i1 = T::cdef[SLOT][1], i2 = T::cdef[SLOT][2];
        const Pk<V> c0 = s[i0], c1 = s[i1], c2 = s[i2];
        const Pk<V> h = shl_add(c2, 1, shl_add(c2, 3, shl_add(c1, 1, c0)));
        return lut72<T, V, true>(h);
    } else {
        // edge_pieceInds[c0 + 10*c1] (py333.py:168,226) read as a table of the two colours: exact for any colouring
        constexpr int e = SLOT - T::NC;
        constexpr int i0 = T::edef[e][0], i1 = T::edef[e][1];
        return lut_pair<T, V, -1>(s[i0], s[i1]);
    }
}

template <class T, int V>
__device__ __forceinline__ void encode(const Pk<V> (&s)[T::S], Pk<V> (&code)[T::SLOTS]) {
    sfor<T::SLOTS>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        code[p] = encode_slot<T, V, p>(s);
    });
}

// ---- codes of a parent AND all its children from one shared set of look-ups -----------------
// A face turn carries whole cubies: slot p of child a holds the cubie of parent slot q, its three (two)
// stickers read in a fixed order sigma.  So every child code is LUT[hash(parent colours of slot q in order
// sigma)] for one of few (q, sigma) pairs: 31 corner + 20 edge pairs cover the 3x3x3 parent and its 12
// children (13 x 20 = 260 slot codes), 15 cover the 2x2x2 family -- all derived at compile time.
struct SlotSrc { int q; int j[3]; };

template <class T>
constexpr SlotSrc corner_src(int a, int p) {   // a < 0: the parent itself
    SlotSrc r{p, {0, 1, 2}};
    if (a < 0) return r;
    for (int q = 0; q < T::NC; ++q)
        for (int k = 0; k < 3; ++k)
            if (kPerm<T>.v[a][T::cdef[p][0]] == T::cdef[q][k]) r.q = q;
    for (int k = 0; k < 3; ++k)
        for (int j = 0; j < 3; ++j)
            if (kPerm<T>.v[a][T::cdef[p][k]] == T::cdef[r.q][j]) r.j[k] = j;
    return r;
}
template <class T>
constexpr SlotSrc edge_src(int a, int e) {
    SlotSrc r{e, {0, 1, 0}};
    if (a < 0) return r;
    for (int q = 0; q < T::NE; ++q)
        for (int k = 0; k < 2; ++k)
            if (kPerm<T>.v[a][T::edef[e][0]] == T::edef[q][k]) r.q = q;
    for (int k = 0; k < 2; ++k)
        for (int j = 0; j < 2; ++j)
            if (kPerm<T>.v[a][T::edef[e][k]] == T::edef[r.q][j]) r.j[k] = j;
    return r;
}
constexpr int corner_sigma_id(const SlotSrc &s) { return s.j[0] * 2 + (s.j[1] > s.j[2] ? 1 : 0); }   // 0..5, identity = 0
constexpr int edge_sigma_id(const SlotSrc &s) { return s.j[0]; }                                      // 0 | 1, identity = 0
template <class T>
constexpr bool corner_pair_used(int q, int id) {
    if (id == 0) return true;
    for (int a = 0; a < T::A; ++a)
        for (int p = 0; p < T::NC; ++p) {
            const SlotSrc s = corner_src<T>(a, p);
            if (s.q == q && corner_sigma_id(s) == id) return true;
        }
    return false;
}
template <class T>
constexpr bool edge_pair_used(int q, int id) {
    if (id == 0) return true;
    for (int a = 0; a < T::A; ++a)
        for (int e = 0; e < T::NE; ++e) {
            const SlotSrc s = edge_src<T>(a, e);
            if (s.q == q && edge_sigma_id(s) == id) return true;
        }
    return false;
}

template <class T, int V>
struct FamilyCodes {
    Pk<V> c[T::NC][6];
    Pk<V> e[T::NE > 0 ? T::NE : 1][2];
};

// REACH: the state is reachable from the solved cube (ADI walks), so corner codes come from the two-colour tables too;
// otherwise (rc_expand_children: any colouring) corners keep the literal hash + 72-entry table.
template <class T, int V, bool REACH>
__device__ __forceinline__ void family_codes(const Pk<V> (&s)[T::S], FamilyCodes<T, V> &f) {
    sfor<T::NC>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        sfor<6>([&](auto ic) {
            constexpr int id = decltype(ic)::value;
            if constexpr (corner_pair_used<T>(q, id)) {
                constexpr int j0 = id / 2, r0 = j0 == 0 ? 1 : 0, r1 = j0 == 2 ? 1 : 2;
                constexpr int j1 = id % 2 ? r1 : r0, j2 = id % 2 ? r0 : r1;
                constexpr int i0 = T::cdef[q][j0], i1 = T::cdef[q][j1], i2 = T::cdef[q][j2];
                if constexpr (REACH) {
                    f.c[q][id] = lut_pair<T, V, T::cpair_id[q][id]>(s[i0], s[i1]);
                } else {
                    const Pk<V> h = shl_add(s[i2], 1, shl_add(s[i2], 3, shl_add(s[i1], 1, s[i0])));   // c0 + 2 c1 + 10 c2
                    f.c[q][id] = lut72<T, V, true>(h);
                }
            }
        });
    });
    sfor<T::NE>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        sfor<2>([&](auto ic) {
            constexpr int id = decltype(ic)::value;
            if constexpr (edge_pair_used<T>(q, id)) {
                constexpr int i0 = T::edef[q][id], i1 = T::edef[q][1 - id];
                f.e[q][id] = lut_pair<T, V, -1>(s[i0], s[i1]);                                       // ecode[c0 + 10 c1]
            }
        });
    });
}

// codes of child A_ (A_ = -1: the parent) picked out of the family's look-ups
template <class T, int V, int A_>
__device__ __forceinline__ void family_pick(const FamilyCodes<T, V> &f, Pk<V> (&code)[T::SLOTS]) {
    sfor<T::SLOTS>([&](auto pc) {
        constexpr int p = decltype(pc)::value;
        if constexpr (p < T::NC) {
            constexpr SlotSrc src = corner_src<T>(A_, p);
            code[p] = f.c[src.q][corner_sigma_id(src)];
        } else {
            constexpr SlotSrc src = edge_src<T>(A_, p - T::NC);
            code[p] = f.e[src.q][edge_sigma_id(src)];
        }
    });
}

// ---- the FAMILY record: the shared look-ups of a parent as stored rows ---------------------------------
// Instead of the 13 x SLOTS picked codes of a parent and its children (260 bytes per 3x3x3 state), the ADI generator can emit the
// family look-ups themselves: one byte per used (slot, reading order) pair -- NF = 51 rows for the 3x3x3 (31 corner + 20 edge), 15
// for the 2x2x2 -- in this fixed order: corner slots q = 0 .. NC-1, reading orders id = 0 .. 5 where used, then edge slots q, id = 0, 1.
// row[a][p] = the family row that IS slot p's code of child a (a = A: the parent itself), i.e. family_pick as a table.
template <class T>
struct FamilyLayout {
    int nf, nfc;
    uint8_t cidx[T::NC][6], eidx[T::NE > 0 ? T::NE : 1][2];
    uint8_t row[T::A + 1][T::SLOTS];
    constexpr FamilyLayout() : nf(0), nfc(0), cidx{}, eidx{}, row{} {
        for (int q = 0; q < T::NC; ++q)
            for (int id = 0; id < 6; ++id)
                if (corner_pair_used<T>(q, id)) cidx[q][id] = (uint8_t)nf++;
        nfc = nf;
        for (int q = 0; q < T::NE; ++q)
            for (int id = 0; id < 2; ++id)
                if (edge_pair_used<T>(q, id)) eidx[q][id] = (uint8_t)nf++;
        for (int a = 0; a <= T::A; ++a)
            for (int p = 0; p < T::SLOTS; ++p) {
                if (p < T::NC) {
                    const SlotSrc src = corner_src<T>(a == T::A ? -1 : a, p);
                    row[a][p] = cidx[src.q][corner_sigma_id(src)];
                } else {
                    const SlotSrc src = edge_src<T>(a == T::A ? -1 : a, p - T::NC);
                    row[a][p] = eidx[src.q][edge_sigma_id(src)];
                }
            }
    }
};
template <class T>
inline constexpr FamilyLayout<T> kFamily{};

// ------------------------------------------------------- dense one-hot from an LDS code tile
// lds_code: [SLOTS][tp] bytes (tp = padded tile width), `ncubes` valid cubes, output element
// type E in {uint8_t, uint16_t (IEEE half bits), Bf16, float}; out points at the tile's first cube.
template <class E> struct One;
template <> struct One<uint8_t> { static constexpr uint32_t v = 1u; };
template <> struct One<uint16_t> { static constexpr uint32_t v = 0x3C00u; };
struct Bf16 { uint16_t bits; };   // bfloat16 payload (RC_FMT_BF16)
template <> struct One<Bf16> { static constexpr uint32_t v = 0x3F80u; };
template <> struct One<float> { static constexpr uint32_t v = 0x3F800000u; };

template <int BYTES> struct UIntOf;
template <> struct UIntOf<1> { using type = uint8_t; };
template <> struct UIntOf<2> { using type = uint16_t; };
template <> struct UIntOf<4> { using type = uint32_t; };

template <class T>
__device__ __forceinline__ bool onehot_bit(const uint8_t *lds_code, int tp, uint32_t e) {
    constexpr uint32_t RC = T::R * T::C;
    const uint32_t cube = e / RC, w = e - cube * RC;
    const uint32_t r = w / T::C, c = w - r * T::C;
    if constexpr (T::SIZE == 3) {
        return lds_code[r * tp + cube] == c;                    // row = slot, column = code
    } else {
        const uint32_t slot = c / 3, o = c - slot * 3;          // row = piece, column = slot*3+ori
        return lds_code[slot * tp + cube] == r * 3 + o;
    }
}

// flags of the 4 consecutive elements e..e+3 (e % 4 == 0), bit j = element e+j
template <class T>
__device__ __forceinline__ uint32_t onehot_bits4(const uint8_t *lds_code, int tp, uint32_t e, uint32_t total) {
    if constexpr (T::SIZE == 3) {
        // 24 % 4 == 0: the four elements share one row
        const uint32_t cube = e / 480u, w = e - cube * 480u;
        const uint32_t r = w / 24u, c0 = w - r * 24u;
        const uint32_t d = (uint32_t)lds_code[r * tp + cube] - c0;
        return d < 4u ? (1u << d) : 0u;
    } else {
        uint32_t b = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (e + j < total && onehot_bit<T>(lds_code, tp, e + j)) b |= 1u << j;
        return b;
    }
}

// 3x3x3 fast path (256-thread workgroups): a cube is 480 elements = 120 / 60 / 30 sixteen-byte chunks (f32 / 16-bit / u8), so
// 240 threads cover exactly 2 / 4 / 8 whole cubes per pass and thread t ALWAYS writes the same chunk of a cube: its one-hot
// row(s) and columns are fixed before the loop, a pass costs one LDS byte read, a few compares and one 16-byte store
// (the generic loop below re-derives cube / row / column from the element index with two divisions per store).
// NT = writer threads per pass, a multiple of 120 (so that it is a multiple of CPC for every format): 240 of a 256-thread
// workgroup, 960 of the wide 960-thread form (15 waves sweeping ONE contiguous stream per compute unit).
// The loop is software-pipelined: the code bytes of RC_DENSE_PIPE passes are read from LDS before the first of their stores is
// issued, so a wave has that many 16-byte stores in flight per LDS round trip instead of one.
// Build-time switches of the round-4 control experiment (tools/dense_control.py; the shipped build defines none of them):
//   RC_DENSE_CTRL 1  the LDS read is replaced by a register value (what is left is address arithmetic + stores)
//   RC_DENSE_CTRL 2  as 1, and the kernels skip the code loads, LDS writes and barriers too (store-only)
//   RC_DENSE_AUX     cache policy of the dense stores (default: write-once stream, sc0 sc1 nt)
#ifndef RC_DENSE_PIPE
#define RC_DENSE_PIPE 4
#endif
#ifndef RC_DENSE_CTRL
#define RC_DENSE_CTRL 0
#endif
#ifndef RC_DENSE_AUX
#define RC_DENSE_AUX kAuxStreamStore
#endif

// the 16-byte chunk whose element d (of EPT) is the 1; all zero when d >= EPT
template <class E>
__device__ __forceinline__ Pk<4> onehot_chunk(uint32_t d) {
    Pk<4> u;
    if constexpr (sizeof(E) == 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) u.d[j] = d == (uint32_t)j ? One<E>::v : 0u;
    } else {
        // the 1 sits in element d of 8: one 64-bit shift gives the half it falls into, two compares pick the half
        const uint64_t x = (uint64_t)One<E>::v << ((d & 3u) << 4);
        const uint32_t xl = (uint32_t)x, xh = (uint32_t)(x >> 32);
        const bool lo_half = d < 4u, hi_half = d - 4u < 4u;
        u.d[0] = lo_half ? xl : 0u; u.d[1] = lo_half ? xh : 0u;
        u.d[2] = hi_half ? xl : 0u; u.d[3] = hi_half ? xh : 0u;
    }
    return u;
}

template <class T, class E, int NT = 240>
__device__ __forceinline__ void dense_write_333(const uint8_t *lds_code, int tp, E *out, int ncubes, int tid) {
    static_assert(T::SIZE == 3 && T::R * T::C == 480 && T::C % 4 == 0 && NT % 120 == 0);
    constexpr int EPT = 16 / (int)sizeof(E), CPC = 480 / EPT, CPP = NT / CPC;   // elements per chunk, chunks per cube, cubes per pass
    constexpr int U = RC_DENSE_PIPE;
    constexpr uint32_t STEP = CPP * 480u * (uint32_t)sizeof(E);                 // bytes between a thread's consecutive passes
    if (tid >= NT) return;
    const int sub = tid / CPC, k = tid - sub * CPC;
    const __amdgpu_buffer_rsrc_t srd = make_srd(out);                           // `out` is workgroup-uniform
    uint32_t off = (uint32_t)sub * 480u * (uint32_t)sizeof(E) + (uint32_t)k * 16u;
    // WHOLE: every pass of the pipelined group exists (ncubes is a multiple of U * CPP -- any full tile): no guards in the loop
    auto sweep = [&](auto whole_c) {
        constexpr bool WHOLE = decltype(whole_c)::value;
        if constexpr (sizeof(E) >= 2) {                                          // the whole chunk lies in one row (24 % EPT == 0)
            const int e0 = k * EPT, r = e0 / T::C;
            const uint32_t c0 = (uint32_t)(e0 - r * T::C);
            const uint8_t *src = lds_code + r * tp;
            for (int cube = sub; cube < ncubes; cube += U * CPP, off += U * STEP) {
                uint32_t d[U];                                                   // which element of the chunk is the 1 (if < EPT)
#pragma unroll
                for (int j = 0; j < U; ++j) {
                    const int c = cube + j * CPP;
#if RC_DENSE_CTRL
                    d[j] = (uint32_t)(c & 7);
#else
                    d[j] = (WHOLE || c < ncubes ? (uint32_t)src[c] : 0xffu) - c0;
#endif
                }
#pragma unroll
                for (int j = 0; j < U; ++j)
                    if (WHOLE || cube + j * CPP < ncubes) bst<4, RC_DENSE_AUX>(srd, off + (uint32_t)j * STEP, 0, onehot_chunk<E>(d[j]));
            }
        } else {                                                                 // u8: 16 elements may straddle two rows; per dword (4 elements, one row)
            int rr[4];
            uint32_t cc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) { const int e = k * 16 + 4 * j; rr[j] = e / T::C; cc[j] = (uint32_t)(e - rr[j] * T::C); }
            for (int cube = sub; cube < ncubes; cube += U * CPP, off += U * STEP) {
                Pk<4> u[U];
#pragma unroll
                for (int q = 0; q < U; ++q) {
                    const int c = cube + q * CPP;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
#if RC_DENSE_CTRL
                        const uint32_t d = (uint32_t)((c + j) & 7);
#else
                        const uint32_t d = (WHOLE || c < ncubes ? (uint32_t)lds_code[rr[j] * tp + c] : 0xffu) - cc[j];
#endif
                        u[q].d[j] = d < 4u ? 1u << (8u * d) : 0u;
                    }
                }
#pragma unroll
                for (int q = 0; q < U; ++q)
                    if (WHOLE || cube + q * CPP < ncubes) bst<4, RC_DENSE_AUX>(srd, off + (uint32_t)q * STEP, 0, u[q]);
            }
        }
    };
    if (ncubes % (U * CPP) == 0) sweep(std::true_type{});                        // workgroup-uniform
    else sweep(std::false_type{});
}

// 2x2x2 fast path (320-thread workgroups).  147 elements per cube are not a whole number of 16-byte chunks, but EPT cubes are exactly
// 147 chunks: a PASS = 4 / 8 / 16 whole cubes (f32 / 16-bit / u8) = 2352 bytes.  Threads 0..146 write one pass, 147..293 the next
// (294 of 320 lanes busy), and a thread always writes the same chunk of its pass -- which cube of the pass, which slot's code and which
// value make each of its elements a 1 (row = piece, column = slot * 3 + orientation, cube_env.py:143-147) is fixed before the loop,
// so an element costs one LDS byte read and one compare.  The generic loop below re-derives cube / piece / slot / orientation with four
// divisions PER ELEMENT: 1M cubes ran at 0.66 (f32), 0.45 (16-bit), 0.23 (u8) of the HBM peak, falling with the element size --
// instruction-bound, not store-bound (profiles/r05_dense222.json).
template <class T, class E, int NT, int U = 1>
__device__ __forceinline__ void dense_write_222(const uint8_t *lds_code, int tp, E *out, int ncubes, int tid) {
    static_assert(T::SIZE == 2 && T::R * T::C == 147);
    constexpr int EPT = 16 / (int)sizeof(E), CPP = EPT, NCH = 147, PPI = NT / NCH;   // CPP cubes * 147 elements = NCH chunks; PPI passes per round
    static_assert(PPI >= 1 && U >= 1);
    const int q = tid / NCH, t = tid - q * NCH;                                    // pass of the round, chunk of the pass
    if (q >= PPI) return;
    uint32_t off[EPT], want[EPT];
#pragma unroll
    for (int j = 0; j < EPT; ++j) {
        const uint32_t g = (uint32_t)t * EPT + j, c = g / 147u, e = g - c * 147u;
        const uint32_t piece = e / 21u, rem = e - piece * 21u, slot = rem / 3u, ori = rem - slot * 3u;
        off[j] = slot * (uint32_t)tp + c + (uint32_t)q * CPP;                      // LDS byte of that cube's code of `slot`, relative to the round
        want[j] = piece * 3u + ori;                                               // the code that puts the 1 on this element
    }
    const uint32_t total = (uint32_t)ncubes * 147u;
    const __amdgpu_buffer_rsrc_t srd = make_srd(out);                             // `out` is workgroup-uniform
    // U rounds per loop iteration (software pipeline, like dense_write_333): the LDS bytes of all U rounds are read before the first
    // of their stores is issued.  Cubes past ncubes: stale but in-bounds tile bytes (the tile width is a multiple of U * CPP * PPI
    // for every shipped shape, and tp = TILE + 4 covers CPP - 1 <= 3 more for the f32 pass), never stored.
    for (int base = 0; base + q * CPP < ncubes; base += U * CPP * PPI) {
        uint32_t w[U][4];
#pragma unroll
        for (int r = 0; r < U; ++r) {
            w[r][0] = w[r][1] = w[r][2] = w[r][3] = 0u;
            if (U > 1 && base + r * CPP * PPI + q * CPP >= ncubes) continue;       // a round past the tile: nothing to read or store
#pragma unroll
            for (int j = 0; j < EPT; ++j) {
                const uint32_t one = (uint32_t)lds_code[off[j] + base + r * CPP * PPI] == want[j] ? One<E>::v : 0u;
                w[r][j * (int)sizeof(E) / 4] |= one << (8 * ((j * (int)sizeof(E)) & 3));
            }
        }
#pragma unroll
        for (int r = 0; r < U; ++r) {
            const int first = base + r * CPP * PPI + q * CPP;                     // first cube of this thread's pass
            if (first >= ncubes) break;
            const uint32_t e0 = (uint32_t)first * 147u + (uint32_t)t * EPT;
            if (e0 + EPT <= total) {
                Pk<4> u;
                u.d[0] = w[r][0]; u.d[1] = w[r][1]; u.d[2] = w[r][2]; u.d[3] = w[r][3];
                bst<4, kAuxStreamStore>(srd, e0 * (uint32_t)sizeof(E), 0, u);
            } else {                                                              // ragged end of the last pass
                for (uint32_t j = 0; e0 + j < total; ++j) {
                    const uint32_t bit = j * (uint32_t)sizeof(E) * 8u;
                    out[e0 + j] = __builtin_bit_cast(E, (typename UIntOf<sizeof(E)>::type)(w[r][bit >> 5] >> (bit & 31u)));
                }
            }
        }
    }
}

// 2x2x2 float32 shape (round 6 A/B at 2^12 .. 2^22 cubes, tools/exp/d222_f32.py, profiles/r06_d222_f32.json; fraction of the HBM peak at
// 2^16+3 / 2^20 / 2^22 cubes, 64-cube tiles): the generic loop on 256 threads 0.49 / 0.65 / 0.63; the fixed mapping on 320 threads 0.41 /
// 0.49 / 0.48 (2 or 4 rounds in flight: the same), on 448 threads 0.43 / 0.54 / 0.47, on 640 threads 0.58 / 0.67 / 0.67.  Cutting the
// instructions per store by 4 x does not help by itself -- the launch is not VALU-bound for f32; what helps is a longer contiguous burst
// per workgroup round (588 lanes x 16 B = 9.4 KB).  RC_D222_F32 = 0: the generic loop; = NT: dense_write_222 on NT threads with
// RC_D222_F32_PIPE rounds in flight.
#ifndef RC_D222_F32
#define RC_D222_F32 640
#endif
#ifndef RC_D222_F32_PIPE
#define RC_D222_F32_PIPE 2
#endif

template <class T, class E>
__device__ __forceinline__ void dense_write(const uint8_t *lds_code, int tp, E *out, int ncubes, int tid, int nthreads) {
    if constexpr (T::SIZE == 3) {
        if (nthreads == 256) { dense_write_333<T, E>(lds_code, tp, out, ncubes, tid); return; }
    } else if constexpr (sizeof(E) == 4 && RC_D222_F32 != 0) {
        if (nthreads == RC_D222_F32) { dense_write_222<T, E, RC_D222_F32, RC_D222_F32_PIPE>(lds_code, tp, out, ncubes, tid); return; }
    } else {
        if (nthreads == 320) { dense_write_222<T, E, 320>(lds_code, tp, out, ncubes, tid); return; }
    }
    constexpr int EPT = 16 / (int)sizeof(E);  // elements per 16-byte store
    const uint32_t total = (uint32_t)ncubes * T::R * T::C;
    const uint32_t chunks = (total + EPT - 1) / EPT;
    const __amdgpu_buffer_rsrc_t srd = make_srd(out);                           // `out` is workgroup-uniform
    for (uint32_t ch = tid; ch < chunks; ch += nthreads) {
        const uint32_t e0 = ch * EPT;
        uint32_t w[4];
        if constexpr (sizeof(E) == 4) {
            const uint32_t b = onehot_bits4<T>(lds_code, tp, e0, total);
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = (b >> j & 1u) ? One<E>::v : 0u;
        } else if constexpr (sizeof(E) == 2) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const uint32_t b = onehot_bits4<T>(lds_code, tp, e0 + 4 * g, total);
                w[2 * g] = ((b & 1u) ? One<E>::v : 0u) | ((b & 2u) ? One<E>::v << 16 : 0u);
                w[2 * g + 1] = ((b & 4u) ? One<E>::v : 0u) | ((b & 8u) ? One<E>::v << 16 : 0u);
            }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const uint32_t b = onehot_bits4<T>(lds_code, tp, e0 + 4 * g, total);
                w[g] = (b & 1u) | ((b & 2u) << 7) | ((b & 4u) << 14) | ((b & 8u) << 21);
            }
        }
        if (e0 + EPT <= total) {
            Pk<4> u;
            u.d[0] = w[0]; u.d[1] = w[1]; u.d[2] = w[2]; u.d[3] = w[3];
            bst<4, kAuxStreamStore>(srd, e0 * (uint32_t)sizeof(E), 0, u);      // write-once stream (tile bytes < 2^32)
        } else {  // ragged end (2x2x2 only: 147 elements per cube)
            for (uint32_t j = 0; e0 + j < total; ++j) {
                const uint32_t bit = j * (uint32_t)sizeof(E) * 8u;   // element j inside the 128-bit chunk
                out[e0 + j] = __builtin_bit_cast(E, (typename UIntOf<sizeof(E)>::type)(w[bit >> 5] >> (bit & 31u)));
            }
        }
    }
}

// --------------------------------------------------------------------------- ADI RNG
// DESIGN.md "RNG": per-walk xoroshiro128+ (24,16,37) seeded through splitmix64 from
// (seed, stream_id, walk); action = (high 32 bits of the output * A) >> 32.
__device__ __forceinline__ uint64_t sm64_next(uint64_t &st) {
    uint64_t z = (st += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct WalkRng {
    uint64_t s0, s1;
    __device__ __forceinline__ void seed(uint64_t sd, uint64_t stream, uint64_t walk) {
        uint64_t st = sd;
        const uint64_t a = sm64_next(st);
        st = a ^ stream;
        const uint64_t b = sm64_next(st);
        st = b ^ walk;
        s0 = sm64_next(st);
        s1 = sm64_next(st);
        if ((s0 | s1) == 0) s1 = 0x9E3779B97F4A7C15ull;
    }
    __device__ __forceinline__ uint32_t action(uint32_t A) {
        const uint64_t r = s0 + s1;
        uint64_t t = s1 ^ s0;
        s0 = ((s0 << 24) | (s0 >> 40)) ^ t ^ (t << 16);
        s1 = (t << 37) | (t >> 27);
        return __umulhi((uint32_t)(r >> 32), A);
    }
};

}  // namespace rc
