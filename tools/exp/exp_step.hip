// exp_step.hip -- development experiments for the step kernel's memory behaviour (not shipped).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rubiks-cube-solver_amd/csrc -o tools/exp/exp_step tools/exp/exp_step.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "rc_device.h"
using namespace rc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

using T = Cube3;
struct Args { const uint8_t *in; uint8_t *out; const uint8_t *act; uint8_t *done; int64_t n, pitch, tile; };

// address of row i for the lane: plain SoA (tile == 0) or tiled SoA [n/tile][S][tile]
template <int V, bool MOVE, bool NT, int BLOCK, bool TILED, bool INPLACE = false, int WPE = 1, int XCD = 0>
__global__ void __launch_bounds__(BLOCK, WPE) k_step(Args a) {
    int64_t blk = blockIdx.x;
    if constexpr (XCD > 0) {
        // XCD-aware: blocks b, b+8, ... run on one XCD (round-robin dispatch): give each XCD whole runs of XCD
        // consecutive blocks so that one L2 sees contiguous row segments
        const int64_t x = blk % 8, j = blk / 8;
        blk = (j / XCD) * (8 * XCD) + x * XCD + (j % XCD);
    }
    const int64_t g0 = blk * (BLOCK * 4 * V);
    const uint32_t lo = threadIdx.x * (4 * V);
    const int64_t n0 = g0 + lo;
    if (n0 >= a.n) return;
    int64_t base, rs;
    if constexpr (TILED) { const int64_t t = g0 / a.tile; base = t * T::S * a.tile + (g0 - t * a.tile); rs = a.tile; }
    else { base = g0; rs = a.pitch; }
    Pk<V> s[T::S];
    { const uint8_t *row = a.in + base;
#pragma unroll
      for (int i = 0; i < T::S; ++i) { s[i] = ld<V, NT>(row + lo); row += rs; } }
    if constexpr (MOVE) {
        const Pk<V> act = ld<V, false>(a.act + n0);
        Pk<V> m[T::A];
        action_masks<T, V>(act, m);
        {
        Pk<V> o[T::S];
        apply_move<T, V>(s, m, o);
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = o[i];
        }
    }
    { uint8_t *row = a.out + base;
#pragma unroll
      for (int i = 0; i < T::S; ++i) { st<V, NT>(row + lo, s[i]); row += rs; } }
    if constexpr (MOVE) st<V, false>(a.done + n0, done_bytes(unsolved<T, V>(s)));
}

// The design as literally stated in BASELINE.json's north_star: sticker rows staged in LDS, move applied as a
// per-cube BYTE GATHER through a constant-memory permutation table (divergent index -> vector loads).
__constant__ PermTable<Cube3> c_perm{};
template <int BLOCK>
__global__ void __launch_bounds__(BLOCK) k_step_table(Args a) {
    constexpr int TP = BLOCK * 4 + 4;                      // LDS row pitch (bytes), +4 spreads rows over banks
    __shared__ __attribute__((aligned(16))) uint8_t tile[T::S * TP];
    const int64_t g0 = (int64_t)blockIdx.x * (BLOCK * 4);
    const uint32_t lo = threadIdx.x * 4;
    const int64_t t = g0 / a.tile; const int64_t base = t * T::S * a.tile + (g0 - t * a.tile);
    { const uint8_t *row = a.in + base;
#pragma unroll
      for (int i = 0; i < T::S; ++i) { *reinterpret_cast<uint32_t *>(tile + i * TP + lo) = *reinterpret_cast<const uint32_t *>(row + lo); row += a.tile; } }
    const uint32_t act = *reinterpret_cast<const uint32_t *>(a.act + g0 + lo);
    __syncthreads();
    uint32_t uns = 0, first[6];
    uint8_t *row = a.out + base;
#pragma unroll 6
    for (int i = 0; i < T::S; ++i) {
        uint32_t o = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int src = c_perm.v[(act >> (8 * j)) & 0xff][i];
            o |= (uint32_t)tile[src * TP + lo + j] << (8 * j);
        }
        *reinterpret_cast<uint32_t *>(row + lo) = o; row += a.tile;
        if (i % 9 == 0) first[i / 9] = o; else uns |= o ^ first[i / 9];
    }
    Pk<1> u; u.d[0] = uns;
    *reinterpret_cast<uint32_t *>(a.done + g0 + lo) = done_bytes(u).d[0];
}

// persistent, software pipelined: each wave walks items; next item's rows are loaded while the current is computed
template <int V, bool NT>
__global__ void __launch_bounds__(64) k_step_pipe(Args a, int64_t items) {
    const uint32_t lo = threadIdx.x * (4 * V);
    Pk<V> cur[T::S], nxt[T::S];
    Pk<V> act_c, act_n;
    int64_t item = blockIdx.x;
    auto load = [&](int64_t it, Pk<V> (&s)[T::S], Pk<V> &act) {
        const int64_t g0 = it * (64 * 4 * V);
        const uint8_t *row = a.in + g0;
#pragma unroll
        for (int i = 0; i < T::S; ++i) { s[i] = ld<V, NT>(row + lo); row += a.pitch; }
        act = ld<V, false>(a.act + g0 + lo);
    };
    auto work = [&](int64_t it, Pk<V> (&s)[T::S], Pk<V> act) {
        const int64_t g0 = it * (64 * 4 * V);
        Pk<V> m[T::A];
        action_masks<T, V>(act, m);
        Pk<V> o[T::S];
        apply_move<T, V>(s, m, o);
        uint8_t *row = a.out + g0;
#pragma unroll
        for (int i = 0; i < T::S; ++i) { st<V, NT>(row + lo, o[i]); row += a.pitch; }
        st<V, false>(a.done + g0 + lo, done_bytes(unsolved<T, V>(o)));
    };
    if (item >= items) return;
    load(item, cur, act_c);
    while (true) {
        int64_t nx = item + gridDim.x;
        if (nx < items) load(nx, nxt, act_n);
        work(item, cur, act_c);
        if (nx >= items) break;
        item = nx; nx = item + gridDim.x;
        if (nx < items) load(nx, cur, act_c);
        work(item, nxt, act_n);
        if (nx >= items) break;
        item = nx;
    }
}

__global__ void k_copy(const uint4 *in, uint4 *out, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) out[i] = in[i];
}
__global__ void k_copy_nt(const u32x4 *in, u32x4 *out, int64_t n16) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

template <class F> double timeit(F &&f, int iters = 30) {
    for (int i = 0; i < 5; i++) f();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; i++) f();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / iters);
    }
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2] * 1e-3;
}

int main(int argc, char **argv) {
    const int64_t n = 1 << 22, pitch = n;
    uint8_t *A, *B, *act, *done;
    CK(hipMalloc(&A, 54 * pitch)); CK(hipMalloc(&B, 54 * pitch)); CK(hipMalloc(&act, n)); CK(hipMalloc(&done, n));
    std::vector<uint8_t> h(54 * pitch), ha(n);
    for (int64_t i = 0; i < 54 * pitch; i++) h[i] = (uint8_t)((i / pitch) / 9);
    for (int64_t i = 0; i < n; i++) ha[i] = (uint8_t)((i * 2654435761u >> 16) % 12);
    CK(hipMemcpy(A, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), h.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(act, ha.data(), n, hipMemcpyHostToDevice));
    const double bytes = 110.0 * n, cbytes = 108.0 * n;
    uint8_t *buf[2] = {A, B};
    auto report = [&](const char *name, double t, double by) { printf("%-44s %8.2f us  %7.1f GB/s  %6.2f Gsteps/s\n", name, t * 1e6, by / t / 1e9, n / t / 1e9); fflush(stdout); };
    {
        double t = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(2048), dim3(256), 0, 0, (const uint4 *)buf[0], (uint4 *)buf[1], 54 * pitch / 16); std::swap(buf[0], buf[1]); });
        report("linear copy uint4 grid2048x256", t, cbytes);
        t = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, (const uint4 *)buf[0], (uint4 *)buf[1], 54 * pitch / 16); std::swap(buf[0], buf[1]); });
        report("linear copy uint4 grid8192x256", t, cbytes);
        t = timeit([&] { hipLaunchKernelGGL(k_copy_nt, dim3(2048), dim3(256), 0, 0, (const u32x4 *)buf[0], (u32x4 *)buf[1], 54 * pitch / 16); std::swap(buf[0], buf[1]); });
        report("linear copy nt grid2048x256", t, cbytes);
        t = timeit([&] { CK(hipMemcpyAsync(buf[1], buf[0], 54 * pitch, hipMemcpyDeviceToDevice, 0)); std::swap(buf[0], buf[1]); });
        report("hipMemcpyAsync D2D", t, cbytes);
    }
#define RUN(V, MOVE, NT, BLOCK, TILED, TILE, NAME) { \
        Args a{buf[0], buf[1], act, done, n, pitch, TILE}; \
        const int64_t blocks = n / (BLOCK * 4 * V); \
        double t = timeit([&] { a.in = buf[0]; a.out = buf[1]; hipLaunchKernelGGL((k_step<V, MOVE, NT, BLOCK, TILED>), dim3(blocks), dim3(BLOCK), 0, 0, a); std::swap(buf[0], buf[1]); }); \
        report(NAME, t, MOVE ? bytes : cbytes); }
    {
        const int64_t n = (int64_t)1 << 22;
        const double bytes = 110.0 * n, cbytes = 108.0 * n;
        auto report = [&](const char *name, double t, double by) { printf("%-40s %8.2f us  %7.1f GB/s  %6.2f Gsteps/s\n", name, t * 1e6, by / t / 1e9, n / t / 1e9); fflush(stdout); };
#define RUNI(V, INP, WPE, NAME) { \
        Args a{buf[0], buf[1], act, done, n, pitch, 32768}; \
        const int64_t blocks = n / (64 * 4 * V); \
        double t = timeit([&] { a.in = buf[0]; a.out = buf[1]; hipLaunchKernelGGL((k_step<V, true, true, 64, true, INP, WPE>), dim3(blocks), dim3(64), 0, 0, a); std::swap(buf[0], buf[1]); }); \
        report(NAME, t, bytes); }
#define RUNX(XC, NAME) { \
        Args a{buf[0], buf[1], act, done, n, pitch, 32768}; \
        const int64_t blocks = n / (64 * 4 * 2); \
        double t = timeit([&] { a.in = buf[0]; a.out = buf[1]; hipLaunchKernelGGL((k_step<2, true, true, 64, true, false, 1, XC>), dim3(blocks), dim3(64), 0, 0, a); std::swap(buf[0], buf[1]); }); \
        report(NAME, t, bytes); }
        for (int rep = 0; rep < 3; ++rep) {
            RUNX(0, "step V2 nt tiled32768 plain block order");
            RUNX(2, "step V2 nt tiled32768 xcd runs of 2");
            RUNX(8, "step V2 nt tiled32768 xcd runs of 8");
            RUNX(64, "step V2 nt tiled32768 xcd runs of 64 (= 1 tile)");
            RUNX(512, "step V2 nt tiled32768 xcd runs of 512");
        }
    }
    return 0;
}
