// exp_step2.hip -- round-2 experiment: raw BUFFER loads / stores with cache-policy bits on the step kernel's
// row traffic (tiled SoA, 32768-cube tiles), against the shipped global_load/store form.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I rubiks-cube-solver_amd/csrc -o tools/exp/exp_step2 tools/exp/exp_step2.hip
// aux bits of the buffer builtins on gfx950: 1 = sc0, 2 = nt, 16 = sc1  (sc1:sc0 = scope, nt = streaming)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rc_device.h"
using namespace rc;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

using T = Cube3;
struct Args { const uint8_t *in; uint8_t *out; const uint8_t *act; uint8_t *done; int64_t n, tile; };

// bld / bst: rc_device.h

template <int V, bool MOVE, int LAUX, int SAUX>
__global__ void __launch_bounds__(64) k_step_buf(Args a) {
    const int64_t g0 = (int64_t)blockIdx.x * (64 * 4 * V);
    const uint32_t lo = threadIdx.x * (4 * V);
    if (g0 + lo >= a.n) return;
    const int64_t t = g0 / a.tile;
    const int64_t base = t * T::S * a.tile + (g0 - t * a.tile);
    const uint32_t rs = (uint32_t)a.tile;
    Pk<V> s[T::S];
    {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t *>(a.in) + base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = bld<V, LAUX>(r, lo, i * rs);
    }
    if constexpr (MOVE) {
        const Pk<V> act = ld<V, false>(a.act + g0 + lo);
        Pk<V> m[T::A];
        action_masks<T, V>(act, m);
        Pk<V> o[T::S];
        apply_move<T, V>(s, m, o);
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = o[i];
    }
    {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(a.out + base, 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int i = 0; i < T::S; ++i) bst<V, SAUX>(r, lo, i * rs, s[i]);
    }
    if constexpr (MOVE) st<V, false>(a.done + g0 + lo, done_bytes(unsolved<T, V>(s)));
}

// the shipped form (global loads / stores, nontemporal)
template <int V, bool MOVE, bool NT>
__global__ void __launch_bounds__(64) k_step_glob(Args a) {
    const int64_t g0 = (int64_t)blockIdx.x * (64 * 4 * V);
    const uint32_t lo = threadIdx.x * (4 * V);
    if (g0 + lo >= a.n) return;
    const int64_t t = g0 / a.tile;
    const int64_t base = t * T::S * a.tile + (g0 - t * a.tile);
    Pk<V> s[T::S];
    { const uint8_t *row = a.in + base;
#pragma unroll
      for (int i = 0; i < T::S; ++i) { s[i] = ld<V, NT>(row + lo); row += a.tile; } }
    if constexpr (MOVE) {
        const Pk<V> act = ld<V, false>(a.act + g0 + lo);
        Pk<V> m[T::A];
        action_masks<T, V>(act, m);
        Pk<V> o[T::S];
        apply_move<T, V>(s, m, o);
#pragma unroll
        for (int i = 0; i < T::S; ++i) s[i] = o[i];
    }
    { uint8_t *row = a.out + base;
#pragma unroll
      for (int i = 0; i < T::S; ++i) { st<V, NT>(row + lo, s[i]); row += a.tile; } }
    if constexpr (MOVE) st<V, false>(a.done + g0 + lo, done_bytes(unsolved<T, V>(s)));
}


// The design as literally stated in BASELINE.json's north_star: sticker rows staged in LDS, the move applied as a per-cube
// BYTE GATHER through a constant-memory permutation table (per-lane divergent action -> vector loads of the table).
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
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    return ts[ts.size() / 2] * 1e-3;
}

int main(int argc, char **argv) {
    const int64_t n = (argc > 1 ? atoll(argv[1]) : 22) >= 32 ? atoll(argv[1]) : (int64_t)1 << (argc > 1 ? atoi(argv[1]) : 22);
    const int64_t tile = 32768;
    uint8_t *A, *B, *act, *done;
    CK(hipMalloc(&A, 54 * n)); CK(hipMalloc(&B, 54 * n)); CK(hipMalloc(&act, n)); CK(hipMalloc(&done, n));
    std::vector<uint8_t> h(54 * n), ha(n);
    for (int64_t i = 0; i < 54 * n; i++) h[i] = (uint8_t)(((i / tile) % 54) / 9);
    for (int64_t i = 0; i < n; i++) ha[i] = (uint8_t)((i * 2654435761u >> 16) % 12);
    CK(hipMemcpy(A, h.data(), h.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), h.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(act, ha.data(), n, hipMemcpyHostToDevice));
    const bool inplace = argc > 2 && atoi(argv[2]) == 1;
    uint8_t *buf[2] = {A, inplace ? A : B};
    printf("n = %lld cubes, %s\n", (long long)n, inplace ? "in place" : "ping-pong");
    auto report = [&](const char *name, double t, double by) { printf("%-52s %8.2f us  %7.1f GB/s  %6.2f Gsteps/s\n", name, t * 1e6, by * n / t / 1e9, n / t / 1e9); fflush(stdout); };
#define RUNG(V, MOVE, NT, NAME) { \
        Args a{buf[0], buf[1], act, done, n, tile}; const int64_t blocks = n / (64 * 4 * V); \
        double t = timeit([&] { a.in = buf[0]; a.out = buf[1]; hipLaunchKernelGGL((k_step_glob<V, MOVE, NT>), dim3(blocks), dim3(64), 0, 0, a); std::swap(buf[0], buf[1]); }); \
        report(NAME, t, MOVE ? 110.0 : 108.0); }
#define RUNB(V, MOVE, LA, SA) { \
        Args a{buf[0], buf[1], act, done, n, tile}; const int64_t blocks = n / (64 * 4 * V); \
        double t = timeit([&] { a.in = buf[0]; a.out = buf[1]; hipLaunchKernelGGL((k_step_buf<V, MOVE, LA, SA>), dim3(blocks), dim3(64), 0, 0, a); std::swap(buf[0], buf[1]); }); \
        char nm[96]; snprintf(nm, sizeof nm, "%s buffer V%d ld aux %2d st aux %2d", MOVE ? "step" : "copy", V, LA, SA); report(nm, t, MOVE ? 110.0 : 108.0); }

    if (argc > 3 && atoi(argv[3]) == 1) {       // design A/B (run it under rocprofv3 --kernel-trace --stats for the kernel-stat rows)
        for (int rep = 0; rep < 2; ++rep) {
            { Args a{buf[0], buf[1], act, done, n, tile}; const int64_t blocks = n / (256 * 4);
              double t = timeit([&] { a.in = buf[0]; a.out = buf[1]; hipLaunchKernelGGL((k_step_table<256>), dim3(blocks), dim3(256), 0, 0, a); std::swap(buf[0], buf[1]); });
              report("LITERAL north_star: LDS tile + constant-table byte gather b256", t, 110.0); }
            { Args a{buf[0], buf[1], act, done, n, tile}; const int64_t blocks = n / (64 * 4);
              double t = timeit([&] { a.in = buf[0]; a.out = buf[1]; hipLaunchKernelGGL((k_step_table<64>), dim3(blocks), dim3(64), 0, 0, a); std::swap(buf[0], buf[1]); });
              report("LITERAL north_star: LDS tile + constant-table byte gather b64", t, 110.0); }
            RUNG(2, true, true, "round-1 shipped: packed select network, global nt V2");
            RUNB(2, true, 2, 17);
            RUNG(2, false, true, "row-structured copy of the same buffers (global nt V2)");
            RUNB(2, false, 2, 17);
            { double t = timeit([&] { CK(hipMemcpyAsync(buf[1], buf[0], 54 * n, hipMemcpyDeviceToDevice, 0)); std::swap(buf[0], buf[1]); });
              report("hipMemcpyAsync D2D", t, 108.0); }
        }
        return 0;
    }

    if (argc > 3 && atoi(argv[3]) == 2) {       // finer policy sweep around the round-2 choice (ld nt, st sc0 sc1)
        for (int rep = 0; rep < 2; ++rep) {
            RUNB(2, true, 2, 17); RUNB(2, true, 2, 16); RUNB(2, true, 2, 1); RUNB(2, true, 2, 0);
            RUNB(2, true, 3, 17); RUNB(2, true, 18, 17); RUNB(2, true, 19, 17); RUNB(2, true, 3, 16); RUNB(2, true, 18, 16);
            RUNB(4, true, 2, 17); RUNB(4, true, 2, 16); RUNB(1, true, 2, 17);
        }
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        RUNG(2, true, true, "step global V2 nt (shipped form)");
        RUNG(2, true, false, "step global V2 cached");
        RUNG(2, false, true, "copy global V2 nt");
        RUNB(2, true, 0, 0); RUNB(2, true, 2, 2); RUNB(2, true, 2, 19); RUNB(2, true, 19, 19); RUNB(2, true, 0, 19); RUNB(2, true, 2, 17); RUNB(2, true, 17, 17);
        RUNB(2, true, 2, 3); RUNB(2, true, 2, 18); RUNB(2, true, 3, 3); RUNB(2, true, 18, 18); RUNB(2, true, 2, 0); RUNB(2, true, 0, 2); RUNB(2, true, 16, 16); RUNB(2, true, 1, 1);
        RUNB(2, false, 2, 2); RUNB(2, false, 2, 19); RUNB(2, false, 19, 19);
        RUNB(4, true, 2, 2); RUNB(4, true, 2, 19); RUNB(1, true, 2, 2); RUNB(1, true, 2, 19);
    }
    return 0;
}
