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

template <int V, int AUX> __device__ __forceinline__ Pk<V> bld(__amdgpu_buffer_rsrc_t r, uint32_t lo, uint32_t so) {
    Pk<V> p;
    if constexpr (V == 1) p.d[0] = __builtin_amdgcn_raw_buffer_load_b32(r, lo, so, AUX);
    else if constexpr (V == 2) { u32x2 u = __builtin_amdgcn_raw_buffer_load_b64(r, lo, so, AUX); p.d[0] = u[0]; p.d[1] = u[1]; }
    else { u32x4 u = __builtin_amdgcn_raw_buffer_load_b128(r, lo, so, AUX); p.d[0] = u[0]; p.d[1] = u[1]; p.d[2] = u[2]; p.d[3] = u[3]; }
    return p;
}
template <int V, int AUX> __device__ __forceinline__ void bst(__amdgpu_buffer_rsrc_t r, uint32_t lo, uint32_t so, Pk<V> p) {
    if constexpr (V == 1) __builtin_amdgcn_raw_buffer_store_b32(p.d[0], r, lo, so, AUX);
    else if constexpr (V == 2) { u32x2 u = {p.d[0], p.d[1]}; __builtin_amdgcn_raw_buffer_store_b64(u, r, lo, so, AUX); }
    else { u32x4 u = {p.d[0], p.d[1], p.d[2], p.d[3]}; __builtin_amdgcn_raw_buffer_store_b128(u, r, lo, so, AUX); }
}

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
