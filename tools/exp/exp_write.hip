// write-only bandwidth ceiling probes (development experiment)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <bool NT> __global__ void k_fill16(u32x4 *out, int64_t n16, unsigned v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (int64_t)gridDim.x * blockDim.x) {
        u32x4 x = {v, v, v, v};
        if (NT) __builtin_nontemporal_store(x, out + i); else out[i] = x;
    }
}
// each wave writes ROWS row segments of 256 B (dword per lane) at stride `pitch` -- the ADI store shape
template <bool NT> __global__ void __launch_bounds__(64) k_rows4(unsigned *out, int64_t pitch_dw, int rows, int64_t groups, unsigned v) {
    for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
        unsigned *p = out + g * 64 + threadIdx.x;
        for (int r = 0; r < rows; ++r) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; p += pitch_dw; }
    }
}
// ADI-shaped writer: [D][A][tiles][S][pitch]; one wave = 256*V walks, `parts` waves per walk group split the children
template <int V, bool NT, bool CHILD_IN_TILE> __global__ void __launch_bounds__(64) k_adi_shape(unsigned char *out, int64_t n_walks, int64_t pitch, int shift, int64_t tiles,
                                                                  int depth, int parts, unsigned v) {
    typedef unsigned int vec __attribute__((ext_vector_type(V)));
    const int64_t item = blockIdx.x, g = item / parts; const int part = (int)(item - g * parts);
    const int64_t g0 = g * (64 * 4 * V); const unsigned lo = threadIdx.x * 4 * V;
    if (g0 + lo >= n_walks) return;
    const int64_t toff = g0 + (g0 >> shift) * 53 * pitch;
    vec x; for (int k = 0; k < V; ++k) x[k] = v + k;
    for (int d = 0; d < depth; ++d)
        for (int c = part; c < 12; c += parts) {
            unsigned char *row = CHILD_IN_TILE ? out + (((int64_t)d * tiles + (g0 >> shift)) * 12 + c) * 54 * pitch + (g0 & (pitch - 1))
                                               : out + ((int64_t)(d * 12 + c) * tiles) * 54 * pitch + toff;
            asm volatile("" : "+s"(row));
#pragma unroll
            for (int i = 0; i < 54; ++i) { if (NT) __builtin_nontemporal_store(x, (vec *)(row + lo)); else *(vec *)(row + lo) = x; row += pitch; }
        }
}

// depth-sliced ADI shape: each wave lives for DS depths only (blocks ordered depth-slice major)
template <int V> __global__ void __launch_bounds__(64) k_adi_slice(unsigned char *out, int64_t n_walks, int64_t pitch, int shift, int64_t tiles,
                                                             int depth, int parts, int ds, int64_t items_per_slice, unsigned v) {
    typedef unsigned int vec __attribute__((ext_vector_type(V)));
    const int64_t slice = blockIdx.x / items_per_slice, item = blockIdx.x - slice * items_per_slice;
    const int64_t g = item / parts; const int part = (int)(item - g * parts);
    const int64_t g0 = g * (64 * 4 * V); const unsigned lo = threadIdx.x * 4 * V;
    if (g0 + lo >= n_walks) return;
    const int64_t toff = g0 + (g0 >> shift) * 53 * pitch;
    vec x; for (int k = 0; k < V; ++k) x[k] = v + k;
    const int d1 = (int)min((int64_t)depth, (slice + 1) * ds);
    for (int d = (int)(slice * ds); d < d1; ++d)
        for (int c = part; c < 12; c += parts) {
            unsigned char *row = out + ((int64_t)(d * 12 + c) * tiles) * 54 * pitch + toff;
            asm volatile("" : "+s"(row));
#pragma unroll
            for (int i = 0; i < 54; ++i) { *(vec *)(row + lo) = x; row += pitch; }
        }
}

template <class F> double timeit(F &&f, int iters = 10) {
    for (int i = 0; i < 3; i++) f();
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ts;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(e0));
        for (int i = 0; i < iters; i++) f();
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ts.push_back(ms / iters);
    }
    std::sort(ts.begin(), ts.end());
    return ts[2] * 1e-3;
}
int main() {
    const int64_t bytes = (int64_t)2400 << 20;
    void *buf; CK(hipMalloc(&buf, bytes));
    for (int grid : {1024, 2048, 4096, 16384}) {
        double t = timeit([&] { hipLaunchKernelGGL(k_fill16<false>, dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16, 1u); });
        printf("fill16 cached grid %5d: %.1f GB/s\n", grid, bytes / t / 1e9);
        t = timeit([&] { hipLaunchKernelGGL(k_fill16<true>, dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16, 1u); });
        printf("fill16 nt     grid %5d: %.1f GB/s\n", grid, bytes / t / 1e9);
    }
    { double t = timeit([&] { CK(hipMemsetAsync(buf, 1, bytes, 0)); }); printf("hipMemsetAsync: %.1f GB/s\n", bytes / t / 1e9); }
    {
        const int64_t W = 100000; const int D = 30;
        for (int64_t pitch : {100096, 4096})
        for (int V : {1, 2})
        for (int parts : {1, 3, 12})
        for (int ds : {1, 2, 5, 30}) {
            const bool tiled = pitch < W; const int64_t tiles = tiled ? (W + pitch - 1) / pitch : 1;
            int shift = 63; if (tiled) { shift = 0; while (((int64_t)1 << shift) < pitch) ++shift; }
            const int64_t groups = (W + 64 * 4 * V - 1) / (64 * 4 * V), per = groups * parts, slices = (D + ds - 1) / ds;
            double t = timeit([&] {
                if (V == 1) hipLaunchKernelGGL((k_adi_slice<1>), dim3(per * slices), dim3(64), 0, 0, (unsigned char *)buf, W, pitch, shift, tiles, D, parts, ds, per, 1u);
                else hipLaunchKernelGGL((k_adi_slice<2>), dim3(per * slices), dim3(64), 0, 0, (unsigned char *)buf, W, pitch, shift, tiles, D, parts, ds, per, 1u);
            }, 5);
            printf("adi-slice pitch %6lld V%d parts %2d ds %2d waves %6lld: %7.1f us  %.1f GB/s\n", (long long)pitch, V, parts, ds, (long long)(per * slices), t * 1e6, (double)D * 12 * 54 * W / t / 1e9);
            fflush(stdout);
        }
    }
    return 0;
}
