// which write (and copy) access patterns reach the HBM ceiling on MI355X?  (development experiment)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// A: each workgroup owns a contiguous chunk of CH bytes
template <bool NT> __global__ void __launch_bounds__(256) k_chunk(u32x4 *out, int64_t ch16, unsigned v) {
    u32x4 *p = out + (int64_t)blockIdx.x * ch16;
    u32x4 x = {v, v, v, v};
    for (int64_t i = threadIdx.x; i < ch16; i += 256) { if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x; }
}
// B: each WAVE owns a contiguous chunk
template <bool NT> __global__ void __launch_bounds__(256) k_wavechunk(u32x4 *out, int64_t ch16, unsigned v) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    u32x4 *p = out + ((int64_t)blockIdx.x * 4 + wave) * ch16;
    u32x4 x = {v, v, v, v};
    for (int64_t i = lane; i < ch16; i += 64) { if (NT) __builtin_nontemporal_store(x, p + i); else p[i] = x; }
}
// C: grid-stride, persistent grid
template <bool NT> __global__ void __launch_bounds__(256) k_gs(u32x4 *out, int64_t n16, unsigned v) {
    u32x4 x = {v, v, v, v};
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (int64_t)gridDim.x * 256) { if (NT) __builtin_nontemporal_store(x, out + i); else out[i] = x; }
}
// copy: each WG owns a chunk
template <bool NT> __global__ void __launch_bounds__(256) k_copychunk(const u32x4 *in, u32x4 *out, int64_t ch16) {
    const u32x4 *q = in + (int64_t)blockIdx.x * ch16; u32x4 *p = out + (int64_t)blockIdx.x * ch16;
    for (int64_t i = threadIdx.x; i < ch16; i += 256) { if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(q + i), p + i); else p[i] = q[i]; }
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
    const int64_t bytes = (int64_t)2048 << 20;
    void *buf, *buf2; CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&buf2, bytes));
    { double t = timeit([&] { CK(hipMemsetAsync(buf, 1, bytes, 0)); }); printf("hipMemsetAsync 2GB: %.1f GB/s\n", bytes / t / 1e9); }
    { double t = timeit([&] { CK(hipMemcpyAsync(buf2, buf, bytes, hipMemcpyDeviceToDevice, 0)); }); printf("hipMemcpyAsync 2GB: %.1f GB/s (r+w)\n", 2.0 * bytes / t / 1e9); }
    for (int64_t ch : {4096, 16384, 65536, 262144, 1048576, 4194304}) {
        const int64_t ch16 = ch / 16;
        double t = timeit([&] { hipLaunchKernelGGL(k_chunk<false>, dim3(bytes / ch), dim3(256), 0, 0, (u32x4 *)buf, ch16, 1u); });
        double t2 = timeit([&] { hipLaunchKernelGGL(k_chunk<true>, dim3(bytes / ch), dim3(256), 0, 0, (u32x4 *)buf, ch16, 1u); });
        printf("WG-chunk %8lld B: cached %.1f  nt %.1f GB/s\n", (long long)ch, bytes / t / 1e9, bytes / t2 / 1e9);
    }
    for (int64_t ch : {1024, 4096, 16384, 65536, 262144}) {
        const int64_t ch16 = ch / 16;
        double t = timeit([&] { hipLaunchKernelGGL(k_wavechunk<false>, dim3(bytes / ch / 4), dim3(256), 0, 0, (u32x4 *)buf, ch16, 1u); });
        double t2 = timeit([&] { hipLaunchKernelGGL(k_wavechunk<true>, dim3(bytes / ch / 4), dim3(256), 0, 0, (u32x4 *)buf, ch16, 1u); });
        printf("wave-chunk %8lld B: cached %.1f  nt %.1f GB/s\n", (long long)ch, bytes / t / 1e9, bytes / t2 / 1e9);
    }
    for (int grid : {256, 512, 1024, 2048, 4096, 8192, 32768, 131072}) {
        double t = timeit([&] { hipLaunchKernelGGL(k_gs<false>, dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16, 1u); });
        double t2 = timeit([&] { hipLaunchKernelGGL(k_gs<true>, dim3(grid), dim3(256), 0, 0, (u32x4 *)buf, bytes / 16, 1u); });
        printf("grid-stride %6d WGs: cached %.1f  nt %.1f GB/s\n", grid, bytes / t / 1e9, bytes / t2 / 1e9);
    }
    for (int64_t ch : {4096, 16384, 65536, 262144, 1048576}) {
        const int64_t ch16 = ch / 16;
        double t = timeit([&] { hipLaunchKernelGGL(k_copychunk<false>, dim3(bytes / ch), dim3(256), 0, 0, (const u32x4 *)buf, (u32x4 *)buf2, ch16); });
        double t2 = timeit([&] { hipLaunchKernelGGL(k_copychunk<true>, dim3(bytes / ch), dim3(256), 0, 0, (const u32x4 *)buf, (u32x4 *)buf2, ch16); });
        printf("copy WG-chunk %8lld B: cached %.1f  nt %.1f GB/s (r+w)\n", (long long)ch, 2.0 * bytes / t / 1e9, 2.0 * bytes / t2 / 1e9);
    }
    return 0;
}
