// exp_dense.hip -- round-2 experiment: compact code [20][N] -> dense f32 one-hot [N][20][24] (1920 B per cube, write-bound).
// The shipped k_code_to_dense runs 305-390 us for 1M cubes depending on WHERE the 2 GB output lives (bimodal between
// allocations).  Which shape of the writer is fast on EVERY placement?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/exp/exp_dense tools/exp/exp_dense.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_srd(const void *p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, 0xffffffff, 0x00020000);
}

// TILE cubes per workgroup pass, 256 threads (240 write: thread t owns chunk t % 120 of cube t / 120 of each pair).
// PERSIST: grid-stride over tiles.  AUX: cache bits of the stores.
template <int TILE, bool PERSIST, int AUX>
__global__ void __launch_bounds__(256) k_dense(const unsigned char *code, int64_t n, int64_t pitch, float *dense) {
    constexpr int TP = TILE + 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[20 * TP];
    const int tid = threadIdx.x;
    for (int64_t tile0 = (int64_t)blockIdx.x * TILE; tile0 < n; tile0 += (int64_t)gridDim.x * TILE) {
        if (tid * 4 < TILE) {
#pragma unroll
            for (int p = 0; p < 20; ++p) *reinterpret_cast<unsigned *>(lds + p * TP + tid * 4) = *reinterpret_cast<const unsigned *>(code + p * pitch + tile0 + tid * 4);
        }
        __syncthreads();
        if (tid < 240) {
            const int sub = tid / 120, k = tid - sub * 120;
            const int r = k / 6;
            const unsigned c0 = (unsigned)(k - r * 6) * 4u;
            const __amdgpu_buffer_rsrc_t srd = make_srd(dense + tile0 * 480);
            unsigned off = (unsigned)sub * 1920u + (unsigned)k * 16u;
            const int ncubes = n - tile0 < TILE ? (int)(n - tile0) : TILE;
            for (int cube = sub; cube < ncubes; cube += 2, off += 3840u) {
                const unsigned d = (unsigned)lds[r * TP + cube] - c0;
                u32x4 u = {d == 0 ? 0x3F800000u : 0u, d == 1 ? 0x3F800000u : 0u, d == 2 ? 0x3F800000u : 0u, d == 3 ? 0x3F800000u : 0u};
                __builtin_amdgcn_raw_buffer_store_b128(u, srd, off, 0, AUX);
            }
        }
        if (!PERSIST) return;
        __syncthreads();
    }
}

// wave-per-tile variant: 64 threads, each wave owns TILE cubes; lane l writes chunks l, l+64, ... of the tile's 120*TILE chunks
// (every store instruction of the wave = 1 KiB contiguous), codes read straight from global/L2 (no LDS, no barrier)
template <int TILE, int AUX>
__global__ void __launch_bounds__(64) k_dense_wave(const unsigned char *code, int64_t n, int64_t pitch, float *dense) {
    const int lane = threadIdx.x;
    for (int64_t tile0 = (int64_t)blockIdx.x * TILE; tile0 < n; tile0 += (int64_t)gridDim.x * TILE) {
        const __amdgpu_buffer_rsrc_t srd = make_srd(dense + tile0 * 480);
        const int ncubes = n - tile0 < TILE ? (int)(n - tile0) : TILE;
        // chunk index ch = cube * 120 + k; lane handles ch = lane + 64 * i
        int cube = 0, k = lane;                                                  // lane < 120
        for (int ch = lane; ch < ncubes * 120; ch += 64) {
            const int r = k / 6;
            const unsigned c0 = (unsigned)(k - r * 6) * 4u;
            const unsigned d = (unsigned)code[r * pitch + tile0 + cube] - c0;
            u32x4 u = {d == 0 ? 0x3F800000u : 0u, d == 1 ? 0x3F800000u : 0u, d == 2 ? 0x3F800000u : 0u, d == 3 ? 0x3F800000u : 0u};
            __builtin_amdgcn_raw_buffer_store_b128(u, srd, (unsigned)ch * 16u, 0, AUX);
            k += 64;
            if (k >= 120) { k -= 120; ++cube; }
        }
    }
}


// 1024-thread workgroups, one per CU, sweeping memory like the runtime's fill kernel: 960 threads write 8 whole cubes
// (15360 contiguous bytes) per pass, TILE cubes per iteration, grid-stride over tiles.
template <int TILE, int AUX>
__global__ void __launch_bounds__(1024) k_dense_big(const unsigned char *code, int64_t n, int64_t pitch, float *dense) {
    constexpr int TP = TILE + 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[20 * TP];
    const int tid = threadIdx.x;
    for (int64_t tile0 = (int64_t)blockIdx.x * TILE; tile0 < n; tile0 += (int64_t)gridDim.x * TILE) {
        if (tid * 4 < TILE) {
#pragma unroll
            for (int p = 0; p < 20; ++p) *reinterpret_cast<unsigned *>(lds + p * TP + tid * 4) = *reinterpret_cast<const unsigned *>(code + p * pitch + tile0 + tid * 4);
        }
        __syncthreads();
        if (tid < 960) {
            const int sub = tid / 120, k = tid - sub * 120;
            const int r = k / 6;
            const unsigned c0 = (unsigned)(k - r * 6) * 4u;
            const __amdgpu_buffer_rsrc_t srd = make_srd(dense + tile0 * 480);
            unsigned off = (unsigned)sub * 1920u + (unsigned)k * 16u;
            const int ncubes = n - tile0 < TILE ? (int)(n - tile0) : TILE;
            for (int cube = sub; cube < ncubes; cube += 8, off += 8u * 1920u) {
                const unsigned d = (unsigned)lds[r * TP + cube] - c0;
                u32x4 u = {d == 0 ? 0x3F800000u : 0u, d == 1 ? 0x3F800000u : 0u, d == 2 ? 0x3F800000u : 0u, d == 3 ? 0x3F800000u : 0u};
                __builtin_amdgcn_raw_buffer_store_b128(u, srd, off, 0, AUX);
            }
        }
        __syncthreads();
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
    CK(hipGetLastError());
    std::sort(ts.begin(), ts.end());
    return ts[2] * 1e-3;
}

int main() {
    const int64_t n = 1 << 20, pitch = n;
    unsigned char *code;
    CK(hipMalloc(&code, 20 * pitch));
    std::vector<unsigned char> h(20 * pitch);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned char)((i * 2654435761u >> 13) % 24);
    CK(hipMemcpy(code, h.data(), h.size(), hipMemcpyHostToDevice));
    const size_t bytes = (size_t)n * 1920;
    float *bufs[4];
    void *pad[4];
    for (int i = 0; i < 4; ++i) { CK(hipMalloc(&bufs[i], bytes)); CK(hipMalloc(&pad[i], (size_t)(37 + 64 * i) << 20)); }
    for (int p = 0; p < 4; ++p) {
        float *out = bufs[p];
        printf("--- placement %d (%p)\n", p, (void *)out);
        { double t = timeit([&] { CK(hipMemsetAsync(out, 0, bytes, 0)); }); printf("hipMemsetAsync                      : %7.1f us %7.1f GB/s\n", t * 1e6, bytes / t / 1e9); }
#define RUN(TILE, PERSIST, AUX, GRID) { const int64_t tiles = (n + TILE - 1) / TILE; const int64_t g = PERSIST ? std::min<int64_t>(tiles, GRID) : tiles; \
        double t = timeit([&] { hipLaunchKernelGGL((k_dense<TILE, PERSIST, AUX>), dim3(g), dim3(256), 0, 0, code, n, pitch, out); }); \
        printf("wg   tile %4d persist %d aux %2d grid %6lld: %7.1f us %7.1f GB/s\n", TILE, (int)PERSIST, AUX, (long long)g, t * 1e6, bytes / t / 1e9); fflush(stdout); }
#define RUNW(TILE, AUX, GRID) { const int64_t tiles = (n + TILE - 1) / TILE; const int64_t g = std::min<int64_t>(tiles, GRID); \
        double t = timeit([&] { hipLaunchKernelGGL((k_dense_wave<TILE, AUX>), dim3(g), dim3(64), 0, 0, code, n, pitch, out); }); \
        printf("wave tile %4d           aux %2d grid %6lld: %7.1f us %7.1f GB/s\n", TILE, AUX, (long long)g, t * 1e6, bytes / t / 1e9); fflush(stdout); }
#define RUNBIG(TILE, AUX, GRID) { double t = timeit([&] { hipLaunchKernelGGL((k_dense_big<TILE, AUX>), dim3(GRID), dim3(1024), 0, 0, code, n, pitch, out); }); \
        printf("big  tile %4d           aux %2d grid %6d: %7.1f us %7.1f GB/s\n", TILE, AUX, GRID, t * 1e6, bytes / t / 1e9); fflush(stdout); }
        RUNBIG(32, 19, 256); RUNBIG(64, 19, 256); RUNBIG(128, 19, 256); RUNBIG(64, 19, 512); RUNBIG(64, 2, 256); RUNBIG(64, 0, 256); RUNBIG(256, 19, 256); RUNBIG(8, 19, 256); RUNBIG(16, 19, 256);
        RUN(8, true, 19, 256); RUN(16, true, 19, 256); RUN(32, true, 19, 256); RUN(64, true, 19, 256); RUN(16, true, 0, 256); RUN(16, true, 2, 256); RUN(16, true, 17, 256);
        RUN(16, true, 19, 1024); RUN(32, true, 19, 1024); RUN(64, true, 19, 512);
        RUN(256, false, 19, 0); RUN(256, true, 19, 2048);
        RUN(64, false, 19, 0); RUN(32, false, 19, 0); RUN(16, false, 19, 0); RUN(8, false, 19, 0);
        RUN(16, true, 19, 2048); RUN(16, true, 19, 512); RUN(8, true, 19, 2048);
        RUN(16, false, 2, 0); RUN(16, false, 0, 0); RUN(256, false, 0, 0); RUN(256, false, 2, 0);
        RUNW(8, 19, 1 << 30); RUNW(16, 19, 1 << 30); RUNW(64, 19, 1 << 30); RUNW(8, 19, 2048); RUNW(8, 19, 1024); RUNW(8, 19, 512); RUNW(8, 19, 256); RUNW(16, 19, 256);
        RUNW(8, 0, 1 << 30); RUNW(8, 2, 1 << 30);
    }
    return 0;
}
