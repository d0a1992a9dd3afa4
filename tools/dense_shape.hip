// dense_shape.hip -- round-4 experiment: WHICH property of a write-only kernel's shape makes it depend on where the buffer lives?
// Round-4 control (profiles/r04_dense_control.json): on the same buffers hipMemsetAsync / torch.fill_ run at 0.80-0.88 of the
// 8 TB/s peak on EVERY allocation, the dense one-hot writers AND their store-only controls at 0.67-0.75 on some allocations and
// 0.83-0.93 on others.  So the limit is the shape of the store stream, not the LDS look-up.  This harness writes a constant
// pattern with plain 16-byte-per-lane stores and varies only the shape:
//   chunk   contiguous bytes one workgroup writes before it ends (non-persistent, address-ordered workgroups) -- the width of the
//           chip's write front is (resident workgroups) x chunk
//   grid    0 = one workgroup per chunk; G > 0 = G persistent workgroups, workgroup b writes chunks b, b + G, ...
//   pass    4096 (256 lanes x 16 B) or 3840 (240 lanes: the dense writers' 2-cube f32 / 4-cube bf16 pass)
//   aux     cache bits of the stores (0 default, 19 = sc0 sc1 nt)
// on NB separately hipMalloc'ed buffers, with hipMemsetAsync beside it.  One JSON line per measurement.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/dense_shape tools/dense_shape.hip
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

template <int AUX, int LANES>
__global__ void __launch_bounds__(256) k_fill(unsigned char *base, long long bytes, long long chunk, int persistent) {
    const int tid = threadIdx.x;
    const long long nchunks = (bytes + chunk - 1) / chunk;
    for (long long c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const long long c0 = c * chunk;
        const long long len = bytes - c0 < chunk ? bytes - c0 : chunk;
        const __amdgpu_buffer_rsrc_t srd = make_srd(base + c0);
        if (tid < LANES) {
            const u32x4 u = {0x3F800000u, 0u, 0u, (unsigned)tid};
            for (long long off = (long long)tid * 16; off + 16 <= len; off += LANES * 16)
                __builtin_amdgcn_raw_buffer_store_b128(u, srd, (unsigned)off, 0, AUX);
        }
        if (!persistent) return;
    }
}

// ONE pass per workgroup: LANES of 256 threads store 16 bytes each (LANES x 16 contiguous bytes per workgroup), optionally after a
// dependent byte load per lane (DEP: the code byte a dense one-hot chunk depends on -- 20 rows of one byte per 960 output bytes,
// the bf16 dense writer's access).  per_xcd > 0: blocks b, b + 8, ... (one XCD) take consecutive chunks of one eighth of the buffer.
template <int AUX, int BLOCK, int LANES, bool DEP>
__global__ void __launch_bounds__(BLOCK) k_fill_block(unsigned char *base, long long bytes, const unsigned char *side, long long side_pitch, long long per_xcd) {
    const long long chunk = per_xcd > 0 ? (long long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3) : (long long)blockIdx.x;
    const long long c0 = chunk * (LANES * 16);
    const int tid = threadIdx.x;
    if (tid >= LANES || c0 + (long long)tid * 16 + 16 > bytes) return;
    unsigned v = (unsigned)tid;
    if (DEP) v = side[(long long)((tid % 60) / 3) * side_pitch + (c0 + tid * 16) / 960];
    const u32x4 u = {0x3F800000u, 0u, 0u, v};
    __builtin_amdgcn_raw_buffer_store_b128(u, make_srd(base + c0), (unsigned)tid * 16u, 0, AUX);
}

template <int AUX, int BLOCK, int LANES, bool DEP>
float run_block(unsigned char *buf, long long bytes, const unsigned char *side, long long side_pitch, bool xcd, hipStream_t st, int iters) {
    const long long chunks = (bytes + LANES * 16 - 1) / (LANES * 16);
    const long long per_xcd = xcd ? (chunks + 7) / 8 : 0;
    const unsigned g = (unsigned)(xcd ? per_xcd * 8 : chunks);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_fill_block<AUX, BLOCK, LANES, DEP>), dim3(g), dim3(BLOCK), 0, st, buf, bytes, side, side_pitch, per_xcd);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_fill_block<AUX, BLOCK, LANES, DEP>), dim3(g), dim3(BLOCK), 0, st, buf, bytes, side, side_pitch, per_xcd);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters * 1e3f;
}

// Row-copy front (the sticker expansion as pure data movement): out[a][tile][row][:] = in[tile][(row * 5 + a) % 54][:], A = 12 children,
// 54 rows of P bytes per tile.  One 4-KiB pass per workgroup; blocks b, b + 8, ... (one XCD) walk the passes of that XCD's tiles in
// the order (tile, a, row, segment), so a tile's 54 rows are read from HBM once and found in the XCD's L2 by the other 11 children.
template <int AUX>
__global__ void __launch_bounds__(256) k_copy_front(const unsigned char *in, unsigned char *out, int tiles, int P, long long per_xcd) {
    const int segs = P / 4096;
    const long long local = blockIdx.x >> 3;                                   // pass index inside this XCD's list
    const int xcd = blockIdx.x & 7;
    if (local >= per_xcd) return;
    const long long per_tile = 12ll * 54 * segs;
    const int tl = (int)(local / per_tile);                                     // the XCD's tl-th tile = global tile tl * 8 + xcd
    const int tile = tl * 8 + xcd;
    if (tile >= tiles) return;
    long long r = local - (long long)tl * per_tile;
    const int a = (int)(r / (54 * segs)); r -= (long long)a * 54 * segs;
    const int row = (int)(r / segs), seg = (int)(r - (long long)row * segs);
    const int srow = (row * 5 + a) % 54;
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(make_srd(in + ((long long)tile * 54 + srow) * P + (long long)seg * 4096), threadIdx.x * 16u, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v, make_srd(out + (((long long)a * tiles + tile) * 54 + row) * P + (long long)seg * 4096), threadIdx.x * 16u, 0, AUX);
}

template <int AUX, int LANES>
float run(unsigned char *buf, long long bytes, long long chunk, int grid, hipStream_t st, int iters) {
    const long long nchunks = (bytes + chunk - 1) / chunk;
    const unsigned g = grid > 0 ? (unsigned)std::min<long long>(grid, nchunks) : (unsigned)nchunks;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_fill<AUX, LANES>), dim3(g), dim3(256), 0, st, buf, bytes, chunk, grid > 0);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_fill<AUX, LANES>), dim3(g), dim3(256), 0, st, buf, bytes, chunk, grid > 0);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return ms / iters * 1e3f;
}

int main(int argc, char **argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 6;
    const long long bytes = argc > 2 ? atoll(argv[2]) : 2013265920ll;      // 2^20 cubes x 1920 B
    const int iters = 5;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::vector<unsigned char *> bufs(nb);
    for (auto &b : bufs) CK(hipMalloc(&b, bytes));
    auto emit = [&](const char *what, int buf, long long chunk, int grid, int pass, int aux, float us) {
        printf("{\"what\": \"%s\", \"buf\": %d, \"chunk\": %lld, \"grid\": %d, \"pass\": %d, \"aux\": %d, \"us\": %.1f, \"frac\": %.4f}\n", what, buf, chunk, grid,
               pass, aux, us, bytes / (us * 1e-6) / 8e12);
        fflush(stdout);
    };
    const long long chunks[] = {4096, 8192, 15360};
    const long long side_pitch = ((bytes / 960 + 4096) >> 8) << 8;                // one byte per 960 output bytes (a bf16 cube) and row, 20 rows
    unsigned char *side;
    CK(hipMalloc(&side, side_pitch * 20));
    CK(hipMemset(side, 1, side_pitch * 20));
    if (argc > 3) {   // mode "copy": the row-copy front at 2^20 parents (32 tiles of 32768), against a fill of the same 679 MB output
        const int tiles = 32, P = 32768;
        const long long obytes = 12ll * tiles * 54 * P, ibytes = (long long)tiles * 54 * P;
        unsigned char *in;
        CK(hipMalloc(&in, ibytes));
        CK(hipMemset(in, 3, ibytes));
        const long long per_xcd = (tiles / 8) * 12ll * 54 * (P / 4096);
        for (int b = 0; b < nb; ++b) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEvent_t e0, e1;
                CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k_copy_front<19>), dim3((unsigned)(per_xcd * 8)), dim3(256), 0, st, in, bufs[b], tiles, P, per_xcd);
                CK(hipEventRecord(e0, st));
                for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_copy_front<19>), dim3((unsigned)(per_xcd * 8)), dim3(256), 0, st, in, bufs[b], tiles, P, per_xcd);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                CK(hipGetLastError());
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                const float us = ms / iters * 1e3f;
                printf("{\"what\": \"copy_front\", \"buf\": %d, \"us\": %.1f, \"frac_written_plus_read_once\": %.4f, \"frac_written_only\": %.4f}\n", b, us,
                       (obytes + ibytes) / (us * 1e-6) / 8e12, obytes / (us * 1e-6) / 8e12);
                fflush(stdout);
            }
            const float t = run_block<19, 256, 256, false>(bufs[b], obytes, side, side_pitch, true, st, iters);
            printf("{\"what\": \"fill_same_bytes\", \"buf\": %d, \"us\": %.1f, \"frac_written_only\": %.4f}\n", b, t, obytes / (t * 1e-6) / 8e12);
        }
        return 0;
    }
    for (int b = 0; b < nb; ++b) {
        {
            for (int i = 0; i < 2; ++i) CK(hipMemsetAsync(bufs[b], 0, bytes, st));
            hipEvent_t e0, e1;
            CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < iters; ++i) CK(hipMemsetAsync(bufs[b], 0, bytes, st));
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            emit("hipMemsetAsync", b, 0, 0, 0, 0, ms / iters * 1e3f);
        }
        // what = shape, chunk = bytes per workgroup, grid = 1 when blocks of one XCD take consecutive chunks, pass = workgroup size
        emit("fill_block", b, 4096, 0, 256, 19, run_block<19, 256, 256, false>(bufs[b], bytes, side, side_pitch, false, st, iters));
        emit("fill_block", b, 3840, 0, 256, 19, run_block<19, 256, 240, false>(bufs[b], bytes, side, side_pitch, false, st, iters));
        emit("fill_block", b, 4096, 1, 256, 19, run_block<19, 256, 256, false>(bufs[b], bytes, side, side_pitch, true, st, iters));
        emit("fill_block", b, 3840, 1, 256, 19, run_block<19, 256, 240, false>(bufs[b], bytes, side, side_pitch, true, st, iters));
        emit("fill_block", b, 8192, 1, 512, 19, run_block<19, 512, 512, false>(bufs[b], bytes, side, side_pitch, true, st, iters));
        emit("fill_block_dep", b, 3840, 0, 256, 19, run_block<19, 256, 240, true>(bufs[b], bytes, side, side_pitch, false, st, iters));
        emit("fill_block_dep", b, 3840, 1, 256, 19, run_block<19, 256, 240, true>(bufs[b], bytes, side, side_pitch, true, st, iters));
        emit("fill_block_dep", b, 4096, 1, 256, 19, run_block<19, 256, 256, true>(bufs[b], bytes, side, side_pitch, true, st, iters));
        emit("fill_block_dep", b, 7680, 1, 512, 19, run_block<19, 512, 480, true>(bufs[b], bytes, side, side_pitch, true, st, iters));
        emit("fill_block_dep", b, 3840, 1, 256, 0, run_block<0, 256, 240, true>(bufs[b], bytes, side, side_pitch, true, st, iters));
        for (long long ch : chunks) {
            emit("fill", b, ch, 0, 4096, 19, run<19, 256>(bufs[b], bytes, ch, 0, st, iters));
            emit("fill", b, ch, 0, 4096, 0, run<0, 256>(bufs[b], bytes, ch, 0, st, iters));
            emit("fill", b, ch, 0, 3840, 19, run<19, 240>(bufs[b], bytes, ch, 0, st, iters));
        }
        // persistent grids over small and tile-sized chunks
        for (int grid : {2048})
            for (long long ch : {15360ll}) {
                emit("fill", b, ch, grid, 4096, 19, run<19, 256>(bufs[b], bytes, ch, grid, st, iters));
                emit("fill", b, ch, grid, 3840, 19, run<19, 240>(bufs[b], bytes, ch, grid, st, iters));
            }
    }
    return 0;
}
