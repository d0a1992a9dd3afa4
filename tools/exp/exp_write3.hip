// Round-2 write-path experiment: what limits the ADI / expansion output stream on MI355X?
//   (1) store instruction form: FLAT (what the round-1 kernels emitted after the asm launder),
//       GLOBAL with an SGPR base (saddr form), raw BUFFER stores (SRD + scalar row offset);
//   (2) pack width V (4 / 8 / 16 B per lane), children split (parts), tile pitch;
//   (3) occupancy cap (waves per CU) through a dummy LDS allocation;
//   (4) "sweep" shape: one short-lived wave per (depth, child, walk group) in output-address order,
//       optionally reading the parent tile first (the two-kernel design: walks -> parents, then children
//       as row-renamed copies of the parents), with the block -> tile map stable per XCD so the 12
//       re-reads of a parent tile hit the same L2.
// Store-only shapes: the bytes written are exactly the ADI child stream (depth x 12 x 54 x walks).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

enum { FLAT = 0, GLOBAL = 1, BUFFER = 2 };
template <int V> struct Vec { typedef unsigned int type __attribute__((ext_vector_type(V))); };
template <> struct Vec<1> { typedef unsigned int type; };

template <int V, int KIND>
__device__ __forceinline__ void store_rows(unsigned char *base, int64_t pitch, unsigned lo, typename Vec<V>::type x) {
    if constexpr (KIND == FLAT) {
        unsigned char *row = base;
        asm volatile("" : "+s"(row));
#pragma unroll
        for (int i = 0; i < 54; ++i) { *(typename Vec<V>::type *)(row + lo) = x; row += pitch; }
    } else if constexpr (KIND == GLOBAL) {
        unsigned char *row = base;
#pragma unroll
        for (int i = 0; i < 54; ++i) {
            if constexpr (V == 1) asm volatile("global_store_dword %0, %1, %2" ::"v"(lo), "v"(x), "s"(row) : "memory");
            else if constexpr (V == 2) asm volatile("global_store_dwordx2 %0, %1, %2" ::"v"(lo), "v"(x), "s"(row) : "memory");
            else asm volatile("global_store_dwordx4 %0, %1, %2" ::"v"(lo), "v"(x), "s"(row) : "memory");
            row += pitch;
        }
    } else {
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x00020000);
        unsigned so = 0;
#pragma unroll
        for (int i = 0; i < 54; ++i) {
            if constexpr (V == 1) __builtin_amdgcn_raw_buffer_store_b32(x, r, lo, so, 0);
            else if constexpr (V == 2) __builtin_amdgcn_raw_buffer_store_b64(x, r, lo, so, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(x, r, lo, so, 0);
            so += (unsigned)pitch;
        }
    }
}

// persistent-over-depth ADI shape: [D][12][tiles][54][pitch]
template <int V, int KIND>
__global__ void __launch_bounds__(64) k_adi_shape(unsigned char *out, int64_t n_walks, int64_t pitch, int shift, int64_t tiles, int depth, int parts, unsigned v) {
    extern __shared__ unsigned char lds_dummy[];
    const int64_t item = blockIdx.x, g = item / parts;
    const int part = (int)(item - g * parts);
    const int64_t g0 = g * (64 * 4 * V);
    const unsigned lo = threadIdx.x * 4 * V;
    if (g0 + lo >= n_walks) return;
    const int64_t toff = g0 + (g0 >> shift) * 53 * pitch;
    typename Vec<V>::type x;
    if constexpr (V == 1) x = v; else for (int k = 0; k < V; ++k) x[k] = v + k;
    for (int d = 0; d < depth; ++d)
        for (int c = part; c < 12; c += parts) {
            unsigned char *row = out + ((int64_t)(d * 12 + c) * tiles) * 54 * pitch + toff;
            store_rows<V, KIND>(row, pitch, lo, x);
        }
}

// sweep shape: block b -> (depth, child, group) in output-address order; each wave writes one child of one group and
// exits.  READ: first load the parent tile rows (54 row segments) and store those (a row-renamed copy).
// Groups are padded to a multiple of 8 so group % 8 (= the XCD of the block under round-robin dispatch) is stable
// over children: the 12 re-reads of a parent tile stay in one XCD's L2.
template <int V, int KIND, bool READ>
__global__ void __launch_bounds__(64) k_sweep(unsigned char *out, const unsigned char *parents, int64_t n_walks, int64_t groups8, int64_t pitch, int shift,
                                              int64_t tiles, unsigned v) {
    const int64_t b = blockIdx.x;
    const int64_t dc = b / groups8, g = b - dc * groups8;       // dc = depth * 12 + child
    const int64_t d = dc / 12;
    const int64_t g0 = g * (64 * 4 * V);
    const unsigned lo = threadIdx.x * 4 * V;
    if (g0 + lo >= n_walks) return;
    const int64_t toff = g0 + (g0 >> shift) * 53 * pitch;
    unsigned char *row = out + (dc * tiles) * 54 * pitch + toff;
    typedef typename Vec<V>::type vec;
    if constexpr (!READ) {
        vec x;
        if constexpr (V == 1) x = v; else for (int k = 0; k < V; ++k) x[k] = v + k;
        store_rows<V, KIND>(row, pitch, lo, x);
    } else {
        const unsigned char *prow = parents + (d * tiles) * 54 * pitch + toff;
        vec s[54];
#pragma unroll
        for (int i = 0; i < 54; ++i) { s[i] = *(const vec *)(prow + lo); prow += pitch; }
        __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(row, 0, 0x7fffffff, 0x00020000);
        unsigned so = 0;
#pragma unroll
        for (int i = 0; i < 54; ++i) {
            const vec x = s[(i * 7 + 3) % 54];                    // some fixed renaming, like a face turn
            if constexpr (V == 1) __builtin_amdgcn_raw_buffer_store_b32(x, r, lo, so, 0);
            else if constexpr (V == 2) __builtin_amdgcn_raw_buffer_store_b64(x, r, lo, so, 0);
            else __builtin_amdgcn_raw_buffer_store_b128(x, r, lo, so, 0);
            so += (unsigned)pitch;
        }
    }
}


// generation 2: BLOCK threads per workgroup (adjacent waves of one workgroup write adjacent row pieces), optional
// XCD-aware remap (workgroup b runs on XCD b % 8 under round-robin dispatch: give every XCD a contiguous range of
// walk groups, so one L2 sees whole rows), optional aux bits on the buffer store (1 = sc0, 2 = nt, 16 = sc1)
template <int V, int BLOCK, bool XCD, int AUX>
__global__ void __launch_bounds__(BLOCK) k_adi_shape2(unsigned char *out, int64_t n_walks, int64_t pitch, int shift, int64_t tiles, int depth, int parts, int64_t wgs, unsigned v) {
    int64_t item = blockIdx.x;
    if constexpr (XCD) {                         // wgs is a multiple of 8
        const int64_t per = wgs / 8;
        item = (item & 7) * per + (item >> 3);
    }
    const int64_t g = item / parts;
    const int part = (int)(item - g * parts);
    const int64_t g0 = g * (BLOCK * 4 * V) + (threadIdx.x >> 6) * (64 * 4 * V);   // wave-uniform
    const unsigned lo = (threadIdx.x & 63) * 4 * V;
    if (g0 + lo >= n_walks) return;
    const int64_t toff = g0 + (g0 >> shift) * 53 * pitch;
    typename Vec<V>::type x;
    if constexpr (V == 1) x = v; else for (int k = 0; k < V; ++k) x[k] = v + k;
    for (int d = 0; d < depth; ++d)
        for (int c = part; c < 12; c += parts) {
            unsigned char *row = out + ((int64_t)(d * 12 + c) * tiles) * 54 * pitch + toff;
            row = (unsigned char *)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((uint64_t)row >> 32)) << 32) | (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uint64_t)row));
            __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(row, 0, 0x7fffffff, 0x00020000);
            unsigned so = 0;
#pragma unroll
            for (int i = 0; i < 54; ++i) {
                if constexpr (V == 1) __builtin_amdgcn_raw_buffer_store_b32(x, r, lo, so, AUX);
                else if constexpr (V == 2) __builtin_amdgcn_raw_buffer_store_b64(x, r, lo, so, AUX);
                else __builtin_amdgcn_raw_buffer_store_b128(x, r, lo, so, AUX);
                so += (unsigned)pitch;
            }
        }
}

template <int V, int BLOCK, bool XCD, int AUX>
void run_shape2(unsigned char *buf, int64_t bytes, int64_t W, int D, int64_t pitch, int parts) {
    const bool tiled = pitch < W;
    const int64_t span = BLOCK * 4 * V;
    int64_t groups = (W + span - 1) / span;
    int64_t wgs = groups * parts;
    if (XCD) { wgs = (wgs + 7) / 8 * 8; groups = (wgs + parts - 1) / parts; }
    const int64_t wp = tiled ? pitch : (groups * span + 255) / 256 * 256;
    const int64_t tiles = tiled ? (groups * span + pitch - 1) / pitch : 1;
    int shift = 63;
    if (tiled) { shift = 0; while (((int64_t)1 << shift) < pitch) ++shift; }
    if (tiled && pitch < span) return;
    if ((int64_t)D * 12 * tiles * 54 * wp > bytes) { printf("skip (buffer too small)\n"); return; }
    double t = timeit([&] { hipLaunchKernelGGL((k_adi_shape2<V, BLOCK, XCD, AUX>), dim3(wgs), dim3(BLOCK), 0, 0, buf, W, wp, shift, tiles, D, parts, wgs, 1u); });
    printf("shape2 V%d block %3d xcd %d aux %2d pitch %7lld parts %2d wgs %5lld: %7.1f us  %7.1f GB/s\n", V, BLOCK, (int)XCD, AUX, (long long)pitch, parts, (long long)wgs,
           t * 1e6, (double)D * 12 * 54 * W / t / 1e9);
    fflush(stdout);
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_chunk(u32x4 *out, int64_t ch16, unsigned v) {
    u32x4 *p = out + (int64_t)blockIdx.x * ch16;
    u32x4 x = {v, v, v, v};
    for (int64_t i = threadIdx.x; i < ch16; i += 256) p[i] = x;
}

template <class F> double timeit(F &&f, int iters = 5) {
    for (int i = 0; i < 2; i++) f();
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

static const char *kname[] = {"flat  ", "global", "buffer"};

template <int V, int KIND>
void run_shape(unsigned char *buf, int64_t bytes, int64_t W, int D, int64_t pitch, int parts, int wpc) {
    const bool tiled = pitch < W;
    int64_t span = 64 * 4 * V;
    const int64_t groups = (W + span - 1) / span;
    const int64_t wp = tiled ? pitch : (groups * span + 255) / 256 * 256;
    const int64_t tiles = tiled ? (groups * span + pitch - 1) / pitch : 1;
    int shift = 63;
    if (tiled) { shift = 0; while (((int64_t)1 << shift) < pitch) ++shift; }
    if ((int64_t)D * 12 * tiles * 54 * wp > bytes) { printf("skip (buffer too small)\n"); return; }
    if (tiled && pitch < span) return;
    // occupancy cap: wpc waves per CU through LDS (160 KiB per CU)
    const size_t lds = wpc == 1 ? (size_t)82 * 1024 : wpc > 0 ? (size_t)(160 * 1024 / wpc) - 512 : 0;
    if (lds > 65536) CK(hipFuncSetAttribute((const void *)k_adi_shape<V, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    double t = timeit([&] { hipLaunchKernelGGL((k_adi_shape<V, KIND>), dim3(groups * parts), dim3(64), lds, 0, buf, W, wp, shift, tiles, D, parts, 1u); });
    printf("adi-shape %s V%d pitch %6lld parts %2d wpc %2d waves %5lld: %7.1f us  %7.1f GB/s\n", kname[KIND], V, (long long)pitch, parts, wpc, (long long)(groups * parts),
           t * 1e6, (double)D * 12 * 54 * W / t / 1e9);
    fflush(stdout);
}

template <int V, int KIND, bool READ>
void run_sweep(unsigned char *buf, unsigned char *par, int64_t bytes, int64_t W, int D, int64_t pitch) {
    const bool tiled = pitch < W;
    const int64_t span = 64 * 4 * V;
    const int64_t groups = (W + span - 1) / span, groups8 = (groups + 7) / 8 * 8;
    const int64_t wp = tiled ? pitch : (groups8 * span + 255) / 256 * 256;
    const int64_t tiles = tiled ? (groups8 * span + pitch - 1) / pitch : 1;
    int shift = 63;
    if (tiled) { shift = 0; while (((int64_t)1 << shift) < pitch) ++shift; }
    if (tiled && pitch < span) return;
    if ((int64_t)D * 12 * tiles * 54 * wp > bytes) { printf("skip (buffer too small)\n"); return; }
    double t = timeit([&] { hipLaunchKernelGGL((k_sweep<V, KIND, READ>), dim3(groups8 * 12 * D), dim3(64), 0, 0, buf, par, W, groups8, wp, shift, tiles, 1u); });
    printf("sweep %s %s V%d pitch %6lld waves %6lld: %7.1f us  %7.1f GB/s written\n", READ ? "copy " : "store", kname[KIND], V, (long long)pitch, (long long)(groups8 * 12 * D),
           t * 1e6, (double)D * 12 * 54 * W / t / 1e9);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int64_t bytes = (int64_t)2600 << 20;
    unsigned char *buf, *par;
    CK(hipMalloc(&buf, bytes));
    CK(hipMalloc(&par, (int64_t)256 << 20));
    CK(hipMemset(par, 1, (int64_t)256 << 20));
    const int64_t W = 100000; const int D = 30;
    { double t = timeit([&] { CK(hipMemsetAsync(buf, 1, bytes, 0)); }); printf("hipMemsetAsync: %.1f GB/s\n", bytes / t / 1e9); }
    { const int64_t ch = 4096; double t = timeit([&] { hipLaunchKernelGGL(k_chunk, dim3(bytes / ch), dim3(256), 0, 0, (u32x4 *)buf, ch / 16, 1u); }); printf("WG-chunk 4096: %.1f GB/s\n", bytes / t / 1e9); }

    if (argc > 1 && atoi(argv[1]) == 2) {
        for (int64_t pitch : {4096, 16384, 65536, 1 << 20}) for (int parts : {1, 2, 6, 12}) {
            run_shape2<2, 64, false, 0>(buf, bytes, W, D, pitch, parts);
            run_shape2<2, 64, true, 0>(buf, bytes, W, D, pitch, parts);
            run_shape2<4, 64, false, 0>(buf, bytes, W, D, pitch, parts);
            run_shape2<4, 64, true, 0>(buf, bytes, W, D, pitch, parts);
            run_shape2<1, 256, false, 0>(buf, bytes, W, D, pitch, parts);
            run_shape2<1, 256, true, 0>(buf, bytes, W, D, pitch, parts);
            run_shape2<2, 256, false, 0>(buf, bytes, W, D, pitch, parts);
            run_shape2<2, 256, true, 0>(buf, bytes, W, D, pitch, parts);
            run_shape2<4, 256, false, 0>(buf, bytes, W, D, pitch, parts);
            run_shape2<4, 256, true, 0>(buf, bytes, W, D, pitch, parts);
            run_shape2<2, 64, false, 2>(buf, bytes, W, D, pitch, parts);
            run_shape2<2, 64, false, 17>(buf, bytes, W, D, pitch, parts);
            run_shape2<2, 64, false, 19>(buf, bytes, W, D, pitch, parts);
        }
        return 0;
    }

    if (argc > 1 && atoi(argv[1]) == 3) {
        unsigned char *b2, *b3;
        CK(hipMalloc(&b2, bytes)); CK(hipMalloc(&b3, bytes));
        unsigned char *bs[3] = {buf, b2, b3};
        for (int p = 0; p < 3; ++p) {
            printf("--- placement %d (%p)\n", p, (void *)bs[p]);
            for (int64_t pitch : {4096, 8192, 16384, 32768, 65536}) for (int parts : {1, 2, 6}) {
                run_shape2<2, 64, false, 0>(bs[p], bytes, W, D, pitch, parts);
                run_shape2<2, 64, false, 2>(bs[p], bytes, W, D, pitch, parts);
                run_shape2<2, 64, false, 3>(bs[p], bytes, W, D, pitch, parts);
                run_shape2<2, 64, false, 16>(bs[p], bytes, W, D, pitch, parts);
                run_shape2<2, 64, false, 18>(bs[p], bytes, W, D, pitch, parts);
                run_shape2<2, 64, false, 19>(bs[p], bytes, W, D, pitch, parts);
                run_shape2<1, 64, false, 19>(bs[p], bytes, W, D, pitch, parts);
                run_shape2<4, 64, false, 19>(bs[p], bytes, W, D, pitch, parts);
                run_shape2<4, 64, false, 2>(bs[p], bytes, W, D, pitch, parts);
                run_shape2<1, 64, false, 2>(bs[p], bytes, W, D, pitch, parts);
            }
        }
        return 0;
    }
    // (1) instruction form x V, at the shipped layout (pitch 4096) and the round-1 best (16384)
    for (int64_t pitch : {4096, 16384}) for (int parts : {1, 6, 12}) {
        run_shape<1, FLAT>(buf, bytes, W, D, pitch, parts, 0); run_shape<1, GLOBAL>(buf, bytes, W, D, pitch, parts, 0); run_shape<1, BUFFER>(buf, bytes, W, D, pitch, parts, 0);
        run_shape<2, FLAT>(buf, bytes, W, D, pitch, parts, 0); run_shape<2, GLOBAL>(buf, bytes, W, D, pitch, parts, 0); run_shape<2, BUFFER>(buf, bytes, W, D, pitch, parts, 0);
        run_shape<4, FLAT>(buf, bytes, W, D, pitch, parts, 0); run_shape<4, GLOBAL>(buf, bytes, W, D, pitch, parts, 0); run_shape<4, BUFFER>(buf, bytes, W, D, pitch, parts, 0);
    }
    // (2) wave-contiguous tiles (pitch = wave span) and plain rows
    for (int parts : {1, 6}) {
        run_shape<1, BUFFER>(buf, bytes, W, D, 256, parts, 0); run_shape<2, BUFFER>(buf, bytes, W, D, 512, parts, 0); run_shape<4, BUFFER>(buf, bytes, W, D, 1024, parts, 0);
        run_shape<2, BUFFER>(buf, bytes, W, D, 1 << 20, parts, 0);
    }
    // (3) occupancy caps
    for (int wpc : {1, 2, 4, 8}) for (int parts : {1, 6, 12}) {
        run_shape<2, BUFFER>(buf, bytes, W, D, 4096, parts, wpc);
        run_shape<4, BUFFER>(buf, bytes, W, D, 4096, parts, wpc);
    }
    // (4) sweep shapes
    for (int64_t pitch : {256, 512, 1024, 4096, 16384, 1 << 20}) {
        run_sweep<1, BUFFER, false>(buf, par, bytes, W, D, pitch); run_sweep<2, BUFFER, false>(buf, par, bytes, W, D, pitch); run_sweep<4, BUFFER, false>(buf, par, bytes, W, D, pitch);
        run_sweep<1, GLOBAL, false>(buf, par, bytes, W, D, pitch); run_sweep<2, FLAT, false>(buf, par, bytes, W, D, pitch);
        run_sweep<1, BUFFER, true>(buf, par, bytes, W, D, pitch); run_sweep<2, BUFFER, true>(buf, par, bytes, W, D, pitch); run_sweep<4, BUFFER, true>(buf, par, bytes, W, D, pitch);
    }
    return 0;
}
