// How fast can ONE CU take data into LDS?  One 512-thread workgroup per CU streams 1 KiB pieces (16 bytes per lane) from an
// L2-resident source into its LDS, D stages of P pieces per wave in flight:
//   mode 0   LDS-DMA: global_load_lds_dwordx4, counted vmcnt (what the weight-gradient and blur kernels use)
//   mode 1   register staging: global_load_dwordx4 into VGPRs, ds_write_b128 a stage later
//   mode 2   as 1, but the loads only (no LDS write): the vector-memory path alone
// Every wave owns its slots, there is no barrier: this is the raw intake rate, bytes per clock and CU.
// usage: hipcc -O3 --offload-arch=gfx950 tools/lds_fill_probe.hip -o /tmp/lds_fill_probe && /tmp/lds_fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))
#define P 5   // pieces per wave and stage (the weight-gradient kernel's)
#define NB 3  // stages in the ring

template <int MODE>
__global__ __launch_bounds__(512, 1) void fill(const unsigned char* __restrict__ src, const size_t region, const int iters, unsigned* __restrict__ sink,
                                               unsigned long long* __restrict__ cycles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const size_t off0 = (((size_t)blockIdx.x * 8 + wave) * (P * 1024)) % region;
    auto at = [&](const size_t k) { return src + (off0 + k * 8192) % region + lane * 16; };  // piece k of this wave
    unsigned char* my = lds + wave * (NB * P * 1024);
    unsigned long long t0 = __builtin_readcyclecounter();
    unsigned acc = 0;
    if (MODE == 0) {
        for (int s = 0; s < NB - 1; ++s)
#pragma unroll
            for (int p = 0; p < P; ++p)
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)at((size_t)(s * P + p)), (LDS_AS void*)(my + (s * P + p) * 1024), 16, 0, 0);
        for (int i = 0; i < iters; ++i) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"((NB - 2) * P) : "memory");
            const int sl = (i + NB - 1) % NB;
#pragma unroll
            for (int p = 0; p < P; ++p)
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)at((size_t)((i + NB - 1) * P + p)),
                                                 (LDS_AS void*)(my + (sl * P + p) * 1024), 16, 0, 0);
            acc += *reinterpret_cast<volatile unsigned*>(my + (i % NB) * P * 1024 + lane * 4);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        u32x4 ra[P], rb[P];
#pragma unroll
        for (int p = 0; p < P; ++p) ra[p] = *reinterpret_cast<const u32x4*>(at((size_t)p));
#pragma unroll
        for (int p = 0; p < P; ++p) rb[p] = *reinterpret_cast<const u32x4*>(at((size_t)(P + p)));
        for (int i = 0; i < iters; i += 2) {
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (MODE == 1) *reinterpret_cast<u32x4*>(my + ((i % NB) * P + p) * 1024 + lane * 16) = ra[p];
                else acc += ra[p][0];
                ra[p] = *reinterpret_cast<const u32x4*>(at((size_t)((i + 2) * P + p)));
            }
#pragma unroll
            for (int p = 0; p < P; ++p) {
                if (MODE == 1) *reinterpret_cast<u32x4*>(my + (((i + 1) % NB) * P + p) * 1024 + lane * 16) = rb[p];
                else acc += rb[p][0];
                rb[p] = *reinterpret_cast<const u32x4*>(at((size_t)((i + 3) * P + p)));
            }
            if (MODE == 1) acc += *reinterpret_cast<volatile unsigned*>(my + (i % NB) * P * 1024 + lane * 4);
        }
#pragma unroll
        for (int p = 0; p < P; ++p) acc += ra[p][1] + rb[p][1];
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (acc == 0x12345678u) sink[0] = acc;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
    const size_t region = (size_t)64 << 20;  // spread over the L2s / MALL; the stride of 8 KiB per piece keeps requests apart
    unsigned char* src;
    unsigned* sink;
    unsigned long long* cyc;
    if (hipMalloc(&src, region + (1 << 20)) != hipSuccess) return 1;
    (void)hipMemset(src, 1, region + (1 << 20));
    (void)hipMalloc(&sink, 64);
    (void)hipMalloc(&cyc, 256 * 8);
    const int iters = 2048;
    const size_t lds = 8 * NB * P * 1024;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        float best = 1e9f;
        for (int it = 0; it < 4; ++it) {
            (void)hipEventRecord(e0, 0);
            if (mode == 0) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fill<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(fill<0>, dim3(256), dim3(512), lds, 0, src, region, iters, sink, cyc);
            } else if (mode == 1) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fill<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(fill<1>, dim3(256), dim3(512), lds, 0, src, region, iters, sink, cyc);
            } else {
                (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fill<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
                hipLaunchKernelGGL(fill<2>, dim3(256), dim3(512), lds, 0, src, region, iters, sink, cyc);
            }
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            if (it && ms < best) best = ms;
        }
        unsigned long long h[256];
        (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < 256; ++i) mean += (double)h[i] / 256;
        const double bytes_cu = (double)iters * 8 * P * 1024;
        printf("mode %d: %.3f ms, %.2f TB/s over 256 CUs, %.0f cycles per workgroup -> %.1f B/clk/CU (%.0f cycles per 40-piece stage)\n", mode, best,
               bytes_cu * 256 / best / 1e9, mean, bytes_cu / mean, mean / iters);
    }
    return 0;
}
