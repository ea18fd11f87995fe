// HBM write bandwidth of the training forward's store pattern: every wave writes one 196 KiB record, 1 KiB per store instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int MODE>  // 0: 1 KiB contiguous per instruction, 1: 16 B per lane at 64-B stride (two instructions fill 2 KiB), 2: mode 0 nontemporal
__global__ __launch_bounds__(512) void wk(unsigned char* out, long recs, int kib, int spin) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    long rec = (long)blockIdx.x * 8 + wave;
    if (rec >= recs) return;
    unsigned char* p = out + rec * (size_t)kib * 1024;
    u32x4 v = {(unsigned)rec, (unsigned)lane, 1u, 2u};
    for (int i = 0; i < kib; i += 2) {
        if (MODE == 1) {
            unsigned char* q = p + (size_t)i * 1024 + 64 * (lane & 31) + 16 * (lane >> 5);
            *(u32x4*)q = v;
            *(u32x4*)(q + 32) = v;
        } else if (MODE == 2) {
            __builtin_nontemporal_store(v, (u32x4*)(p + (size_t)i * 1024 + lane * 16));
            __builtin_nontemporal_store(v, (u32x4*)(p + (size_t)(i + 1) * 1024 + lane * 16));
        } else {
            *(u32x4*)(p + (size_t)i * 1024 + lane * 16) = v;
            *(u32x4*)(p + (size_t)(i + 1) * 1024 + lane * 16) = v;
        }
        // idle time between tiles (the MFMA phase), in s_sleep units of 64 clocks
        for (int s = 0; s < spin; ++s) __builtin_amdgcn_s_sleep(1);
        v[2] += 1;
    }
}
int main(int argc, char** argv) {
    const long recs = 16384;
    const int kib = 196;
    unsigned char* d;
    if (hipMalloc(&d, recs * (size_t)kib * 1024) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int spin : {0, 4, 8, 16, 24}) {
        for (int mode = 0; mode < 3; ++mode) {
            float best = 1e9f;
            for (int it = 0; it < 4; ++it) {
                (void)hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(wk<0>, dim3(recs / 8), dim3(512), 0, 0, d, recs, kib, spin);
                if (mode == 1) hipLaunchKernelGGL(wk<1>, dim3(recs / 8), dim3(512), 0, 0, d, recs, kib, spin);
                if (mode == 2) hipLaunchKernelGGL(wk<2>, dim3(recs / 8), dim3(512), 0, 0, d, recs, kib, spin);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms; (void)hipEventElapsedTime(&ms, e0, e1);
                if (it && ms < best) best = ms;
            }
            printf("spin %2d mode %d: %.3f ms  %.2f TB/s\n", spin, mode, best, recs * (double)kib * 1024 / best / 1e9);
        }
    }
    return 0;
}
