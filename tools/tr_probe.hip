// stand-alone check of the ds_read_b64_tr_b16 addressing used by dw_x16_body (tile image: [32 samples][32 channels], 64-byte rows,
// row c = fragment k-step 0's 32 bytes (lane halves h = 0, 1) then k-step 1's)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const short* frag_in /*[2][64][8]*/, short* out /*[2 ksteps][64 lanes][8]*/) {
    __shared__ __attribute__((aligned(256))) unsigned char lds[2048];
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    // the forward's store: fragment s of lane (c, h) at 64 c + 32 s + 16 h
    for (int s = 0; s < 2; ++s) *(s16x8*)(lds + 64 * c + 32 * s + 16 * h) = *(const s16x8*)(frag_in + (s * 64 + lane) * 8);
    __syncthreads();
    const int r = lane & 31, q = (r & 15) >> 2, p = r & 3, u = (p & 1) * 2 + (p >> 1);
    const unsigned base = 64 * (8 * h + q) + 32 * (r >> 4) + 8 * u;
    for (int sp = 0; sp < 2; ++sp) {
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + base + 1024 * sp));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + base + 1024 * sp + 256));
        s16x8 o = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        *(s16x8*)(out + (sp * 64 + lane) * 8) = o;
    }
}
int main() {
    // X[channel][sample] = channel * 32 + sample; fragment s, lane (c, h), element j = channel 16 s + 8 (j >> 2) + 4 h + (j & 3) of sample c
    std::vector<short> fin(2 * 64 * 8), fout(2 * 64 * 8);
    for (int s = 0; s < 2; ++s)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int c = lane & 31, h = lane >> 5, ch = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
                fin[(s * 64 + lane) * 8 + j] = (short)(ch * 32 + c);
            }
    short *di, *dout;
    hipMalloc(&di, fin.size() * 2); hipMalloc(&dout, fout.size() * 2);
    hipMemcpy(di, fin.data(), fin.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(fout.data(), dout, fout.size() * 2, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int sp = 0; sp < 2; ++sp)
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 8; ++j) {
                const int r = lane & 31, h = lane >> 5, sample = 16 * sp + 8 * h + j;
                const short want = (short)(r * 32 + sample), got = fout[(sp * 64 + lane) * 8 + j];
                if (want != got && bad++ < 10) printf("sp %d lane %d j %d: want %d got %d (ch %d sample %d)\n", sp, lane, j, want, got, got / 32, got % 32);
            }
    printf(bad ? "FAILED %d\n" : "tr addressing OK\n", bad);
    return bad != 0;
}
