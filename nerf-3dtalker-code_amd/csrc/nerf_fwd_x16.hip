// Fused volumetric-render kernel, 16-bit MFMA mode (bf16 or f16 inputs, fp32 accumulate):
// the roofline path.
//
// One wavefront carries NB blocks of 32 consecutive samples of a ray (NB = 2 and N_s = 64: one
// wavefront per ray) through the whole MLP with v_mfma_f32_32x32x16_{bf16,f16}.  Activations are
// kept TRANSPOSED, H^T [channel][sample]: the sample sits on the MFMA column (lane & 31), the
// channels in the accumulator registers.  A 32x32 accumulator tile, ReLU'd and packed to 16 bit,
// is then directly the B operand of the next layer's MFMAs (its k order is a fixed permutation,
// matched by the weight packing, cdna guide section 3), so activations never leave the register
// file and LDS carries only weights.
//
// Weights: one flat stream of 1 KiB pieces (one MFMA A fragment each, lane-linear) in execution
// order.  All waves of the workgroup consume the same stream; it is staged L2 -> LDS by LDS-DMA
// (global_load_lds_dwordx4) in 24-piece chunks, double buffered, one workgroup barrier per chunk.
// Every fragment read is a conflict-free lane-linear ds_read_b128.
//
// Epilogue per 32-sample block: density -> alpha -> in-block transmittance scan over the 32
// lanes -> weights; the RGB_layer_1 activations are weighted and reduced over the samples with a
// 5-step butterfly, so the only HBM traffic per block is one 196-float partial.
#include <stdlib.h>

#include <type_traits>

#include "n3dt_device.h"
#include "n3dt_layout.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define X16_BS 32
#define X16_CH 24                      // pieces per chunk (24 KiB)
#define X16_PIECE 1024                 // bytes
#define X16_NCHUNK (2280 / X16_CH)     // 48 + 6*288 + 336 + 24 + 144 pieces (RGB_layer_0 is merged into RGB_layer_1)
#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

template <int PREC>
struct X16;
template <>
struct X16<N3DT_BF16> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag pack(const float* v) {
        frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (__bf16)v[j];
        return f;
    }
    // ReLU on the packed 16-bit values: a signed 16-bit max with 0 (v_pk_max_i16) clears exactly
    // the negative floats, and rounding commutes with it
    // `lo` = 0: ReLU;  lo = -32768 (the most negative 16-bit pattern): identity -- lets one rolled loop body serve
    // both the ReLU layers and the linear RGB_layer_0
    static __device__ __forceinline__ frag relu(frag f, short lo) {
        s16x8 s = __builtin_bit_cast(s16x8, f);
        s = __builtin_elementwise_max(s, (s16x8)(lo));
        return __builtin_bit_cast(frag, s);
    }
    // bias as an MFMA: A = [b_hi, b_lo, 0...] on the k = 0, 1 slots (lanes of the lower half), B = [1, 1, 0...]
    static __device__ __forceinline__ frag bias_frag(float b, bool lower_half) {
        __bf16 hi = (__bf16)b;
        __bf16 lo = (__bf16)(b - (float)hi);
        frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (__bf16)0.0f;
        if (lower_half) {
            f[0] = hi;
            f[1] = lo;
        }
        return f;
    }
    static __device__ __forceinline__ frag ones_frag() {
        frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (__bf16)(j < 2 ? 1.0f : 0.0f);
        return f;
    }
};
template <>
struct X16<N3DT_F16> {
    typedef f16x8 frag;
    static __device__ __forceinline__ f32x16 mfma(frag a, frag b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ frag pack(const float* v) {
        frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (_Float16)v[j];
        return f;
    }
    // `lo` = 0: ReLU;  lo = -32768 (the most negative 16-bit pattern): identity -- lets one rolled loop body serve
    // both the ReLU layers and the linear RGB_layer_0
    static __device__ __forceinline__ frag relu(frag f, short lo) {
        s16x8 s = __builtin_bit_cast(s16x8, f);
        s = __builtin_elementwise_max(s, (s16x8)(lo));
        return __builtin_bit_cast(frag, s);
    }
    static __device__ __forceinline__ frag bias_frag(float b, bool lower_half) {
        _Float16 hi = (_Float16)b;
        _Float16 lo = (_Float16)(b - (float)hi);
        frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (_Float16)0.0f;
        if (lower_half) {
            f[0] = hi;
            f[1] = lo;
        }
        return f;
    }
    static __device__ __forceinline__ frag ones_frag() {
        frag f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (_Float16)(j < 2 ? 1.0f : 0.0f);
        return f;
    }
};

// The weight stream: X16_NCHUNK chunks of X16_CH pieces, staged L2 -> LDS by LDS-DMA into a ring of three
// chunk buffers (chunk c lives in buffer c % 3), and read by every wave through a small register ring of
// X16_DEPTH fragments so that the matrix pipe never waits for an LDS round trip:
//   * the prologue stages chunks 0 and 1 and meets once;
//   * rendezvous n (one workgroup barrier) is met X16_DEPTH pieces BEFORE the end of chunk n: it makes chunk
//     n+1 visible (issued at rendezvous n-1) and issues chunk n+2 into the buffer of chunk n-1, which every
//     wave has left by then -- so fragment prefetches run across chunk boundaries without a bubble;
//   * X16_STAGGER: waves WAVES/2.. (the SIMD partners of waves 0..) meet half a chunk earlier in THEIR stream,
//     i.e. they run half a chunk behind, so one wave's accumulator epilogue overlaps its partner's MFMAs.
#ifndef X16_STAGGER
#define X16_STAGGER 0
#endif
#ifndef X16_DEPTH
#define X16_DEPTH 2
#endif
#ifndef X16_DEFER
#define X16_DEFER 0
#endif
#ifndef X16_DEFAULT_TILING
#define X16_DEFAULT_TILING 1  // 1: 8 waves x 32 samples, 2: 4 waves x 64 samples (N3DT_X16_TILING overrides at run time)
#endif
#define X16_NBUF 3
#define X16_CHUNK_BYTES (X16_CH * X16_PIECE)

// Diagnostic build only (-DX16_STAMP): per-wave cycle sums of the three phases of a tile, written to the
// `wlocal` debug buffer (never read by the library).  Not for timing the kernel: the stamps fence overlap.
#ifdef X16_STAMP
__device__ __forceinline__ unsigned long long x16_now() {
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define X16_T(x) x
#else
#define X16_T(x)
#endif

template <int PREC, int WAVES>
struct WeightStream {
    X16_T(unsigned long long t_rv = 0; unsigned long long t_mfma = 0; unsigned long long t_epi = 0; unsigned long long t_bias = 0;
          int tile_no = 0; float* tl = nullptr;)
    typedef typename X16<PREC>::frag frag;
    const unsigned char* gsrc;  // per-lane: packed + (wave*PPW)*1KiB + lane*16
    unsigned char* ring;        // LDS, 3 chunk buffers
    unsigned lds_addr0;         // LDS byte address of ring + lane*16
    unsigned cur_addr, nxt_addr;  // LDS byte addresses (+ lane*16) of the buffers of chunk `chunk` and `chunk`+1
    int chunk;                  // chunk of the piece being consumed
    int meets;                  // rendezvous done so far
    int wave;
    frag a[X16_DEPTH];          // piece p sits in a[p % X16_DEPTH]
    static constexpr int PPW = X16_CH / WAVES;  // pieces each wave stages per chunk

    __device__ __forceinline__ void issue(int c) {
        const unsigned char* src = gsrc + (size_t)c * X16_CHUNK_BYTES;
        unsigned char* dst = ring + (c % X16_NBUF) * X16_CHUNK_BYTES + wave * PPW * X16_PIECE;
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + i * X16_PIECE), (LDS_AS void*)(dst + i * X16_PIECE), 16, 0,
                                             0);
    }
    __device__ __forceinline__ void prologue_issue() {
        issue(0);
        issue(1);
    }
    __device__ __forceinline__ void prologue_wait() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        chunk = 0;
        meets = 0;
        cur_addr = lds_addr0;
        nxt_addr = lds_addr0 + X16_CHUNK_BYTES;
        preload<0>();
    }
    template <int J>
    __device__ __forceinline__ void preload() {
        if constexpr (J < X16_DEPTH - 1) {
            read_frag<J * X16_PIECE>(a[J], cur_addr);
            preload<J + 1>();
        }
    }
    __device__ __forceinline__ void rendezvous() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of chunk meets+1 have landed
        __syncthreads();                                   // everyone's have; everyone has left chunk meets-1
        if (meets + 2 < X16_NCHUNK) issue(meets + 2);
        ++meets;
    }
    // The fragment reads are issued from inline asm so that their completion can be awaited with a COUNTED
    // s_waitcnt lgkmcnt(DEPTH-1): hipcc's own bookkeeping waits lgkmcnt(0) here, i.e. for the prefetch it has just
    // issued, which puts a full LDS round trip in front of every other MFMA (45 % of the wave time parked).
    // LDS returns in order, so "at most DEPTH-1 younger operations outstanding" means this piece has landed; younger
    // compiler-issued LDS operations only make the wait more conservative.  The wait names the fragment as "+v", so
    // the consuming MFMA cannot be scheduled above it.
    template <int OFF>
    __device__ __forceinline__ void read_frag(frag& dst, const unsigned addr) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
    }
    __device__ __forceinline__ void await_frag(frag& f) {
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "i"(X16_DEPTH - 1));
    }
    // Fragment of stage-local piece P (stages are whole chunks, so P % X16_CH and P % X16_DEPTH equal their
    // stream-global values).  LAST/NP: the final stage must not prefetch past the end of the stream.
    template <bool LATE, bool LAST, int NP, int P>
    __device__ __forceinline__ frag next() {
        constexpr int rv = (LATE ? X16_CH / 2 : X16_CH) - X16_DEPTH;
        if (P % X16_CH == rv) {
            X16_T(const unsigned long long r0 = x16_now();)
            rendezvous();
            X16_T(t_rv += x16_now() - r0;)
        }
        constexpr int Q = P + X16_DEPTH - 1;
        if (!(LAST && Q >= NP)) {
            const unsigned addr = (Q % X16_CH < P % X16_CH) ? nxt_addr : cur_addr;
            read_frag<(Q % X16_CH) * X16_PIECE>(a[Q % X16_DEPTH], addr);
        }
        await_frag(a[P % X16_DEPTH]);
        const frag r = a[P % X16_DEPTH];
        if ((P + 1) % X16_CH == 0) {
            ++chunk;
            cur_addr = nxt_addr;
            nxt_addr = lds_addr0 + ((chunk + 1) % X16_NBUF) * X16_CHUNK_BYTES;
        }
        return r;
    }
};

// positional-encoding channel `ch` (0..63) via v_sin_f32 on a two-term phase in revolutions.  The three
// coordinates travel as separate scalars and are picked with selects: with float[3] arguments hipcc turned the
// runtime `dim` into an index into a private-memory copy (28 B/lane of scratch, 117 MB of traffic per launch).
__device__ __forceinline__ float pick3(const int d, const float a, const float b, const float c) {
    const float ab = d == 0 ? a : b;
    return d == 2 ? c : ab;
}
__device__ __forceinline__ float pe_fast(const float p0, const float p1, const float p2, const float h0, const float h1,
                                         const float h2, const float l0, const float l1, const float l2, const int ch) {
    const int cc = ch < 3 ? 0 : ch - 3;
    const int k = cc / 6, w = cc % 6, dim = w >= 3 ? w - 3 : w;
    const float sc = (float)(1 << k);
    const float hi = pick3(dim, h0, h1, h2), lo = pick3(dim, l0, l1, l2);
    const float r = __builtin_amdgcn_fractf(hi * sc) + lo * sc + (w >= 3 ? 0.25f : 0.0f);  // cos x = sin(x + pi/2)
    const float sv = __builtin_amdgcn_sinf(r);
    const float raw = pick3(ch, p0, p1, p2);
    return ch < 3 ? raw : (ch >= N3DT_PE_DIM ? 0.0f : sv);
}

enum { MODE_HIDDEN = 0, MODE_LINEAR = 1, MODE_DENSITY = 2, MODE_COMPOSITE = 3 };

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) -- the piece index has to reach the
// inline-asm immediates as a constant expression
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// sum over the 32 lanes of a half-wave of 32 per-lane values: lane c ends with value index rev5(c)
__device__ __forceinline__ float butterfly32(float (&v)[32], const int c) {
#pragma unroll
    for (int step = 0; step < 5; ++step) {
        const int m = 16 >> step;
        const bool bit = (c & m) != 0;
        const int n = 32 >> step;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            float keep = bit ? v[2 * i + 1] : v[2 * i];
            float send = bit ? v[2 * i] : v[2 * i + 1];
            v[i] = keep + __shfl_xor(send, m, 64);
        }
    }
    return v[0];
}

// One stage: out[N x 32*NB] = W'[N x K] . in[K x 32*NB] + bias (+ activation), NT = N/32 out tiles.
// The KPE leading k-steps take their B operand from the wave's PE fragments: registers (pe_reg,
// stage L0) or the wave's LDS copy (pe_lds, skip stage L5); the rest come from hin.
// `bias` is wave-uniform, so the 32 values of a tile arrive by scalar loads; lane half h picks
// rows (r&3)+8(r>>2)+4h of the tile.
template <int PREC, int NB, int WAVES, bool LATE, int KS, int KPE, int NT, int MODE>
__device__ __forceinline__ void x16_stage(WeightStream<PREC, WAVES>& ws, const float* __restrict__ bias,
                                          const typename X16<PREC>::frag (&pe_reg)[NB][4], const unsigned char* pe_lds,
                                          const typename X16<PREC>::frag (&hin)[NB][24], typename X16<PREC>::frag (&hout)[NB][24],
                                          float (&aux)[NB], float* const (&po)[NB], const bool (&live)[NB], const int lane,
                                          const short relu_lo = 0) {
    typedef typename X16<PREC>::frag frag;
    const int h = lane >> 5, c = lane & 31;
    float red[NB][32];
    const frag ones = X16<PREC>::ones_frag();
    float bias_cur = bias[c];
    // Two accumulator sets: tile ot accumulates into acc[ot & 1] while the epilogue of tile ot-1 (pack to 16 bit,
    // ReLU) is placed between this tile's MFMAs -- VALU work issues under the matrix pipe instead of after it.
    // X16_DEFER=1 measured perf-neutral (the wave is not VALU-bound in its epilogue) and costs 12 VGPRs: off by default
    constexpr bool PACKS = (MODE == MODE_HIDDEN || MODE == MODE_LINEAR);
    constexpr bool DEFER = PACKS && X16_DEFER;
    f32x16 acc[2][NB];
    auto finish_half = [&](const int t, const int half) {  // registers 8*half .. 8*half+7 of tile t -> k-step 2t+half
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = acc[DEFER ? (t & 1) : 0][nb][8 * half + r];
            frag f = X16<PREC>::pack(v);
            if (MODE == MODE_HIDDEN) f = X16<PREC>::relu(f, relu_lo);
            hout[nb][2 * t + half] = f;
        }
    };
    constexpr int E0 = KS >= 8 ? 2 : 1, E1 = KS >= 8 ? 6 : (KS - 1);
    static_for<0, NT>([&](auto ot_c) {
        constexpr int ot = decltype(ot_c)::value;
        constexpr int cur = DEFER ? (ot & 1) : 0;
        X16_T(const unsigned long long s0 = x16_now();)
        {
            // acc = bias, broadcast over the samples, by ONE extra MFMA (hi/lo split keeps ~16 mantissa bits):
            // lane r of the lower half holds bias[ot*32 + r], fetched one tile ahead
            const frag bf = X16<PREC>::bias_frag(bias_cur, h == 0);
            if (ot + 1 < NT) bias_cur = bias[(ot + 1) * 32 + c];
            f32x16 zero;
#pragma unroll
            for (int r = 0; r < 16; ++r) zero[r] = 0.0f;
            const f32x16 binit = X16<PREC>::mfma(bf, ones, zero);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[cur][nb] = binit;
        }
        X16_T(const unsigned long long s1 = x16_now(); const unsigned long long rv0 = ws.t_rv;)
        static_for<0, KS>([&](auto ks_c) {
            constexpr int ks = decltype(ks_c)::value;
            const frag a_cur = ws.template next<LATE, MODE == MODE_COMPOSITE, NT * KS, ot * KS + ks>();
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                frag b;
                if (ks < KPE) {
                    if (pe_lds) b = *reinterpret_cast<const frag*>(pe_lds + (nb * 4 + ks) * X16_PIECE);
                    else b = pe_reg[nb][ks < 4 ? ks : 0];
                } else {
                    b = hin[nb][ks >= KPE ? ks - KPE : 0];
                }
                acc[cur][nb] = X16<PREC>::mfma(a_cur, b, acc[cur][nb]);
            }
            if (DEFER && ot > 0 && ks == E0) finish_half(ot - 1, 0);
            if (DEFER && ot > 0 && ks == E1) finish_half(ot - 1, 1);
        });
        X16_T(const unsigned long long s2 = x16_now();)
        if (PACKS && (!DEFER || ot == NT - 1)) {
            finish_half(ot, 0);
            finish_half(ot, 1);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (MODE == MODE_DENSITY) {
                aux[nb] = acc[cur][nb][0];  // row 0 of the tile, valid on lanes with h == 0
            } else if (MODE == MODE_COMPOSITE) {
                // weighted RGB_layer_1 activations; two tiles (32 values) feed one butterfly over the samples
#pragma unroll
                for (int r = 0; r < 16; ++r) red[nb][(ot & 1) * 16 + r] = fmaxf(acc[cur][nb][r], 0.0f) * aux[nb];
                if (ot & 1) {
                    float s = butterfly32(red[nb], c);
                    // bit-reversed lane index = which of the 32 reduced values this lane ended up with
                    const int v = ((c & 1) << 4) | ((c & 2) << 2) | (c & 4) | ((c & 8) >> 2) | ((c & 16) >> 4);
                    const int reg = v & 15, tile = (ot - 1) + (v >> 4);
                    if (live[nb]) po[nb][tile * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h] = s;
                }
            }
        }
        X16_T(const unsigned long long s3 = x16_now(); ws.t_bias += s1 - s0; ws.t_mfma += (s2 - s1) - (ws.t_rv - rv0); ws.t_epi += s3 - s2;
              if (ws.tl && ws.tile_no >= 20 && ws.tile_no < 34 && (threadIdx.x & 63) == 0) {
                  ws.tl[4 + 2 * (ws.tile_no - 20)] = (float)(s1 & 0xFFFFFF);
                  ws.tl[5 + 2 * (ws.tile_no - 20)] = (float)(s2 & 0xFFFFFF);
              }
              ++ws.tile_no;)
    });
}

template <int PREC, int NB, int WAVES, bool LATE>
__device__ __forceinline__ void nerf_fwd_x16_body(
    const N3dtGeom& g, const unsigned char* __restrict__ packed, const float* __restrict__ fold, const float* __restrict__ xy,
    const float* __restrict__ R, const float* __restrict__ T, const float* __restrict__ Kinv, const float* __restrict__ t_rand,
    float* __restrict__ part, float* __restrict__ wlocal, int bpr, long total_blocks, unsigned char* lds, const int wave) {
    typedef typename X16<PREC>::frag frag;
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;

    WeightStream<PREC, WAVES> ws;
    ws.gsrc = packed + (size_t)wave * WeightStream<PREC, WAVES>::PPW * X16_PIECE + lane * 16;
    ws.ring = lds;
    ws.lds_addr0 = (unsigned)(size_t)(LDS_AS unsigned char*)lds + lane * 16;
    ws.wave = wave;
    ws.prologue_issue();  // the sampler / encoder below runs under these loads
    // per-wave LDS copy of the PE fragments for the skip stage: NB*4 lane-linear 1 KiB pieces
    unsigned char* pe_lds = lds + X16_NBUF * X16_CH * X16_PIECE + (size_t)wave * NB * 4 * X16_PIECE + lane * 16;

    // block bookkeeping: the wave handles NB consecutive 32-sample blocks (all of one frame, host-checked)
    bool live[NB];
    float* po[NB];
    long blk[NB];
    float dist[NB], zval[NB];
    frag pe[NB][4];
    int frame = 0;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        long bidx = ((long)blockIdx.x * WAVES + wave) * NB + nb;
        live[nb] = bidx < total_blocks;
        if (!live[nb]) bidx = total_blocks - 1;
        blk[nb] = bidx;
        po[nb] = part + (size_t)bidx * N3DT_PART_STRIDE;
        const int sb = (int)(bidx % bpr);
        const long rayg = bidx / bpr;
        const int ray = (int)(rayg % g.n_rays);
        const int b = (int)(rayg / g.n_rays);
        if (nb == 0) frame = b;
        float p[3];
        n3dt_sample_point(g, xy, R, T, Kinv, t_rand, b, ray, sb * X16_BS + c, p, dist[nb], zval[nb]);
        // phase in revolutions as hi + lo, so that the 2^k scaling of the encoder stays exact
        float rh[3], rl[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float inv2pi_hi = 0.15915494f, inv2pi_lo = 6.2195e-09f;  // 1/(2 pi) split
            rh[i] = p[i] * inv2pi_hi;
            rl[i] = fmaf(p[i], inv2pi_hi, -rh[i]) + p[i] * inv2pi_lo;
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                v[j] = pe_fast(p[0], p[1], p[2], rh[0], rh[1], rh[2], rl[0], rl[1], rl[2],
                               32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3));
            pe[nb][ks] = X16<PREC>::pack(v);
            *reinterpret_cast<frag*>(pe_lds + (nb * 4 + ks) * X16_PIECE) = pe[nb][ks];
        }
    }
    const float* fb = fold + (size_t)__builtin_amdgcn_readfirstlane(frame) * N3DT_FOLD_STRIDE;
    ws.prologue_wait();
    X16_T(if (wlocal && live[0]) ws.tl = wlocal + (size_t)blk[0] * X16_BS;)

    frag ha[NB][24], hb[NB][24];
    float aux[NB];
    // FeaExt_module_0 (reference: NetWorks/models.py:69-71)
    x16_stage<PREC, NB, WAVES, LATE, 4, 4, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(0), pe, nullptr, ha, ha, aux, po, live, lane);
    // FeaExt_module_1..7 with the skip concat after layer 4 (models.py:72-76).  Fully unrolled on purpose: rolling the
    // identical 384->384 layers into a loop (tried: one-layer body + register copy, two-layer ping-pong body) makes the
    // register allocator spill 120-270 VGPRs across the back edge and runs 1.7x slower.
    x16_stage<PREC, NB, WAVES, LATE, 24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(1), pe, nullptr, ha, hb, aux, po, live, lane);
    x16_stage<PREC, NB, WAVES, LATE, 24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(2), pe, nullptr, hb, ha, aux, po, live, lane);
    x16_stage<PREC, NB, WAVES, LATE, 24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(3), pe, nullptr, ha, hb, aux, po, live, lane);
    x16_stage<PREC, NB, WAVES, LATE, 24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(4), pe, nullptr, hb, ha, aux, po, live, lane);
    x16_stage<PREC, NB, WAVES, LATE, 28, 4, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(5), pe, pe_lds, ha, hb, aux, po, live, lane);
    x16_stage<PREC, NB, WAVES, LATE, 24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(6), pe, nullptr, hb, ha, aux, po, live, lane);
    x16_stage<PREC, NB, WAVES, LATE, 24, 0, 12, MODE_HIDDEN>(ws, fb + n3dt_bias_offset(7), pe, nullptr, ha, hb, aux, po, live, lane);
    // density head on h7 (models.py:78,84); the bias rides in the accumulator
    x16_stage<PREC, NB, WAVES, LATE, 24, 0, 1, MODE_DENSITY>(ws, fb + n3dt_bias_offset(8), pe, nullptr, hb, ha, aux, po, live, lane);
    // alpha, in-block transmittance and weights (reference: NetWorks/utils.py:273-289)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float sp = __shfl(aux[nb], c, 64);  // row 0 lives on the h == 0 half
        float sigma = fmaxf(sp, 0.0f);
        float alpha = 1.0f - expf(-sigma * dist[nb]);
        float x = 1.0f - alpha + 1e-10f;
        float Tl = n3dt_exclusive_prod<32>(x, c);
        float w = alpha * Tl;
        float s0 = w, s1 = w * zval[nb];
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) {
            s0 += __shfl_xor(s0, off, 32);
            s1 += __shfl_xor(s1, off, 32);
        }
        float tprod = __shfl(Tl * x, 31, 32);
        if (live[nb] && lane == 0) {
            po[nb][N3DT_G + 0] = s0;
            po[nb][N3DT_G + 1] = s1;
            po[nb][N3DT_G + 2] = tprod;
            po[nb][N3DT_G + 3] = 0.0f;
        }
#ifndef X16_STAMP
        if (live[nb] && wlocal && h == 0) wlocal[(size_t)blk[nb] * X16_BS + c] = w;
#endif
        aux[nb] = w;
    }
    // RGB_layer_0 -> RGB_layer_1 as ONE merged 192 x 384 layer on h7 (no activation sits between them, models.py:79-81;
    // merged matrix and bias built by pack / fold), relu, weighted by the sample weights and reduced over the samples
    x16_stage<PREC, NB, WAVES, LATE, 24, 0, 6, MODE_COMPOSITE>(ws, fb + n3dt_bias_offset(10), pe, nullptr, hb, ha, aux, po, live, lane);
#ifdef X16_STAMP
    if (wlocal && lane == 0 && live[0]) {
        float* dbg = wlocal + (size_t)blk[0] * X16_BS;
        dbg[0] = (float)ws.t_bias;
        dbg[1] = (float)ws.t_mfma;
        dbg[2] = (float)ws.t_epi;
        dbg[3] = (float)ws.t_rv;
    }
#endif
}

template <int PREC, int NB, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void nerf_fwd_x16_kernel(
    N3dtGeom g, const unsigned char* __restrict__ packed, const float* __restrict__ fold, const float* __restrict__ xy,
    const float* __restrict__ R, const float* __restrict__ T, const float* __restrict__ Kinv, const float* __restrict__ t_rand,
    float* __restrict__ part, float* __restrict__ wlocal, int bpr, long total_blocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // the two halves of the workgroup run the same stream half a chunk apart (see WeightStream)
    if (X16_STAGGER && wave >= WAVES / 2)
        nerf_fwd_x16_body<PREC, NB, WAVES, true>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, bpr, total_blocks, lds, wave);
    else
        nerf_fwd_x16_body<PREC, NB, WAVES, false>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, bpr, total_blocks, lds, wave);
}

template <int PREC, int NB, int WAVES>
static void launch_x16(const N3dtGeom* g, const void* packed, const float* fold, const float* xy, const float* R, const float* T,
                       const float* Kinv, const float* t_rand, float* part, float* wlocal, hipStream_t stream) {
    const int bpr = (g->n_samples + X16_BS - 1) / X16_BS;
    const long total = (long)g->batch * g->n_rays * bpr;
    const long per_wg = (long)WAVES * NB;
    const int grid = (int)((total + per_wg - 1) / per_wg);
    const size_t lds_bytes = X16_NBUF * X16_CH * X16_PIECE + (size_t)WAVES * NB * 4 * X16_PIECE;
    auto kern = nerf_fwd_x16_kernel<PREC, NB, WAVES>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, *g, reinterpret_cast<const unsigned char*>(packed),
                       fold, xy, R, T, Kinv, t_rand, part, wlocal, bpr, total);
}

extern "C" void n3dt_launch_nerf_fwd_x16(const N3dtGeom* g, int precision, const void* packed, const float* fold, const float* xy,
                                         const float* R, const float* T, const float* Kinv, const float* t_rand, float* part,
                                         float* wlocal, hipStream_t stream) {
    // two tilings of the same kernel: 8 waves x 32 samples (2 waves per SIMD, <= 256 registers) or 4 waves x 64 samples
    // (one wave per SIMD with the whole 512-register file: each weight fragment read from LDS feeds two MFMAs)
    static const int wide = [] {
        const char* e = getenv("N3DT_X16_TILING");
        return e ? atoi(e) : X16_DEFAULT_TILING;
    }();
    const long blocks = (long)g->batch * g->n_rays * ((g->n_samples + X16_BS - 1) / X16_BS);
    const bool use_wide = wide == 2 && (((long)g->n_rays * ((g->n_samples + X16_BS - 1) / X16_BS)) % 2 == 0) && blocks >= 2;
    if (precision == N3DT_BF16) {
        if (use_wide) launch_x16<N3DT_BF16, 2, 4>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, stream);
        else launch_x16<N3DT_BF16, 1, 8>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, stream);
    } else {
        if (use_wide) launch_x16<N3DT_F16, 2, 4>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, stream);
        else launch_x16<N3DT_F16, 1, 8>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, stream);
    }
}
