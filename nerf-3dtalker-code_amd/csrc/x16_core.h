// Shared machinery of the 16-bit MFMA kernels (fused render forward, fused training forward / backward):
// fragment traits, the LDS weight stream, the fast positional encoding and small compile-time helpers.
#pragma once
#include <stdlib.h>

#include <type_traits>

#include "n3dt_device.h"
#include "n3dt_layout.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define X16_BS 32
#define X16_CH 24                      // pieces per chunk (24 KiB)
#define X16_PIECE 1024                 // bytes
#define X16_NCHUNK (2280 / X16_CH)     // 48 + 6*288 + 336 + 24 + 144 pieces (RGB_layer_0 is merged into RGB_layer_1)
#define X16_XT_TILES 98                // saved activation tiles per block (training): PE 2 + 8 hidden layers x 12
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
    static __device__ __forceinline__ f32x4 mfma16(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
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
    static __device__ __forceinline__ f32x4 mfma16(frag a, frag b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
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
//   (a half-chunk stagger of the SIMD partner waves and an epilogue deferred onto a second accumulator set were
//   measured perf-neutral or worse in round 1 -- DESIGN 3.1 log items 3, 8, 11 -- and are gone from the source.)
#ifndef X16_DEPTH
#define X16_DEPTH 3  // fragments in flight per wave: 3 measured 0.6 % faster than 2 for the render kernel (same box), 4 spills it
#endif
#ifndef X16_DEFAULT_TILING
#define X16_DEFAULT_TILING 1  // 1: 8 waves x 32 samples, 2: 4 waves x 64 samples (N3DT_X16_TILING overrides at run time)
#endif
#ifndef X16_NBUF
#define X16_NBUF 3  // chunk buffers of the ring (4 measured no faster for the render kernel and the renderer blocks)
#endif
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

// compile-time loop: f(std::integral_constant<int, I>) for I in [0, N) -- the piece index has to reach the
// inline-asm immediates as a constant expression
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Keeping scalar loads out of the stream.  The counted fragment waits (s_waitcnt lgkmcnt(DEPTH-1)) share their counter
// with scalar memory loads, which return out of order.  That cannot make a wait pass EARLY -- the awaited fragment is the
// oldest LDS operation in flight and LDS returns in order, so while it is outstanding the younger prefetch behind it is
// too and the counter stays above DEPTH-1 whatever the scalar loads do -- but every stray s_load makes the wait longer
// (it has to drain as well) and costs an issue slot in the MFMA stream.  hipcc fetches kernel arguments lazily, at their
// first use, which can be deep inside the stream (seen: the output pointer of the fused renderer block, two arguments of
// the camera-gradient backward kernel).  Such kernels pass what they need later through x16_pin() before
// WeightStream::prologue_wait(): the empty asm forces the load up there and its opaque result cannot be re-fetched from
// the argument segment.  tools/check_smem_hazard.py scans the generated code for scalar / flat accesses inside streams.
// (pointers travel through the asm as integers and come back as global-address-space pointers: an opaque generic
// pointer would turn every access through it into a flat_ instruction, which counts on BOTH wait counters)
template <class T>
__device__ __forceinline__ void x16_pin(T& v) {
    static_assert(sizeof(T) <= 8, "scalar register pair at most");
    asm volatile("" : "+s"(v));
}
template <class T>
__device__ __forceinline__ void x16_pin(T*& p) {
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+s"(v));
    p = (T*)(GLOBAL_AS T*)v;
}
template <class T>
__device__ __forceinline__ void x16_pin_v(T*& p) {  // per-lane pointers
    unsigned long long v = (unsigned long long)p;
    asm volatile("" : "+v"(v));
    p = (T*)(GLOBAL_AS T*)v;
}

#ifndef X16_RV_RAW
#define X16_RV_RAW 0
#endif
#ifndef X16_NT_STORE
#define X16_NT_STORE 1  // saved tiles are streamed (written once, read by a later kernel): nontemporal stores (fwd 1.65 -> 1.35 ms)
#endif
// DEPTH: fragments in flight per wave (DEPTH; the training forward runs with 2 to stay inside its register budget)
template <int PREC, int WAVES, int NCHUNK = X16_NCHUNK, int NBUF = X16_NBUF, int DEPTH = X16_DEPTH>
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
    // The stream can wrap around (prologue_issue(passes)): a persistent workgroup that walks several tiles keeps it running
    // across them.  Tried for the render kernel and not kept: with the tile body inlined into a tile loop hipcc hoists the
    // encoder's per-lane selectors and address constants out of the loop and spills (140 SGPR + 80 VGPR: 11.5 ms against
    // 7.2); with the body as a noinline callee (uniform arguments moved to SGPRs on entry) it still spills 15 fragments
    // (9.8 ms) -- the one-tile kernel sits at 254 of 256 registers and the loop adds a handful.  Results were correct.
    int limit;                  // chunks to stage in all: passes * NCHUNK
    int src_c;                  // source chunk of the next issue (wraps at NCHUNK: every pass streams the same weights)
    frag a[DEPTH];          // piece p sits in a[p % DEPTH]
    static constexpr int PPW = X16_CH / WAVES;  // pieces each wave stages per chunk

    __device__ __forceinline__ void issue(int c) {  // c: chunk count since the prologue (picks the ring buffer)
        const unsigned char* src = gsrc + (size_t)src_c * X16_CHUNK_BYTES;
        src_c = src_c + 1 == NCHUNK ? 0 : src_c + 1;
        unsigned char* dst = ring + (c % NBUF) * X16_CHUNK_BYTES + wave * PPW * X16_PIECE;
        // (the instruction's immediate offset is not used: with one M0 and offsets 0 / 1 KiB / 2 KiB, and with a per-piece
        // M0 plus the offset on top, the fused kernel's results were wrong -- its effect on the LDS address was not pinned down)
#pragma unroll
        for (int i = 0; i < PPW; ++i)
            __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + i * X16_PIECE), (LDS_AS void*)(dst + i * X16_PIECE), 16, 0,
                                             0);
    }
    __device__ __forceinline__ void prologue_issue(const int passes = 1) {
        limit = passes * NCHUNK;
        src_c = 0;
#pragma unroll
        for (int i = 0; i < NBUF - 1; ++i)
            if (i < limit) issue(i);
    }
    // a persistent workgroup is done: nothing of the stream may still be in flight towards its LDS when it exits
    __device__ __forceinline__ void drain() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }
    __device__ __forceinline__ void prologue_wait() {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        chunk = 0;
        meets = 0;
        vm_cur = 0;
        vm_prev = 0;
        cur_addr = lds_addr0;
        nxt_addr = lds_addr0 + X16_CHUNK_BYTES;
        preload<0>();
    }
    template <int J>
    __device__ __forceinline__ void preload() {
        if constexpr (J < DEPTH - 1) {
            read_frag<J * X16_PIECE>(a[J], cur_addr);
            preload<J + 1>();
        }
    }
    // The wait of a rendezvous is COUNTED.  Chunk meets+1 was issued NBUF-2 rendezvous ago; vector-memory operations retire in
    // order (loads and stores share vmcnt on gfx9-family parts), so "at most K operations outstanding", K = what this wave has
    // issued SINCE that chunk's loads, means they have landed.  K = the (NBUF-3) newer chunks' loads plus the stores the kernel
    // reports through note_stores() -- the training kernels write 2-3 KiB per tile, and with a plain vmcnt(0) every rendezvous
    // also waited for the acknowledgement of the stores issued a tile ago (the write path runs near the HBM write bandwidth
    // there: the forward with SAVE took 1.58 ms against 0.92 ms without).  Safety: K must never exceed the true count, so only
    // stores that are issued UNCONDITIONALLY at their program point may be reported (dead waves write a dump record); loads and
    // stores the kernel does not report only make the wait stricter.  All of this folds at compile time (straight-line code).
    // THE COUNTED WAIT IS LOAD-BEARING IN EVERY BUILD.  The shipped rendezvous (X16_RV_RAW = 0) follows it with __syncthreads(),
    // and it is tempting to think that call's workgroup-scope release fence waits vmcnt(0) anyway (round 2's note said so, an
    // advisor repeated it, and early in round 3 the counted wait was compiled out of the shipped build on that belief).  It does
    // not: outside thread-group-split mode hipcc emits only the lgkmcnt part for that fence -- the shipped inference kernels show
    // `s_waitcnt lgkmcnt(..); s_barrier` with no vmcnt wait at 138 of 480 rendezvous (build/nerf_fwd_x16.s) -- so without the
    // wait below nothing orders a wave's LDS-DMA pieces of chunk meets+1 before the other waves read them.  The race almost
    // never shows (the chunk was issued two rendezvous earlier) and then shows as a parity failure once in a few hundred runs
    // (test_split_precision_mode_properties_at_full_size: 7.7e-4 instead of 5e-5).  tests/test_host_cpu.py now checks the
    // assembly: every rendezvous barrier of a stream kernel has a vmcnt wait in front of it.
    // -DX16_RV_RAW=1 (diagnostic) replaces __syncthreads() by a raw s_barrier between compiler barriers: what the barrier orders
    // needs no fence -- this wave's pieces of chunk meets+1 are awaited just above, and a wave's reads of chunk meets-1 (whose
    // buffer is overwritten next) were consumed by MFMAs a chunk ago.
    int vm_cur = 0, vm_prev = 0;  // reported stores since the last rendezvous / in the period before it
    __device__ __forceinline__ void note_stores(const int n) { vm_cur += n; }
    template <int K>
    __device__ __forceinline__ static void wait_vm() {
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(K) : "memory");
    }
    __device__ __forceinline__ void rendezvous() {
        int newer = limit - 2 - meets;  // chunks issued after chunk meets+1: meets+2 .. min(limit-1, meets+NBUF-2)
        newer = newer < 0 ? 0 : (newer > NBUF - 3 ? NBUF - 3 : newer);
        int k = newer * PPW + vm_cur + (NBUF > 3 ? vm_prev : 0);
        if (k > 15) k = 15;
        static_for<0, 16>([&](auto k_c) {
            constexpr int K = decltype(k_c)::value;
            if (k == K) wait_vm<K>();  // this wave's pieces of chunk meets+1 have landed
        });
#if X16_RV_RAW
        __builtin_amdgcn_s_barrier();  // everyone's have; everyone has left chunk meets-1
        asm volatile("" ::: "memory");
#else
        __syncthreads();
#endif
#ifndef X16_NODMA  // diagnostic build: the stream stops after the prologue (results are garbage, the timing is the point)
        if (meets + NBUF - 1 < limit) issue(meets + NBUF - 1);
#endif
        vm_prev = vm_cur;
        vm_cur = 0;
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
    template <int YOUNGER = DEPTH - 1>  // fragment reads issued after this one that may still be in flight
    __device__ __forceinline__ void await_frag(frag& f) {
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(f) : "i"(YOUNGER));
    }
    // Fragment of stage-local piece P (stages are whole chunks, so P % X16_CH and P % DEPTH equal their
    // stream-global values).  LAST/NP: the final stage must not prefetch past the end of the stream.
    template <bool LAST, int NP, int P>
    __device__ __forceinline__ frag next() {
        constexpr int rv = X16_CH - DEPTH;
        if (P % X16_CH == rv) {
            X16_T(const unsigned long long r0 = x16_now();)
            rendezvous();
            X16_T(t_rv += x16_now() - r0;)
        }
        constexpr int Q = P + DEPTH - 1;
#ifdef X16_NOLDS  // diagnostic build: no fragment reads at all (garbage results)
        if ((P + 1) % X16_CH == 0) ++chunk;
        return a[0];
#endif
        if (!(LAST && Q >= NP)) {
            const unsigned addr = (Q % X16_CH < P % X16_CH) ? nxt_addr : cur_addr;
            read_frag<(Q % X16_CH) * X16_PIECE>(a[Q % DEPTH], addr);
        }
        // at the stream's tail fewer reads are in flight behind this one: the count shrinks with them
        constexpr int younger = (LAST && NP - 1 - P < DEPTH - 1) ? NP - 1 - P : DEPTH - 1;
        await_frag<younger>(a[P % DEPTH]);
        const frag r = a[P % DEPTH];
        if ((P + 1) % X16_CH == 0) {
            ++chunk;
            cur_addr = nxt_addr;
            nxt_addr = lds_addr0 + ((chunk + 1) % NBUF) * X16_CHUNK_BYTES;
        }
        return r;
    }
    // A prefetch that nothing consumes must not be issued (LAST): its destination registers are dead to the compiler, which
    // hands them to other values while the read is still in flight -- the data then lands on top of them (seen in the
    // fused renderer block: store addresses overwritten by weight bytes, i.e. wild stores).  Where the code cannot know at
    // compile time whether the stream continues (a rolled loop over passes: the value is loop-carried, so its registers
    // stay reserved inside the loop), settle() after the loop retires the last prefetch before the registers are reused.
    // A RUN-TIME condition around read_frag / await_frag is not an option: the branch makes the fragment a phi and hipcc
    // copies the (not yet landed) registers ahead of the wait.
    __device__ __forceinline__ void settle() {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]));
#pragma unroll
        for (int i = 1; i < DEPTH; ++i) asm volatile("" : "+v"(a[i]));  // every slot stays reserved up to the wait
    }
};

// LeakyReLU(0.2) = max(a, 0.2 a) in two instructions.  As fmaxf() hipcc emits three: IEEE maxnum wants canonical inputs, and an
// accumulator read is not known to be one, so a v_max_f32 a, a, a goes in front (768 v_max for 384 activations in the C = 256
// renderer block).  A NaN propagates through v_max_f32 as through the three-instruction form, quieted.
__device__ __forceinline__ float x16_lrelu02(const float a) {
    const float b = 0.2f * a;
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// one float from LDS, issued from inline asm and NOT awaited: LDS returns in order, so the value has landed once a fragment read
// issued after it has been awaited (the caller then pins it with asm volatile("" : "+v"(dst)) before the first use)
template <int OFF>
__device__ __forceinline__ void x16_bias_read(float& dst, const unsigned addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF) : "memory");
}

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

// ---------------------------------------------------------------------------------------------
// Saved tiles of the training path: a [sample][channel] image that the weight-gradient kernel reads TRANSPOSED.
//
// A 32x32 tile in "lane = sample" form is two fragments f0, f1: lane (c, h) element j of f_s holds
// X[channel 16s + 8(j>>2) + 4h + (j&3)][sample c] (the B-operand form the fused kernels keep activations in).
// The weight gradients dW = dZ^T X sum over SAMPLES, so their MFMA operands need "lane = channel" fragments (8 samples of
// one channel per lane).  The first version transposed every tile on the matrix pipe before storing it (two selection MFMAs,
// 8 v_cvt_pk, the MFMA -> VALU latency: 8 % more MFMAs in the training forward and in the dX chain).  gfx950 transposes on
// the way OUT of LDS instead (ds_read_b64_tr_b16), so the producers now store their fragments as they are:
//     tile image (2 KiB) = [32 samples][32 channels] 16-bit, 64-byte rows;  lane (c, h) writes f_s at 64 c + 32 s + 16 h
// (row c = [ch 0-3, 8-11, 4-7, 12-15 | ch 16-19, 24-27, 20-23, 28-31]).  dw_x16_body moves the image to LDS as it is (two 1 KiB
// LDS-DMA pieces) and reads it with x16_tr_frag below.
// ---------------------------------------------------------------------------------------------
// dst: per-lane pointer = tile + 64 * (lane & 31) + 16 * (lane >> 5)   (x16_image_lane_offset)
__device__ __forceinline__ int x16_image_lane_offset(const int lane) { return 64 * (lane & 31) + 16 * (lane >> 5); }
template <int PREC>
__device__ __forceinline__ void x16_image_store(const typename X16<PREC>::frag f0, const typename X16<PREC>::frag f1, unsigned char* dst) {
    typedef typename X16<PREC>::frag frag;
#if X16_NT_STORE
    __builtin_nontemporal_store(f0, reinterpret_cast<frag*>(dst));
    __builtin_nontemporal_store(f1, reinterpret_cast<frag*>(dst + 32));
#else
    *reinterpret_cast<frag*>(dst) = f0;
    *reinterpret_cast<frag*>(dst + 32) = f1;
#endif
}
// Per-lane LDS byte offset of the transposed reads inside a tile image: lane (r, h) of the 32x32x16 operand takes channel r,
// samples 16 s' + 8 h + 0..7 as two 4-sample blocks.  In a group of 16 lanes, lane 4q + p supplies the address of sample row q,
// channels 4p .. 4p+3 of the group's 16 (the 8-byte unit (p & 1) * 2 + (p >> 1) of the row's 32 bytes).  A 32-lane half covers
// 4 rows x 64 bytes = all 64 banks once: conflict-free.  Checked against exact integer data on the GPU (tools/tr_probe.hip).
__device__ __forceinline__ unsigned x16_tr_lane_offset(const int lane) {
    const int r = lane & 31, h = lane >> 5, q = (r & 15) >> 2, p = r & 3;
    return 64u * (8 * h + q) + 32u * (r >> 4) + 8u * ((p & 1) * 2 + (p >> 1));
}
// The same for an image whose rows hold the 32 channels in NATURAL order (row-major maps of the 2-D renderer moved to LDS as
// they are, train_x16.inc: rowmajor): channels 4p .. 4p+3 are the p-th 8-byte unit.
__device__ __forceinline__ unsigned x16_tr_lane_offset_natural(const int lane) {
    const int r = lane & 31, h = lane >> 5, q = (r & 15) >> 2, p = r & 3;
    return 64u * (8 * h + q) + 32u * (r >> 4) + 8u * p;
}
// The reads are issued from inline asm (x16_tr_issue) and retired by ONE s_waitcnt (x16_tr_settle) that names every destination:
// through the builtin hipcc orders them behind the LDS-DMA stream in flight (s_waitcnt vmcnt(0) in front of the first read of
// every stage -- the stages still loading are then waited for as well, and the pipeline is one stage deep).
typedef unsigned x16_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned x16_u32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void x16_tr_issue(x16_u32x2& d, const unsigned lds_addr) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(lds_addr), "i"(OFF) : "memory");
}
// four 64-bit reads = the two k-steps of one 32 x 32 tile image at LDS byte address tile_lane (x16_tr_lane_offset included)
struct X16TrTile {
    x16_u32x2 q[4];  // k-step sp: q[2 sp] (samples +0..3), q[2 sp + 1] (samples +4..7)
    template <int TILE_OFF>
    __device__ __forceinline__ void issue(const unsigned tile_lane) {
        x16_tr_issue<TILE_OFF>(q[0], tile_lane);
        x16_tr_issue<TILE_OFF + 256>(q[1], tile_lane);
        x16_tr_issue<TILE_OFF + 1024>(q[2], tile_lane);
        x16_tr_issue<TILE_OFF + 1280>(q[3], tile_lane);
    }
    __device__ __forceinline__ void pin() {  // the destinations stay reserved up to the wait (x16_tr_settle)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(q[i]));
    }
    __device__ __forceinline__ bf16x8 frag(const int sp) const {
        const x16_u32x4 v = {q[2 * sp][0], q[2 * sp][1], q[2 * sp + 1][0], q[2 * sp + 1][1]};
        return __builtin_bit_cast(bf16x8, v);
    }
};
__device__ __forceinline__ void x16_tr_settle() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
