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
#include "x16_core.h"


#ifndef X16_SAVE_DEPTH
#define X16_SAVE_DEPTH 3  // fragments in flight per wave in the training forward (X16_DEPTH for the others)
#endif
// One stage: out[N x 32*NB] = W'[N x K] . in[K x 32*NB] + bias (+ activation), NT = N/32 out tiles.
// The KPE leading k-steps take their B operand from the wave's PE fragments: registers (pe_reg,
// stage L0) or the wave's LDS copy (pe_lds, skip stage L5); the rest come from hin.
// `bias` is wave-uniform, so the 32 values of a tile arrive by scalar loads; lane half h picks
// rows (r&3)+8(r>>2)+4h of the tile.
// SAVE (training forward, NB == 1): every finished output tile is also written to HBM for the backward -- hidden
// tiles as [sample][channel] images (x16_core.h: x16_image_store) at sv.tile0 + 2 KiB * tile, the RGB_layer_1 activations
// lane-linear (lane = sample fragments) at sv.tile0 + 1 KiB * k-step.
template <int PREC>
struct X16SaveStage {
    unsigned char* tile0;  // per-lane pointer (image offset, or lane * 16 for the lane-linear fragments, included)
    unsigned* gate0;       // per-lane pointer to this layer's 6 gate words (64 words apart)
};

// Biases of the training forward come from LDS.  vmcnt retires in order, so hipcc's wait for a bias LOAD also waits for every
// store issued before it: with the inference form (a global load one tile ahead) each tile began by waiting for the previous
// tile's 2-3 KiB of saved-tile stores to be acknowledged, while the write path runs near the HBM write bandwidth -- 0.2 ms of
// the 1.65 ms kernel (diagnostic build without the loads).  Here every wave copies the next stage's bias table (<= 1.5 KiB) into
// its own LDS slot by LDS-DMA a stage ahead (X16BiasLds::stage_in; complete by the next rendezvous wait, which leaves only
// the stores issued since outstanding), and a tile's value is read with an asm ds_read_b32 issued a tile ahead: LDS returns in
// order, so it has landed once any later weight fragment has been awaited -- no wait of its own, nothing on vmcnt.
#define X16_BIAS_SLOT 1536  // bytes: 384 floats
#ifndef X16_SAVE_LDS_BIAS
#define X16_SAVE_LDS_BIAS 1
#endif
struct X16BiasLds {
    unsigned char* slots;  // this wave's two slots (generic pointer, wave-uniform)
    unsigned addr;         // LDS byte address of slot 0 + 4 * (lane & 31)
    // table -> slot by LDS-DMA: n floats (a multiple of 64), 256 B per instruction, lane-linear
    __device__ __forceinline__ void stage_in(const float* table, const int n, const int slot, const int lane) const {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (64 * i < n)
                __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(table + 64 * i + lane), (LDS_AS void*)(slots + slot * X16_BIAS_SLOT + 256 * i), 4, 0, 0);
    }
};

template <int PREC, int NB, int WAVES, int KS, int KPE, int NT, int MODE, bool SAVE = false, class WS>
__device__ __forceinline__ void x16_stage(WS& ws, const float* __restrict__ bias,
                                          const typename X16<PREC>::frag (&pe_reg)[NB][4], const unsigned char* pe_lds,
                                          const typename X16<PREC>::frag (&hin)[NB][24], typename X16<PREC>::frag (&hout)[NB][24],
                                          float (&aux)[NB], float* const (&po)[NB], const bool (&live)[NB], const int lane,
                                          const X16SaveStage<PREC>* sv = nullptr, const short relu_lo = 0,
                                          const unsigned bias_lds = 0 /* SAVE: LDS address of this stage's bias slot + 4 c */) {
    typedef typename X16<PREC>::frag frag;
    static_assert(!SAVE || NB == 1, "the training forward runs one block per wave");
    const int h = lane >> 5, c = lane & 31;
    float red[NB][32];
    unsigned gate_word = 0;
    const frag ones = X16<PREC>::ones_frag();
    float bias_cur = 0.0f, bias_nxt = 0.0f;
    constexpr bool LB = SAVE && X16_SAVE_LDS_BIAS;
    if constexpr (LB) {
        x16_bias_read<0>(bias_cur, bias_lds);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bias_cur));  // once per stage; the tiles' values travel a tile ahead
    } else {
        bias_cur = bias[c];
    }
    constexpr bool PACKS = (MODE == MODE_HIDDEN || MODE == MODE_LINEAR);
    f32x16 acc[1][NB];
    auto finish_half = [&](const int t, const int half) {  // registers 8*half .. 8*half+7 of tile t -> k-step 2t+half
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) v[r] = acc[0][nb][8 * half + r];
            frag f = X16<PREC>::pack(v);
            if (MODE == MODE_HIDDEN) f = X16<PREC>::relu(f, relu_lo);
            hout[nb][2 * t + half] = f;
        }
    };
    static_for<0, NT>([&](auto ot_c) {
        constexpr int ot = decltype(ot_c)::value;
        constexpr int cur = 0;
        X16_T(const unsigned long long s0 = x16_now();)
        {
            // acc = bias, broadcast over the samples, by ONE extra MFMA (hi/lo split keeps ~16 mantissa bits):
            // lane r of the lower half holds bias[ot*32 + r], fetched one tile ahead
            const frag bf = X16<PREC>::bias_frag(bias_cur, h == 0);
            if constexpr (LB) {
                if constexpr (ot + 1 < NT) x16_bias_read<(ot + 1) * 128>(bias_nxt, bias_lds);
            } else {
                if (ot + 1 < NT) bias_cur = bias[(ot + 1) * 32 + c];
            }
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
            const frag a_cur = ws.template next<MODE == MODE_COMPOSITE, NT * KS, ot * KS + ks>();
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
        });
        if constexpr (LB && ot + 1 < NT) {
            // landed: it is older than every fragment read awaited in the k-loop above (the accumulator rides along: the
            // statement then cannot move ahead of the loop's last MFMA)
            asm volatile("" : "+v"(bias_nxt), "+v"(acc[cur][0]));
            bias_cur = bias_nxt;
        }
        X16_T(const unsigned long long s2 = x16_now();)
        if (PACKS) {
            finish_half(ot, 0);
            finish_half(ot, 1);
        }
        if constexpr (SAVE && PACKS) {
            // (unconditional -- dead waves write a dump record -- and reported: the stream's rendezvous waits are counted)
#ifndef X16_DIAG_NOIMG
            x16_image_store<PREC>(hout[0][2 * ot], hout[0][2 * ot + 1], sv->tile0 + ot * 2 * X16_PIECE);
            ws.note_stores(2);
#endif
#ifndef X16_DIAG_NOGATE
            // ReLU gates of the backward chain: bit 8*half + j of this tile's half-word = "stored activation > 0" of accumulator
            // register 8*half + j (same lane, same tile, same register in nerf_bwd_x16_kernel); two tiles share a 32-bit word
            unsigned m = 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const s16x8 hv = __builtin_bit_cast(s16x8, hout[0][2 * ot + half]);
#pragma unroll
                for (int j = 0; j < 8; ++j) m |= (hv[j] != 0 ? 1u : 0u) << (8 * half + j);
            }
            // (v_pk_min_u16 + v_dot2_u32_u16 per packed pair would be 16 operations per tile instead of ~48, but hipcc expands
            // the packed min into compares and selects anyway and the dot2 result came out wrong on gfx950: not used)
            if constexpr (ot & 1) {
                __builtin_nontemporal_store(gate_word | (m << 16), sv->gate0 + (ot >> 1) * 64);
                ws.note_stores(1);
            } else {
                gate_word = m;
            }
#endif
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            if (MODE == MODE_DENSITY) {
                aux[nb] = acc[cur][nb][0];  // row 0 of the tile, valid on lanes with h == 0
            } else if (MODE == MODE_COMPOSITE) {
                // weighted RGB_layer_1 activations; two tiles (32 values) feed one butterfly over the samples
#pragma unroll
                for (int r = 0; r < 16; ++r) red[nb][(ot & 1) * 16 + r] = fmaxf(acc[cur][nb][r], 0.0f) * aux[nb];
                if constexpr (SAVE) {
#pragma unroll
                    for (int half = 0; half < 2; ++half) {
                        float v[8];
#pragma unroll
                        for (int r = 0; r < 8; ++r) v[r] = fmaxf(acc[cur][nb][8 * half + r], 0.0f);
                        __builtin_nontemporal_store(X16<PREC>::pack(v), reinterpret_cast<frag*>(sv->tile0 + (2 * ot + half) * X16_PIECE));
                    }
                    ws.note_stores(2);
                }
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

// What the training forward leaves in HBM per 32-sample block (bf16 fused training path, train_x16.hip):
//   xT   X16_XT_TILES tiles of 2 KiB: PE (2 tiles) | H0 .. H7 (12 tiles each), [sample][channel] images (x16_core.h)
//   gS   12 lane = sample fragments of relu(RGB_layer_1)
//   geo  density pre-activation [32] | plane distance [32]
//   gates  8 layers x 6 words x 64 lanes: the sign bits of H0 .. H7 in accumulator order (the dX chain's ReLU gates)
struct X16TrainSave {
    unsigned char* xT;
    unsigned char* gS;
    float* geo;
    unsigned* gates;
};

template <int PREC, int NB, int WAVES, bool SAVE = false>
__device__ __forceinline__ void nerf_fwd_x16_body(
    const N3dtGeom& g, const unsigned char* __restrict__ packed, const float* __restrict__ fold, const float* __restrict__ xy,
    const float* __restrict__ R, const float* __restrict__ T, const float* __restrict__ Kinv, const float* __restrict__ t_rand,
    float* __restrict__ part, float* __restrict__ wlocal, int bpr, long total_blocks, unsigned char* lds, const int wave,
    const X16TrainSave tsv = X16TrainSave{nullptr, nullptr, nullptr, nullptr}) {
    typedef typename X16<PREC>::frag frag;
    const int lane = threadIdx.x & 63;
    const int c = lane & 31, h = lane >> 5;

    // (the training forward's prefetch depth is its own switch: depth 2 frees four registers and measured the same, 1.32 against 1.32 ms)
    typedef WeightStream<PREC, WAVES, X16_NCHUNK, X16_NBUF, SAVE ? X16_SAVE_DEPTH : X16_DEPTH> WS;
    WS ws;
    ws.gsrc = packed + (size_t)wave * WS::PPW * X16_PIECE + lane * 16;
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
    long ray0 = 0;  // the global ray of the wave's first block (include_vd: NB = 1, so the wave's only ray)
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
        if (nb == 0) frame = b, ray0 = rayg;
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
    // include_vd: the merged RGB stage's bias is per RAY (frame entry + view-direction term; the table sits behind the fold table,
    // n3dt_layout.h).  The launcher picks the one-block-per-wave tiling then, so the wave has one ray.
    const float* b10 = fb + n3dt_bias_offset(10);
#if defined(X16_DIAG_VD_SELECT_ALL)  // diagnostic builds only (docs/tuning_log.md, round 4: tiling 2 miscompares with these)
#if defined(X16_DIAG_VD_TERNARY)
    b10 = g.vd_dim > 0 ? fold + n3dt_rayfold_offset(g.batch) + (size_t)__builtin_amdgcn_readfirstlane((int)ray0) * N3DT_RAYFOLD_STRIDE
                       : fb + n3dt_bias_offset(10);
#elif defined(X16_DIAG_VD_NO_RFL)
    if (g.vd_dim > 0) b10 = fold + n3dt_rayfold_offset(g.batch) + (size_t)ray0 * N3DT_RAYFOLD_STRIDE;
#else
    if (g.vd_dim > 0) b10 = fold + n3dt_rayfold_offset(g.batch) + (size_t)__builtin_amdgcn_readfirstlane((int)ray0) * N3DT_RAYFOLD_STRIDE;
#endif
#else
    if constexpr (NB == 1) {
        if (g.vd_dim > 0) b10 = fold + n3dt_rayfold_offset(g.batch) + (size_t)__builtin_amdgcn_readfirstlane((int)ray0) * N3DT_RAYFOLD_STRIDE;
    }
#endif
    X16SaveStage<PREC> svs;
    unsigned char* xT_blk = nullptr;  // this block's xT tiles (+ the lane's image offset)
    // record of this block in the saved buffers; a dead wave writes the dump record behind the last block (every wave must
    // issue the stores the stream's counted waits are told about, x16_core.h)
    const long rec = live[0] ? blk[0] : total_blocks;
    if constexpr (SAVE) {
        xT_blk = tsv.xT + (size_t)rec * X16_XT_TILES * 2 * X16_PIECE + x16_image_lane_offset(lane);
        x16_image_store<PREC>(pe[0][0], pe[0][1], xT_blk);
        x16_image_store<PREC>(pe[0][2], pe[0][3], xT_blk + 2 * X16_PIECE);
    }
    // output tile 0 of hidden layer l inside the block's xT record
    auto sv_hidden = [&](const int l) -> const X16SaveStage<PREC>* {
        if constexpr (SAVE) {
            svs.tile0 = xT_blk + (size_t)(2 + 12 * l) * 2 * X16_PIECE;
            svs.gate0 = tsv.gates + ((size_t)rec * 8 + l) * 6 * 64 + lane;
            return &svs;
        } else {
            return nullptr;
        }
    };
    // SAVE: per-wave bias slots in LDS, filled a stage ahead (X16BiasLds above); stage k reads slot k & 1
    X16BiasLds bl;
    if constexpr (SAVE && X16_SAVE_LDS_BIAS) {
        bl.slots = lds + X16_NBUF * X16_CH * X16_PIECE + (size_t)WAVES * NB * 4 * X16_PIECE + (size_t)wave * 2 * X16_BIAS_SLOT;
        bl.addr = (unsigned)(size_t)(LDS_AS unsigned char*)bl.slots + 4 * c;
        bl.stage_in(fb + n3dt_bias_offset(0), 384, 0, lane);
    }
    // the bias slot of SAVE stage number k (0 .. 8), after staging the table of the stage that follows it
    auto bias_slot = [&](const int k, const int next_table, const int next_n) -> unsigned {
        if constexpr (SAVE && X16_SAVE_LDS_BIAS) {
            if (next_table >= 0) bl.stage_in(next_table == 10 ? b10 : fb + n3dt_bias_offset(next_table), next_n, (k + 1) & 1, lane);
            return bl.addr + (k & 1) * X16_BIAS_SLOT;
        } else {
            return 0u;
        }
    };
    ws.prologue_wait();
    X16_T(if (wlocal && live[0]) ws.tl = wlocal + (size_t)blk[0] * X16_BS;)

    frag ha[NB][24], hb[NB][24];
    float aux[NB];
    // FeaExt_module_0 (reference: NetWorks/models.py:69-71)
    x16_stage<PREC, NB, WAVES, 4, 4, 12, MODE_HIDDEN, SAVE>(ws, fb + n3dt_bias_offset(0), pe, nullptr, ha, ha, aux, po, live, lane, sv_hidden(0), 0, bias_slot(0, 1, 384));
    // FeaExt_module_1..7 with the skip concat after layer 4 (models.py:72-76).  Fully unrolled on purpose: rolling the
    // identical 384->384 layers into a loop (tried: one-layer body + register copy, two-layer ping-pong body) makes the
    // register allocator spill 120-270 VGPRs across the back edge and runs 1.7x slower.
    x16_stage<PREC, NB, WAVES, 24, 0, 12, MODE_HIDDEN, SAVE>(ws, fb + n3dt_bias_offset(1), pe, nullptr, ha, hb, aux, po, live, lane, sv_hidden(1), 0, bias_slot(1, 2, 384));
    x16_stage<PREC, NB, WAVES, 24, 0, 12, MODE_HIDDEN, SAVE>(ws, fb + n3dt_bias_offset(2), pe, nullptr, hb, ha, aux, po, live, lane, sv_hidden(2), 0, bias_slot(2, 3, 384));
    x16_stage<PREC, NB, WAVES, 24, 0, 12, MODE_HIDDEN, SAVE>(ws, fb + n3dt_bias_offset(3), pe, nullptr, ha, hb, aux, po, live, lane, sv_hidden(3), 0, bias_slot(3, 4, 384));
    x16_stage<PREC, NB, WAVES, 24, 0, 12, MODE_HIDDEN, SAVE>(ws, fb + n3dt_bias_offset(4), pe, nullptr, hb, ha, aux, po, live, lane, sv_hidden(4), 0, bias_slot(4, 5, 384));
    x16_stage<PREC, NB, WAVES, 28, 4, 12, MODE_HIDDEN, SAVE>(ws, fb + n3dt_bias_offset(5), pe, pe_lds, ha, hb, aux, po, live, lane, sv_hidden(5), 0, bias_slot(5, 6, 384));
    x16_stage<PREC, NB, WAVES, 24, 0, 12, MODE_HIDDEN, SAVE>(ws, fb + n3dt_bias_offset(6), pe, nullptr, hb, ha, aux, po, live, lane, sv_hidden(6), 0, bias_slot(6, 7, 384));
    x16_stage<PREC, NB, WAVES, 24, 0, 12, MODE_HIDDEN, SAVE>(ws, fb + n3dt_bias_offset(7), pe, nullptr, ha, hb, aux, po, live, lane, sv_hidden(7), 0, bias_slot(7, 10, 192));
    // density head on h7 (models.py:78,84); the bias rides in the accumulator
    x16_stage<PREC, NB, WAVES, 24, 0, 1, MODE_DENSITY, false>(ws, fb + n3dt_bias_offset(8), pe, nullptr, hb, ha, aux, po, live, lane);
    // alpha, in-block transmittance and weights (reference: NetWorks/utils.py:273-289)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        float sp = __shfl(aux[nb], c, 64);  // row 0 lives on the h == 0 half
        if constexpr (SAVE) {
            if (live[nb] && h == 0) {
                tsv.geo[(size_t)blk[nb] * 64 + c] = sp;
                tsv.geo[(size_t)blk[nb] * 64 + 32 + c] = dist[nb];
            }
        }
        float sigma = fmaxf(sp, 0.0f);
#if defined(X16_DIAG_REMASK_DEAD_LANES)
        // diagnostic builds only (docs/tuning_log.md, round 4: the cause of tiling 2's garbage): the lanes past N_s get their
        // dist = 0 / zval = 0 again HERE, from the sample index, instead of trusting the registers that carried them from the
        // prologue -- hipcc parked those in AGPRs with a copy that ran under the sampler's reduced lane mask
        // (the values are made opaque first: the compiler KNOWS they are 0 in those lanes and folds a plain select away)
        asm volatile("" : "+v"(dist[nb]), "+v"(zval[nb]));
        if ((int)(blk[nb] % bpr) * X16_BS + c >= g.n_samples) dist[nb] = 0.0f, zval[nb] = 0.0f;
#endif
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
    if constexpr (SAVE) {
        svs.tile0 = tsv.gS + (size_t)rec * 12 * X16_PIECE + lane * 16;
        svs.gate0 = nullptr;
    }
    x16_stage<PREC, NB, WAVES, 24, 0, 6, MODE_COMPOSITE, SAVE>(ws, b10, pe, nullptr, hb, ha, aux, po, live, lane,
                                                                     SAVE ? &svs : nullptr, 0, bias_slot(8, -1, 0));
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
    nerf_fwd_x16_body<PREC, NB, WAVES>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, bpr, total_blocks, lds, wave);
}

// Training forward (bf16, one block per wave): the same body, leaving the activations behind (X16TrainSave)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 1) void nerf_fwd_x16_train_kernel(
    N3dtGeom g, const unsigned char* __restrict__ packed, const float* __restrict__ fold, const float* __restrict__ xy,
    const float* __restrict__ R, const float* __restrict__ T, const float* __restrict__ Kinv, const float* __restrict__ t_rand,
    float* __restrict__ part, float* __restrict__ wlocal, int bpr, long total_blocks, X16TrainSave tsv) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    nerf_fwd_x16_body<N3DT_BF16, 1, WAVES, true>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, bpr, total_blocks, lds, wave,
                                                        tsv);
}

#ifndef X16_TRAIN_WAVES
#define X16_TRAIN_WAVES 8
#endif
extern "C" void n3dt_launch_nerf_fwd_x16_train(const N3dtGeom* g, const void* packed, const float* fold, const float* xy, const float* R,
                                               const float* T, const float* Kinv, const float* t_rand, float* part, float* wlocal,
                                               void* xT, void* gS, float* geo, void* gates, hipStream_t stream) {
    constexpr int WAVES = X16_TRAIN_WAVES;
    const int bpr = (g->n_samples + X16_BS - 1) / X16_BS;
    const long total = (long)g->batch * g->n_rays * bpr;
    const int grid = (int)((total + WAVES - 1) / WAVES);
    const size_t lds_bytes = X16_NBUF * X16_CH * X16_PIECE + (size_t)WAVES * 4 * X16_PIECE + (size_t)WAVES * 2 * X16_BIAS_SLOT;
    auto kern = nerf_fwd_x16_train_kernel<WAVES>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    X16TrainSave tsv{reinterpret_cast<unsigned char*>(xT), reinterpret_cast<unsigned char*>(gS), geo, reinterpret_cast<unsigned*>(gates)};
    hipLaunchKernelGGL(kern, dim3(grid), dim3(WAVES * 64), lds_bytes, stream, *g, reinterpret_cast<const unsigned char*>(packed), fold, xy,
                       R, T, Kinv, t_rand, part, wlocal, bpr, total, tsv);
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
    // Tiling 2 is WITHDRAWN from the run-time switch (round 4): a diagnostic build whose only difference is how the RGB stage's
    // bias pointer is selected (-DX16_DIAG_VD_SELECT_ALL -DX16_DIAG_VD_TERNARY) miscompares on `tiny_train` in this tiling, with
    // results that change from run to run on one binary -- i.e. the 4-wave instantiation has a latent ordering hazard that the
    // shipped code generation happens not to expose (the hazard scanners find nothing in either build; tools/tiling_probe.py,
    // docs/tuning_log.md).  It measured equal to the default at best, so it is not worth the risk: N3DT_X16_TILING=2 now needs
    // N3DT_X16_TILING2_DIAG=1 as well and is meant for that investigation only.  The default tiling (8 waves x 32 samples) is the
    // one every parity test, sweep and bench of four rounds ran on.
    static const int wide = [] {
        const char* e = getenv("N3DT_X16_TILING");
        const char* d = getenv("N3DT_X16_TILING2_DIAG");
        const int t = e ? atoi(e) : X16_DEFAULT_TILING;
        return (t == 2 && !(d && atoi(d) == 1)) ? 1 : t;
    }();
    const long blocks = (long)g->batch * g->n_rays * ((g->n_samples + X16_BS - 1) / X16_BS);
    // (include_vd: the bias of the RGB stage is per ray, so a wave must not span two rays: the one-block-per-wave tiling)
    const bool use_wide = wide == 2 && g->vd_dim == 0 && (((long)g->n_rays * ((g->n_samples + X16_BS - 1) / X16_BS)) % 2 == 0) && blocks >= 2;
    if (precision == N3DT_BF16) {
        if (use_wide) launch_x16<N3DT_BF16, 2, 4>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, stream);
        else launch_x16<N3DT_BF16, 1, 8>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, stream);
    } else {
        if (use_wide) launch_x16<N3DT_F16, 2, 4>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, stream);
        else launch_x16<N3DT_F16, 1, 8>(g, packed, fold, xy, R, T, Kinv, t_rand, part, wlocal, stream);
    }
}
