// Buffers of the renderer's fused mixed-precision training path (bf16 maps): what the forward (neural_render.hip,
// run_x16_train) saves for the backward (train_nr.hip, nr_bwd16), and the scratch both carve out of the caller's workspace.
// Offsets are in BYTES; every region starts on a 256-byte boundary.
//
// Per block i (input x_i [M][C], M = nb * h * h pixels; output net_i [4M][CO] at twice the resolution):
//   t1   [M][2C]      lrelu(layer_1 x)                               operand of dW2, gate of layer_1
//   y    [4][M][C]    the four sub-pixel planes of lrelu(layer_2 t1), BEFORE the residual   operand of dWf (+ x), gate of layer_2
//   net  [4M][CO]     lrelu(Blur(feat_layers ps) + b), raster order   operand of d feat_2_rgb, gate, next block's x
// and the sigmoid output img [nb][3][P][P] (fp32) for sigmoid'.
#pragma once
#include <stddef.h>

#include "../../include/n3dt.h"

static inline int nr16_ch(int C, int i) {
    int v = C >> i;
    return v < 32 ? 32 : v;
}
static inline size_t nr16_al(size_t b) { return (b + 255) & ~(size_t)255; }

struct Nr16Saved {
    size_t t1[N3DT_MAX_BLOCKS], y[N3DT_MAX_BLOCKS], net[N3DT_MAX_BLOCKS], img, total;
};
static inline Nr16Saved nr16_saved_layout(const N3dtGeom* g, int nb) {
    Nr16Saved s;
    size_t o = 0;
    for (int i = 0; i < g->n_blocks; ++i) {
        const size_t h = (size_t)g->featmap_size << i, M = (size_t)nb * h * h;
        const size_t ci = nr16_ch(g->feat_nc, i), co = nr16_ch(g->feat_nc, i + 1);
        s.t1[i] = o; o += nr16_al(M * 2 * ci * 2);
        s.y[i] = o; o += nr16_al(4 * M * ci * 2);
        s.net[i] = o; o += nr16_al(4 * M * co * 2);
    }
    const size_t P = (size_t)g->featmap_size << g->n_blocks;
    s.img = o; o += nr16_al((size_t)nb * 3 * P * P * sizeof(float));
    s.total = o;
    return s;
}

// Transposed / permuted fp32 weights of one block for the backward's GEMMs (y[m][n] = sum_k A[m][k] Wt[n][k]):
//   g1 [C][CO]         Wt[c][o]           = Wf[o][c]                       d ps   = d hid . Wf
//   g2 [2C][4C]        Wt[n][q C + c]     = W2[4c + q][n]                  d t1   = d tv (4 planes) . W2
//   g3 [C][2C + 4 CO]  Wt[k][j]           = W1[j][k]            (j < 2C)   d x    = d t1 . W1
//                      Wt[k][2C + q CO + o] = R_q[o][k]                           + sum_q d hid_q . R_q   (the residual path)
//      with R_q[o][k] = sum of Wf[o][c] over the four c with (4c + q) % C == k (zero unless k % 4 == q), as in the forward.
struct Nr16Wt {
    size_t g1, g2, g3, floats;
};
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline Nr16Wt nr16_wt_layout(int C, int CO) {
    Nr16Wt w;
    size_t o = 0;
    w.g1 = o; o += (size_t)C * CO;
    w.g2 = o; o += (size_t)2 * C * 4 * C;
    w.g3 = o; o += (size_t)C * (2 * C + 4 * CO);
    w.floats = (o + 63) & ~(size_t)63;
    return w;
}

struct Nr16Ws {
    // forward: hid planes, the RGB pyramid (two planar fp32 buffers), the packed weight streams of the fused block kernels
    size_t hid, rgbA, rgbB, packw;
    // backward: d_pre (raster) / d_hid (planes) [4M][CO], d_tv planes [4][M][C], d_t1 [M][2C], d_x ping-pong [M][C] (16-bit),
    // planar fp32 d_rgb ping-pong, transposed weights of every block
    size_t dpre, dhid, dtv, dt1, dxa, dxb, drgb, dtmp, wt[N3DT_MAX_BLOCKS];
    size_t dwpart;  // per-XCD partial sums of the layer_2 / layer_1 weight gradients of every block (dw_rowmajor.h)
    size_t total;
};
size_t nrf_packed_bytes_for(int C, int CO);  // neural_render.hip (the fused block kernel's packed stream + bias table)
static inline Nr16Ws nr16_ws_layout(const N3dtGeom* g, int nb) {
    Nr16Ws w;
    size_t e_hid = 0, e_tv = 0, e_t1 = 0, e_x = 0, pack = 0;
    for (int i = 0; i < g->n_blocks; ++i) {
        const size_t h = (size_t)g->featmap_size << i, M = (size_t)nb * h * h;
        const size_t ci = nr16_ch(g->feat_nc, i), co = nr16_ch(g->feat_nc, i + 1);
        if (4 * M * co > e_hid) e_hid = 4 * M * co;
        if (4 * M * ci > e_tv) e_tv = 4 * M * ci;
        if (2 * M * ci > e_t1) e_t1 = 2 * M * ci;
        if (M * ci > e_x) e_x = M * ci;
        pack += nr16_al(nrf_packed_bytes_for((int)ci, (int)co));
    }
    const size_t P = (size_t)g->featmap_size << g->n_blocks, rgb = nr16_al((size_t)nb * 3 * P * P * sizeof(float));
    size_t o = 0;
    w.hid = o; o += nr16_al(e_hid * 2);
    w.rgbA = o; o += rgb;
    w.rgbB = o; o += rgb;
    w.packw = o; o += pack;
    w.dpre = o; o += nr16_al(e_hid * 2);
    w.dhid = o; o += nr16_al(e_hid * 2);
    w.dtv = o; o += nr16_al(e_tv * 2);
    w.dt1 = o; o += nr16_al(e_t1 * 2);
    w.dxa = o; o += nr16_al(e_x * 2);
    w.dxb = o; o += nr16_al(e_x * 2);
    w.drgb = o; o += rgb;
    w.dtmp = o; o += rgb;
    for (int i = 0; i < g->n_blocks; ++i) {
        w.wt[i] = o;
        o += nr16_al(nr16_wt_layout(nr16_ch(g->feat_nc, i), nr16_ch(g->feat_nc, i + 1)).floats * sizeof(float));
    }
    w.dwpart = o;
    for (int i = 0; i < g->n_blocks; ++i) {
        const size_t ci = nr16_ch(g->feat_nc, i);
        o += nr16_al(sizeof(float) * 8 * (4 * ci) * (2 * ci)) + nr16_al(sizeof(float) * 8 * (2 * ci) * ci);
    }
    w.total = o;
    return w;
}
// the fused training path covers the block sizes the fused block kernel is built for
bool nr16_supported(const N3dtGeom* g);  // neural_render.hip
