// Packed-weight layout shared by the pack kernels, the fused render kernels and the host API.
//
// The MLP is executed on "logical" layer matrices whose input-column order is the order of the
// kernel's activation rows, with the per-frame latent columns removed (they are folded into a
// per-frame bias, SURVEY section 6 footnote 1):
//
//   stage  logical [N x K]      source (reference: NetWorks/models.py:32-59)
//   L0     384 x 64             FeaExt_module_0[:, 0:63] | zero column
//   L1-4   384 x 384            FeaExt_module_1..4
//   L5     384 x 448            FeaExt_module_5[:, 0:63] | zero | FeaExt_module_5[:, 63+S : 63+S+384]
//   L6-7   384 x 384            FeaExt_module_6..7
//   DEN    32 x 384             density_module (row 0), rows 1..31 zero
//   RGB0   384 x 384            RGB_layer_0
//   RGB1   192 x 384            RGB_layer_1[:, 0:384]
//   RGB2   handled per ray after compositing (linear layer commutes with the weighted sum):
//          stored transposed fp32 [192][256] + bias[256]
#pragma once
#include <stddef.h>
#include <stdint.h>

#define N3DT_NSTAGE 11  // L0..L7, DEN, RGB0, RGB1

struct N3dtStage {
    int N, K;      // logical rows (padded), logical columns (padded)
    int layer;     // index into N3dtMlpParams
    int relu;
};

__host__ __device__ inline N3dtStage n3dt_stage(int s) {
    switch (s) {
        case 0: return {384, 64, 0, 1};
        case 5: return {384, 448, 5, 1};
        case 8: return {32, 384, 8, 1};
        case 9: return {384, 384, 9, 0};
        case 10: return {192, 384, 10, 1};
        default: return {384, 384, s, 1};
    }
}

// element offset of stage s inside the packed matrix region (elements, not bytes)
__host__ __device__ inline size_t n3dt_stage_offset(int s) {
    size_t off = 0;
    for (int i = 0; i < s; ++i) {
        N3dtStage st = n3dt_stage(i);
        off += (size_t)st.N * st.K;
    }
    return off;
}

__host__ __device__ inline size_t n3dt_packed_matrix_elems() { return n3dt_stage_offset(N3DT_NSTAGE); }

// the fp32 tail that follows the matrices in every precision: W2^T [192][256], b2 [256]
// (+ the fp32 merged RGB matrix W_m [192][384] the 16-bit matrices are packed from)
// ... then the merged RGB matrix W_m [192][384] (fp32), and W2^T once more as bf16 hi + lo MFMA B fragments for the per-ray
// head (n3dt_tail_w2_frags_offset: piece ((ks * 8 + tile) * 2 + part) of 1 KiB, lane-linear; part 0 = hi, 1 = lo)
__host__ __device__ inline size_t n3dt_tail_w2_frags_offset() { return (size_t)192 * 256 + 256 + (size_t)192 * 384; }  // in floats
__host__ __device__ inline size_t n3dt_packed_tail_floats() { return n3dt_tail_w2_frags_offset() + (size_t)12 * 8 * 2 * 256; }

__host__ __device__ inline size_t n3dt_packed_elem_bytes(int precision) { return precision == 0 ? 4 : 2; }

// The 16-bit precisions carry the matrices twice: region A in the fragment order of the 32x32x16 MFMA tiling
// (nerf_fwd_x16.hip, the training kernels), region B in that of the 16x16x32 tiling (nerf_fwd_x16b.hip).
// The split-precision mode (precision 3, nerf_fwd_x16s.hip) carries ONE region in the 32x32x16 order with every 1 KiB piece
// twice: piece 2P = bf16(W), piece 2P + 1 = bf16(W - bf16(W)).
__host__ __device__ inline size_t n3dt_packed_region_bytes(int precision) {
    size_t b = n3dt_packed_matrix_elems() * n3dt_packed_elem_bytes(precision) * (precision == 3 ? 2 : 1);
    return (b + 255) & ~(size_t)255;
}
__host__ __device__ inline size_t n3dt_packed_region_b_offset(int precision) { return n3dt_packed_region_bytes(precision); }

// byte offset of the fp32 tail (256-byte aligned)
__host__ __device__ inline size_t n3dt_packed_tail_offset(int precision) {
    return ((precision == 0 || precision == 3) ? 1 : 2) * n3dt_packed_region_bytes(precision);
}

// per-frame bias table in the workspace, [B][N3DT_FOLD_STRIDE], stage order:
//   b0'[384] | b1..b4 [4x384] | b5'[384] | b6,b7 [2x384] | bden[32] (row 0 = density bias) | brgb0[384] | brgb1'[192]
// b0', b5', brgb1' carry the folded latent codes of the frame; the others are plain copies so that
// a wave reads every bias of its frame through one uniform (scalar-load) base pointer.
#define N3DT_FOLD_STRIDE (384 * 10 + 32 + 192)
__host__ __device__ inline int n3dt_bias_offset(int stage) {
    return stage <= 8 ? 384 * stage : (stage == 9 ? 384 * 8 + 32 : 384 * 9 + 32);
}

// include_vd (N3dtGeom.vd_dim > 0): RGB_layer_1's bias is per RAY -- the frame's folded entry + the caller's `ray_bias` term --
// in a table [B * N_r][192] that sits right behind the fold table wherever that lives (render workspace, the training paths'
// saved buffers), so that the fused kernels find it from the `fold` pointer they already take.
#define N3DT_RAYFOLD_STRIDE 192
__host__ __device__ inline size_t n3dt_rayfold_offset(int batch) { return ((size_t)batch * N3DT_FOLD_STRIDE + 63) & ~(size_t)63; }  // floats
__host__ __device__ inline size_t n3dt_fold_region_floats(int batch, int n_rays, int vd_dim) {
    return vd_dim > 0 ? n3dt_rayfold_offset(batch) + (size_t)batch * n_rays * N3DT_RAYFOLD_STRIDE : (size_t)batch * N3DT_FOLD_STRIDE;
}
