// Row-major front end of the fused weight-gradient kernel (train_x16.inc: dw_x16_body), used by the 2-D renderer's backward.
#pragma once
#include <stddef.h>
#include <hip/hip_runtime.h>

struct N3dtDwRm {       // one product out[rows][cols] += A^T B over M pixels
    const void* A;      // [M][ldA] 16-bit; a_planes > 1: that many planes of [M][ldA], a_plane_elems apart, side by side as rows
    long ldA;
    int a_planes;
    size_t a_plane_elems;
    const void* B;      // [M][ldB] 16-bit
    long ldB;
    long M;
    int rows, cols;     // rows = a_planes * (channels per plane)
    float* out;
    long ld_out;
    int row_perm;       // > 0: output row r = q * row_perm + c goes to row 4 c + q (pixel_shuffle's channel order)
    float* rowsum;      // bias gradient (column sums of A), indexed like the output rows; nullable
};
#define N3DT_DW_RM_MAX 10
extern "C" size_t n3dt_dw_rowmajor_part_bytes(const N3dtDwRm* e, int n);
extern "C" bool n3dt_dw_rowmajor_ok(const N3dtDwRm* e);
extern "C" void n3dt_launch_dw_rowmajor_zero(const N3dtDwRm* e, int n, float* part, hipStream_t s);
extern "C" void n3dt_launch_dw_rowmajor_one(const N3dtDwRm* e, int index, float* part, hipStream_t s);
extern "C" void n3dt_launch_dw_rowmajor_reduce(const N3dtDwRm* e, int n, float* part, hipStream_t s);
