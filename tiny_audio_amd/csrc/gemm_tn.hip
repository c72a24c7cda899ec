// ta355 "TN" GEMM:  out[Ny, Nx] (+)= Y[M, Ny]^T  X[M, Nx]   (bf16 operands, both row-major, contraction over the ROWS).
//
// This is the weight-gradient product of a linear layer, dW[out, in] = dY^T X with dY [tokens, out] and X [tokens, in]
// (full decoder fine-tuning, TF:models/qwen3/modeling_qwen3.py linears under freeze_language_model=False).  An MFMA lane
// wants 8 consecutive k of one row/column, but here k is the row index of both operands: the "NT" kernel of gemm.hip
// therefore needs explicit transposes of dY and X first (two extra HBM passes per product).  This kernel stages
// 32-row tiles of both operands row-major in LDS (coalesced 16-B global loads) and reads the MFMA fragments TRANSPOSED
// with ds_read_b64_tr_b16 (gfx950): in each group of 16 lanes, lane L receives element (L & 3) of the 8-byte chunks
// addressed by lanes (L >> 2) + 4 j, j = 0..3 (measured: scripts/probe/tr_probe.hip) -- so when lane n points at
// chunk (row 8g + (n >> 2), columns 4 (n & 3) ..) of a [4 rows][16 columns] block, lane L ends up with the 4 rows of column
// L, and two reads give the 8 consecutive k of one operand.
//
// Tile: 128 columns of Y x 256 columns of X per workgroup (4 waves; wave w owns X columns [64w, 64w+64) against all 128
// Y columns: 8 x 4 accumulator blocks), a chunk of rows per workgroup (grid.z), partial sums to f32 slabs that
// splitk-style reduce into out.  LDS rows are 256 B (Y) / 512 B (X) apart, so the 32-byte column group index is XOR-ed
// with (row & 3): the four rows of one transposed read land on 4 x 8 distinct banks.
#include <cstdlib>
#include <type_traits>
#include "common.h"

typedef __attribute__((ext_vector_type(8))) short tbf16x8;
typedef __attribute__((ext_vector_type(4))) short tbf16x4;
typedef __attribute__((ext_vector_type(4))) float tf32x4;

__device__ __forceinline__ tbf16x4 tn_tr_read(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tbf16x4*)p);
}

constexpr int TN_BY = 128, TN_BX = 256;

// (A 64-row stage -- half the barriers, one register stage -- measured 5 % slower than the 32-row stage with two register
// stages below: the kernel is bound by LDS bandwidth (24 KB written + 48 KB read per 32-row step per workgroup against 32
// MFMAs per wave), not by its barriers.)
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ Y, int Ny, const bf16_t* __restrict__ X, int Nx,
                                                         float* __restrict__ out, long slab_stride, int M, int rows_per_chunk) {
  __shared__ __attribute__((aligned(16))) bf16_t sx[32 * TN_BX];
  __shared__ __attribute__((aligned(16))) bf16_t sy[32 * TN_BY];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
  const int x0 = blockIdx.x * TN_BX, y0 = blockIdx.y * TN_BY;
  const int m_begin = blockIdx.z * rows_per_chunk, m_end = min(M, m_begin + rows_per_chunk);
  tf32x4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (tf32x4){0.f, 0.f, 0.f, 0.f};

  // two register stages of the next tiles (X: 32 x 256 = 4 chunks of 16 B per thread, Y: 32 x 128 = 2)
  uint4 px[2][4], py[2][2];
  auto fetch = [&](int m0, int s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int id = tid + q * 256, row = id >> 5, ch = id & 31;
      const int m = m0 + row, c = x0 + ch * 8;
      px[s][q] = (m < m_end && c < Nx) ? *(const uint4*)(X + (long)m * Nx + c) : make_uint4(0, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int id = tid + q * 256, row = id >> 4, ch = id & 15;
      const int m = m0 + row, c = y0 + ch * 8;
      py[s][q] = (m < m_end && c < Ny) ? *(const uint4*)(Y + (long)m * Ny + c) : make_uint4(0, 0, 0, 0);
    }
  };
  auto stash = [&](int s) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int id = tid + q * 256, row = id >> 5, ch = id & 31;
      const int sg = (ch >> 1) ^ (row & 3);
      *(uint4*)(sx + row * TN_BX + sg * 16 + (ch & 1) * 8) = px[s][q];
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int id = tid + q * 256, row = id >> 4, ch = id & 15;
      const int sg = (ch >> 1) ^ (row & 3);
      *(uint4*)(sy + row * TN_BY + sg * 16 + (ch & 1) * 8) = py[s][q];
    }
  };
  const int trow = g * 8 + (i >> 2), tcol = (i & 3) * 4;
  auto tile = [&](int m0, auto stage_tag) {
    constexpr int S_ = decltype(stage_tag)::value;
    __syncthreads();                     // previous tile fully consumed
    stash(S_);
    __syncthreads();
    if (m0 + 64 < m_end) fetch(m0 + 64, S_);
    tbf16x8 b[4];
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int cg = wave * 4 + cb;      // 16-column group of X owned by this wave
      const tbf16x4 lo = tn_tr_read(sx + trow * TN_BX + ((cg ^ (trow & 3)) * 16) + tcol);
      const tbf16x4 hi = tn_tr_read(sx + (trow + 4) * TN_BX + ((cg ^ ((trow + 4) & 3)) * 16) + tcol);
      b[cb] = (tbf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
      const tbf16x4 lo = tn_tr_read(sy + trow * TN_BY + ((jb ^ (trow & 3)) * 16) + tcol);
      const tbf16x4 hi = tn_tr_read(sy + (trow + 4) * TN_BY + ((jb ^ ((trow + 4) & 3)) * 16) + tcol);
      const tbf16x8 a = (tbf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[jb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[cb], acc[jb][cb], 0, 0, 0);
    }
  };
  if (m_begin < m_end) fetch(m_begin, 0);
  if (m_begin + 32 < m_end) fetch(m_begin + 32, 1);
  for (int m0 = m_begin; m0 < m_end; m0 += 64) {
    tile(m0, std::integral_constant<int, 0>{});
    if (m0 + 32 < m_end) tile(m0 + 32, std::integral_constant<int, 1>{});
  }
  // D[j, c]: lane (i, g) owns X column c = i and Y columns j = 4 g + q of each 16 x 16 block
  float* slab = out + (long)blockIdx.z * slab_stride;
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int c = x0 + (wave * 4 + cb) * 16 + i;
    if (c >= Nx) continue;
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = y0 + jb * 16 + g * 4 + q;
        if (j < Ny) slab[(long)j * Nx + c] = acc[jb][cb][q];
      }
  }
}

// out[i] = (accumulate ? out[i] : 0) + sum_z slab[z][i]
__global__ void tn_reduce_kernel(const float* __restrict__ slabs, int nz, long slab_stride, float* __restrict__ out, int accumulate,
                                 long n4) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n4; k += (long)gridDim.x * blockDim.x) {
    float4 s = accumulate ? ((const float4*)out)[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z = 0; z < nz; ++z) {
      const float4 v = ((const float4*)(slabs + (long)z * slab_stride))[k];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    ((float4*)out)[k] = s;
  }
}

static int tn_chunks(int M, int Ny, int Nx) {
  const long tiles = (long)ta_cdiv(Nx, TN_BX) * ta_cdiv(Ny, TN_BY);
  const long target = 512;
  long z = (target + tiles - 1) / tiles;               // ~512 workgroups (two per CU)
  const long zmax = ta_cdiv(M, 64);
  if (z > zmax) z = zmax;
  if (z < 1) z = 1;
  return (int)z;
}

#include "../../include/ta355.h"
extern "C" long ta_gemm_bf16_tn_ws_bytes(int M, int Ny, int Nx) {
  return (long)tn_chunks(M, Ny, Nx) * Ny * Nx * 4;
}

// out f32 [Ny, Nx] (+)= Y^T X;  Ny, Nx multiples of 8; ws: ta_gemm_bf16_tn_ws_bytes() bytes of scratch.
extern "C" int ta_gemm_bf16_tn(const void* Y, const void* X, float* out, int M, int Ny, int Nx, int accumulate, void* ws,
                               long ws_bytes, hipStream_t st) {
  if (Ny <= 0 || Nx <= 0) return TA_OK;
  if ((Ny & 7) || (Nx & 7) || (((long)Ny * Nx) & 3)) return TA_ERR_ARG;
  const long n = (long)Ny * Nx;
  if (M <= 0) {
    if (!accumulate && hipMemsetAsync(out, 0, n * 4, st) != hipSuccess) return TA_ERR_LAUNCH;
    return TA_OK;
  }
  const int nz = tn_chunks(M, Ny, Nx);
  if (!ws || ws_bytes < (long)nz * n * 4) return TA_ERR_ARG;
  int rows = (int)((ta_cdiv(M, nz) + 31) / 32 * 32);
  const int nzz = ta_cdiv(M, rows);
  TA_LAUNCH(gemm_tn_kernel, dim3(ta_cdiv(Nx, TN_BX), ta_cdiv(Ny, TN_BY), nzz), dim3(256), 0, st, (const bf16_t*)Y, Ny,
            (const bf16_t*)X, Nx, (float*)ws, n, M, rows);
  TA_CHECK_LAUNCH();
  int blocks = (int)((n / 4 + 255) / 256); if (blocks > 2048) blocks = 2048;
  TA_LAUNCH(tn_reduce_kernel, dim3(blocks), dim3(256), 0, st, (const float*)ws, nzz, n, out, accumulate, n / 4);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
