// ta355 LoRA helper kernels (stage-2 training, BASELINE configs[4]; peft LoraLayer semantics wired at
// tiny_audio/asr_modeling.py:289-301: y = W x + (alpha/r) B (A x), r = 8, alpha = 32, targets q,k,v,o,gate,up,down).
//
// Adapters of the linears that share an input are fused per GROUP g in {qkv, o, gate|up, down}:
//   Acat_g [64, in_g]  rows [j*r, (j+1)*r) = A of member j, remaining rows zero (64 = one GEMM K-tile)
//   Bext_g [N_g, 64]   rows of member j carry B_j in columns [j*r, (j+1)*r), zero elsewhere
// so the adapted linear is ONE extra K-tile of the frozen GEMM:  y = [x | xa] [W | Bext]^T  with  xa = x (s Acat)^T.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "internal.h"

// The fp32 MASTERS keep the exact peft parameter count: la_g [members*r, in_g] (member A's stacked on rows) and
// lb_g [N_g, r] (member B's stacked on rows).  The 64-wide bf16 images are rebuilt from them every forward.

// A master fp32 [R, in] -> s*A image bf16 [64, in] (rows >= R zero), stored in k-step-major blocks [in/32][64][32] (the
// skinny-NT kernel's W operand), and its plain transpose [in, 64]
// blockIdx.y = layer: the masters and the images of consecutive layers are `in_ls` floats / `out_ls` bf16 apart
__global__ __launch_bounds__(256) void lora_pack_a_kernel(const float* __restrict__ in, float scale, bf16_t* __restrict__ out,
                                                          bf16_t* __restrict__ outT, int R, int Cn, long in_ls, long out_ls) {
  in += blockIdx.y * in_ls; out += blockIdx.y * out_ls; outT += blockIdx.y * out_ls;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)64 * Cn) return;
  const int j = (int)(idx / Cn), c = (int)(idx % Cn);
  const bf16_t v = j < R ? f2bf(in[idx] * scale) : (bf16_t)0;
  out[(((long)(c >> 5) * 64 + j) << 5) + (c & 31)] = v;        // k-step-major blocks [Cn/32][64][32] (see lora_skinny_nt_kernel)
  outT[(long)c * 64 + j] = v;
}
// B master fp32 [N, r] -> Bext image bf16 [N, 64] (row n of member j owns columns [j*r, (j+1)*r)) and its transpose
// [64, N] stored in blocks [N/32][64][32] likewise
__global__ __launch_bounds__(256) void lora_pack_b_kernel(const float* __restrict__ in, bf16_t* __restrict__ out,
                                                          bf16_t* __restrict__ outT, int N, int r, int b0, int b1, long in_ls,
                                                          long out_ls) {
  in += blockIdx.y * in_ls; out += blockIdx.y * out_ls; outT += blockIdx.y * out_ls;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)N * 64) return;
  const int n = (int)(idx >> 6), j = (int)(idx & 63);
  const int jj = j - (n < b0 ? 0 : (n < b1 ? 1 : 2)) * r;
  const bf16_t v = (jj >= 0 && jj < r) ? f2bf(in[(long)n * r + jj]) : (bf16_t)0;
  out[idx] = v;
  outT[(((long)(n >> 5) * 64 + j) << 5) + (n & 31)] = v;       // [N/32][64][32]
}

// out[c*so_c + (j - jlo(c))*so_j] += post * sum_m X[m, c] * Y[m, j],  j < R <= 64;  X bf16 [M, C], Y bf16 [M, 64].
// The adapter gradients  dB = dY^T xa  and  dA = s (dY B)^T x  are "TN" products: the contraction index m is the ROW
// index of both row-major operands, while an MFMA lane wants 8 consecutive k of one row/column.  Tiles of 32 rows are
// therefore staged through LDS row-major (coalesced 16-B global loads) and read back transposed (8 ds_read_u16 per
// operand).  The work is HBM-bound (X is read exactly once, arithmetic is 64 flop/byte), so the transposed reads
// are free.  D[j, c] = sum_k Yt[j, k] X[k, c]: A operand from Y, B operand from X; lane (i, g) owns c = i,
// j = 4g + reg of each 16x16 block.  Wave w of the block owns columns [64w, 64w+64) of the 256-column strip.
// LDS bank mapping: rows are 512 B (X) / 128 B (Y) apart, so the four rows of one transposed read would share banks; the
// 32-B column group index is XOR-ed with (row & 3) to spread them.
// With r > 0 only the member block of column c (member boundaries b0, b1; block structure of Bext) is kept and
// lands at j - jlo, i.e. directly in the [N, r] master layout.  grid (ceil(C/256), chunks of 128 rows); out zeroed.
typedef __attribute__((ext_vector_type(8))) short lbf16x8;
typedef __attribute__((ext_vector_type(4))) short lbf16x4;
// LDS transpose read of gfx950 (semantics measured with scripts/probe/tr_probe.hip): in every group of 16 lanes, lane L
// receives element (L & 3) of the 8-byte chunks addressed by lanes (L >> 2) + 4 j, j = 0..3.
__device__ __forceinline__ lbf16x4 tr_read(const bf16_t* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) lbf16x4*)p);
}
typedef __attribute__((ext_vector_type(4))) float lf32x4;

// part != NULL (round 4): row chunk `by` STORES its partial sums at part + by * part_cs (same [c, j] layout as out) and a second kernel
// adds the chunks in a fixed order -- no float atomics: the result does not depend on the order workgroups finish in, and a workgroup's
// last instructions are plain stores instead of 256 x R device-scope atomics that its exit waits for
struct LoraTnArgs { const bf16_t* X; int Cn; const bf16_t* Y; int R; float* out; long so_c, so_j; float post; int r, b0, b1, rows, gx, gy; float* part; long part_cs; };
__device__ __forceinline__ void lora_tn_body(bf16_t* sx, bf16_t* sy, const bf16_t* __restrict__ X, int Cn, const bf16_t* __restrict__ Y,
                                             int R, float* __restrict__ out, long so_c, long so_j, int M, float post, int r, int b0,
                                             int b1, int rows_per_chunk, int bx, int by, float* __restrict__ part = nullptr, long part_cs = 0) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
  const int c0 = bx * 256;
  const int m_begin = by * rows_per_chunk, m_end = min(M, m_begin + rows_per_chunk);
  const int jblocks = (R + 15) >> 4;       // 1..4 (round 4: rank 16 makes the q|k|v group 48 wide)
  lf32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (lf32x4){0.f, 0.f, 0.f, 0.f};

  // Two register stages: the tile two steps ahead is requested while the current one is multiplied (with one stage the loop
  // ran at one memory latency per 32-row tile).
  uint4 px[2][4], py[2];
  auto fetch = [&](int m0, int st_) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int id = tid + q * 256, row = id >> 5, ch = id & 31;
      const int m = m0 + row, c = c0 + ch * 8;
      px[st_][q] = (m < m_end && c < Cn) ? *(const uint4*)(X + (long)m * Cn + c) : make_uint4(0, 0, 0, 0);
    }
    const int row = tid >> 3, ch = tid & 7, m = m0 + row;
    py[st_] = m < m_end ? *(const uint4*)(Y + (long)m * 64 + ch * 8) : make_uint4(0, 0, 0, 0);
  };
  auto stash = [&](int st_) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int id = tid + q * 256, row = id >> 5, ch = id & 31;
      const int sg = (ch >> 1) ^ (row & 3);
      *(uint4*)(sx + row * 256 + sg * 16 + (ch & 1) * 8) = px[st_][q];
    }
    const int row = tid >> 3, ch = tid & 7;
    const int sg = (ch >> 1) ^ (row & 3);
    *(uint4*)(sy + row * 64 + sg * 16 + (ch & 1) * 8) = py[st_];
  };
  auto tile = [&](int m0, auto stage_tag) {
    constexpr int S_ = decltype(stage_tag)::value;
    __syncthreads();                     // previous tile fully consumed
    stash(S_);
    __syncthreads();
    if (m0 + 64 < m_end) fetch(m0 + 64, S_);   // refill this stage: two tiles ahead
    // transposed fragments with ds_read_b64_tr_b16: within a 16-lane group, lane n points at the 4-element chunk
    // (row 8g + (n >> 2) [+4], columns 4 (n & 3) ..) of a [4 rows][16 columns] block and receives the 4 rows of column n --
    // two reads give the 8 consecutive k of one MFMA operand (the scalar version needed 8 ds_read_u16 per operand).
    const int trow = g * 8 + (i >> 2), tcol = (i & 3) * 4;
    lbf16x8 a[4];
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
      if (jb < jblocks) {
        const lbf16x4 lo = tr_read(sy + trow * 64 + ((jb ^ (trow & 3)) * 16) + tcol);
        const lbf16x4 hi = tr_read(sy + (trow + 4) * 64 + ((jb ^ ((trow + 4) & 3)) * 16) + tcol);
        a[jb] = (lbf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      }
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const int cg = wave * 4 + cb;
      const lbf16x4 lo = tr_read(sx + trow * 256 + ((cg ^ (trow & 3)) * 16) + tcol);
      const lbf16x4 hi = tr_read(sx + (trow + 4) * 256 + ((cg ^ ((trow + 4) & 3)) * 16) + tcol);
      const lbf16x8 b = (lbf16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
      for (int jb = 0; jb < 4; ++jb)
        if (jb < jblocks) acc[jb][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[jb], b, acc[jb][cb], 0, 0, 0);
    }
  };
  fetch(m_begin, 0);
  if (m_begin + 32 < m_end) fetch(m_begin + 32, 1);
  for (int m0 = m_begin; m0 < m_end; m0 += 64) {
    tile(m0, std::integral_constant<int, 0>{});
    if (m0 + 32 < m_end) tile(m0 + 32, std::integral_constant<int, 1>{});
  }
#pragma unroll
  for (int cb = 0; cb < 4; ++cb) {
    const int c = c0 + (wave * 4 + cb) * 16 + i;
    if (c >= Cn) continue;
    int jlo = 0, jhi = R;
    if (r > 0) { jlo = (c < b0 ? 0 : (c < b1 ? 1 : 2)) * r; jhi = jlo + r; }
#pragma unroll
    for (int jb = 0; jb < 4; ++jb)
      if (jb < jblocks) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = jb * 16 + g * 4 + q;
          if (j >= jlo && j < jhi) {
            const long o = (long)c * so_c + (long)(j - jlo) * so_j;
            if (part) part[(long)by * part_cs + o] = acc[jb][cb][q] * post;
            else unsafeAtomicAdd(out + o, acc[jb][cb][q] * post);
          }
        }
      }
  }
}

__global__ __launch_bounds__(256) void lora_tn_mfma_kernel(const bf16_t* __restrict__ X, int Cn, const bf16_t* __restrict__ Y,
                                                           int R, float* __restrict__ out, long so_c, long so_j, int M,
                                                           float post, int r, int b0, int b1, int rows_per_chunk) {
  __shared__ __attribute__((aligned(16))) bf16_t sx[32 * 256];
  __shared__ __attribute__((aligned(16))) bf16_t sy[32 * 64];
  lora_tn_body(sx, sy, X, Cn, Y, R, out, so_c, so_j, M, post, r, b0, b1, rows_per_chunk, blockIdx.x, blockIdx.y);
}
// Two TN products in ONE launch (round 3): the adapter gradients dB = dY^T xa and dA = s (dY B)^T x of a group are independent,
// each alone is a 13-14 us launch that fills the chip badly (dA of the q|k|v group: 192 workgroups), and stage 2 issues 112 such
// pairs per step.  Workgroups [0, n0) run problem 0, the rest problem 1.
__global__ __launch_bounds__(256) void lora_tn_dual_kernel(const LoraTnArgs p0, const LoraTnArgs p1, int M) {
  __shared__ __attribute__((aligned(16))) bf16_t sx[32 * 256];
  __shared__ __attribute__((aligned(16))) bf16_t sy[32 * 64];
  const int n0 = p0.gx * p0.gy;
  int b = blockIdx.x;
  const bool second = b >= n0;
  const LoraTnArgs& p = second ? p1 : p0;
  if (second) b -= n0;
  lora_tn_body(sx, sy, p.X, p.Cn, p.Y, p.R, p.out, p.so_c, p.so_j, M, p.post, p.r, p.b0, p.b1, p.rows, b % p.gx, b / p.gx, p.part, p.part_cs);
}

// out[M, 64] (bf16) = X[M, K] W[64, K]^T : the rank-space projections xa = x (sAcat)^T and dyB = dy Bext.
// Both operands are K-major, so MFMA fragments load straight from global memory (16 B per lane, no LDS staging).
// A 128x128 GEMM tile grid would be 48 workgroups here; instead one workgroup owns 32 rows and all 64 columns and
// its EIGHT waves split K (wave w takes k-steps w, w+8, ...), meeting in LDS: 192 workgroups x 8 waves for M = 6144.
// The kernel is a stream over X (HBM) with W (<= 768 KB) resident in L2; two k-steps are kept in flight per wave so
// that ~32 KB of X loads are outstanding per CU.
constexpr int SK_WAVES = 8;
// NCB (round 3): only the first NCB 16-column blocks of W hold an adapter (24 / 8 / 16 / 8 of the 64 rows for the q|k|v, o,
// gate|up and down groups: NCB = 2, 1, 1, 1); the rest of the 64-wide image is zero by construction and is written as zeros
// without being fetched or multiplied.  The W fragments (re-read from L2 by every workgroup) were 4 KB per k-step and wave
// against 2 KB of X: the kernel ran at half the HBM rate on L2 traffic it did not need.
// RB = 16-row blocks per workgroup: 2 (M / 32 workgroups: 188 for the B = 32 step, on 256 CUs) or 1 (376, twice the loads in flight)
template <int NCB, int RB>
__global__ __launch_bounds__(SK_WAVES * 64) void lora_skinny_nt_kernel(const bf16_t* __restrict__ X, int K,
                                                                       const bf16_t* __restrict__ W,
                                                                       bf16_t* __restrict__ out, int M) {
  __shared__ float red[SK_WAVES][16 * RB][NCB * 16 + 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * (16 * RB);
  lf32x4 acc[RB][NCB];
#pragma unroll
  for (int a = 0; a < RB; ++a)
#pragma unroll
    for (int b = 0; b < NCB; ++b) acc[a][b] = (lf32x4){0.f, 0.f, 0.f, 0.f};
  const int ra = min(m0 + i, M - 1), rb = min(m0 + (RB - 1) * 16 + i, M - 1);       // clamped rows are never stored
  const bf16_t* xa = X + (long)ra * K + g * 8;
  const bf16_t* xb = X + (long)rb * K + g * 8;
  // W is stored in k-step-major blocks [K/32][64][32]: the four fragment loads of one k-step read 4 KB contiguously.
  // (Row-major [64, K] put all 64 rows of a k-step K*2 bytes apart -- a multiple of 4 KB for every K of the model, i.e.
  // on ONE L2 channel, with every workgroup asking for the same lines at the same time.)
  const bf16_t* wp = W + (long)i * 32 + g * 8;
  // Every workgroup walks its k-steps from a different starting point (rot): X rows are K*2 bytes apart too, so
  // workgroups in lock step would all sit on the same two L2 channels.
  const int nsteps = K / 32;
  const int ns = nsteps > wave ? (nsteps - wave + SK_WAVES - 1) / SK_WAVES : 0;      // k-steps of this wave
  const int rot = ns ? (int)((blockIdx.x * 7u) % (unsigned)ns) : 0;
  auto kof = [&](int sidx) { int q = sidx + rot; if (q >= ns) q -= ns; return (wave + SK_WAVES * q) * 32; };
  constexpr int PF = 3;
  lbf16x8 xr[PF][RB], wr[PF][NCB];
#pragma unroll
  for (int p = 0; p < PF; ++p)
    if (p < ns) {
      const int kp = kof(p);
      xr[p][0] = *(const lbf16x8*)(xa + kp); if (RB > 1) xr[p][RB - 1] = *(const lbf16x8*)(xb + kp);
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) wr[p][cb] = *(const lbf16x8*)(wp + ((long)(kp >> 5) * 64 + cb * 16) * 32);
    }
  for (int s0 = 0; s0 < ns; s0 += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
      const int sc = s0 + p;
      if (sc < ns) {                                 // wave-uniform
        const lbf16x8 a0 = xr[p][0], a1 = xr[p][RB - 1];
        lbf16x8 b[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) b[cb] = wr[p][cb];
        if (sc + PF < ns) {                          // refill this ring slot
          const int kf = kof(sc + PF);
          xr[p][0] = *(const lbf16x8*)(xa + kf); if (RB > 1) xr[p][RB - 1] = *(const lbf16x8*)(xb + kf);
#pragma unroll
          for (int cb = 0; cb < NCB; ++cb) wr[p][cb] = *(const lbf16x8*)(wp + ((long)(kf >> 5) * 64 + cb * 16) * 32);
        }
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) {
          acc[0][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b[cb], acc[0][cb], 0, 0, 0);
          if (RB > 1) acc[RB - 1][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b[cb], acc[RB - 1][cb], 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int rbk = 0; rbk < RB; ++rbk)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int q = 0; q < 4; ++q) red[wave][rbk * 16 + g * 4 + q][cb * 16 + i] = acc[rbk][cb][q];
  __syncthreads();
  for (int e = tid; e < 16 * RB * 64; e += SK_WAVES * 64) {
    const int row = e >> 6, col = e & 63;
    float v = 0.f;
    if (col < NCB * 16) {
#pragma unroll
      for (int w = 0; w < SK_WAVES; ++w) v += red[w][row][col];
    }
    if (m0 + row < M) out[(long)(m0 + row) * 64 + col] = f2bf(v);
  }
}

int ta_i_lora_pack_a(const float* in, float scale, void* out, void* outT, int R, int Cn, int layers, long in_ls, long out_ls,
                     hipStream_t st) {
  TA_LAUNCH(lora_pack_a_kernel, dim3(ta_cdiv(64 * Cn, 256), layers), dim3(256), 0, st, in, scale, (bf16_t*)out, (bf16_t*)outT, R, Cn,
            in_ls, out_ls);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
int ta_i_lora_pack_b(const float* in, void* out, void* outT, int N, int r, int b0, int b1, int layers, long in_ls, long out_ls,
                     hipStream_t st) {
  TA_LAUNCH(lora_pack_b_kernel, dim3(ta_cdiv(N * 64, 256), layers), dim3(256), 0, st, in, (bf16_t*)out, (bf16_t*)outT, N, r, b0, b1,
            in_ls, out_ls);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
// rows per workgroup: every workgroup ends with 256 x R float atomics, so fewer, longer row chunks are better as long
// as ~512 workgroups remain (Cn = 6144: 288 rows -> 22 x 24 workgroups and 2.1 M atomics instead of 4.7 M)
static int lora_tn_rows(int M, int Cn) {
  int rows = 128;
  const long want = (long)M * ta_cdiv(Cn, 256) / 512;
  if (want > rows) rows = (int)((want + 31) / 32 * 32);
  return rows;
}
int ta_i_lora_skinny_tn(const void* X, int Cn, const void* Y, int ldy, int R, float* out, long so_c, long so_j, int M, float post,
                        int r, int b0, int b1, hipStream_t st) {
  if (R > 64 || R <= 0 || ldy != 64 || Cn % 8) return TA_ERR_ARG;
  if (M <= 0) return TA_OK;
  const int rows = lora_tn_rows(M, Cn);
  TA_LAUNCH(lora_tn_mfma_kernel, dim3(ta_cdiv(Cn, 256), ta_cdiv(M, rows)), dim3(256), 0, st, (const bf16_t*)X, Cn, (const bf16_t*)Y,
            R, out, so_c, so_j, M, post, r, b0, b1, rows);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
// rows per workgroup of a pair: about 384 workgroups over both problems (45.2 / 45.1 / 44.95 / 45.25 ms per LoRA step at 256 / 320 /
// 384 / 512, profiles/r03_u_ab_lora_tn_wgs.txt)
constexpr long LORA_TN_WGS = 384;
int ta_i_lora_tn2_rows(int M, int Cn0, int Cn1) {
  const long wgs_env = LORA_TN_WGS;
  const long gx = ta_cdiv(Cn0, 256) + ta_cdiv(Cn1, 256);
  const long want = ((long)M * gx + wgs_env - 1) / wgs_env;
  return (int)(((want < 32 ? 32 : want) + 31) / 32 * 32);
}
// out[layer][kind][e] = sum over the chunks of part[layer][kind][chunk][e], chunk 0 first: ONE launch for every adapter gradient of
// the step (grid: element blocks x 8 kinds x layers)
__global__ __launch_bounds__(256) void lora_reduce_parts_kernel(LoraReduceDesc d) {
  const int k = blockIdx.y, l = blockIdx.z;
  const long n = d.size[k];
  const float* p = d.part + (long)l * d.part_ls + d.part_off[k];
  float* o = d.out[k] + (long)l * d.out_ls[k];
  for (long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4; e < n; e += (long)gridDim.x * 1024) {
    float4 a = *(const float4*)(p + e);
    for (int c = 1; c < d.chunks[k]; ++c) {
      const float4 b = *(const float4*)(p + (long)c * n + e);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    *(float4*)(o + e) = a;
  }
}
int ta_i_lora_reduce_parts(const LoraReduceDesc& d, int layers, hipStream_t st) {
  long mx = 0;
  for (int k = 0; k < 8; ++k) { if (d.size[k] % 4) return TA_ERR_ARG; mx = d.size[k] > mx ? d.size[k] : mx; }
  if (layers <= 0 || mx <= 0) return TA_OK;
  TA_LAUNCH(lora_reduce_parts_kernel, dim3((unsigned)ta_cdiv(mx, 1024), 8, layers), dim3(256), 0, st, d);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
// both adapter gradients of one group in one launch: problem 0 = (X0, Y0, ...), problem 1 likewise (same M)
int ta_i_lora_skinny_tn2(const void* X0, int Cn0, const void* Y0, int R0, float* out0, long so_c0, long so_j0, float post0, int r0,
                         int b00, int b10, const void* X1, int Cn1, const void* Y1, int R1, float* out1, long so_c1, long so_j1,
                         float post1, int r1, int b01, int b11, int M, float* part0, long part_cs0, float* part1, long part_cs1,
                         hipStream_t st) {
  if (R0 > 64 || R0 <= 0 || R1 > 64 || R1 <= 0 || (Cn0 % 8) || (Cn1 % 8)) return TA_ERR_ARG;
  if (M <= 0) return TA_OK;
  LoraTnArgs p0 = {(const bf16_t*)X0, Cn0, (const bf16_t*)Y0, R0, out0, so_c0, so_j0, post0, r0, b00, b10, lora_tn_rows(M, Cn0), 0, 0, part0, part_cs0};
  LoraTnArgs p1 = {(const bf16_t*)X1, Cn1, (const bf16_t*)Y1, R1, out1, so_c1, so_j1, post1, r1, b01, b11, lora_tn_rows(M, Cn1), 0, 0, part1, part_cs1};
  p0.gx = ta_cdiv(Cn0, 256); p1.gx = ta_cdiv(Cn1, 256);
  // Rows per workgroup of the pair (round 3): as many as leave about ONE workgroup per CU over both problems.  Every workgroup ends
  // with 256 x R float atomics on addresses shared with the other row chunks of its column strip, and a rows-per-workgroup sweep of
  // the LoRA step (profiles/r03_t_ab_lora_tn_rows.txt: 47.5 / 46.2 / 45.8 / 47.1 / 51.4 ms at 96 / 192 / 576 / 1152 / 3008 rows)
  // says the launch is bound by them until the chip runs out of workgroups.  TA355_LORA_TN_WGS = the target count (default 384: 45.2 / 45.1 / 44.95 / 45.25 ms at 256 / 320 / 384 / 512, profiles/r03_u_ab_lora_tn_wgs.txt),
  // TA355_LORA_TN_ROWS = a fixed row count as before.
  const long wgs_env = LORA_TN_WGS;
  if (part0 || part1) p0.rows = p1.rows = ta_i_lora_tn2_rows(M, Cn0, Cn1);       // (the caller sized the partial buffers with it)
  else {
    const long want = ((long)M * (p0.gx + p1.gx) + wgs_env - 1) / wgs_env;
    const int rows = (int)((want < 32 ? 32 : want) + 31) / 32 * 32;
    p0.rows = p1.rows = rows;
  }
  p0.gy = ta_cdiv(M, p0.rows);
  p1.gy = ta_cdiv(M, p1.rows);
  TA_LAUNCH(lora_tn_dual_kernel, dim3(p0.gx * p0.gy + p1.gx * p1.gy), dim3(256), 0, st, p0, p1, M);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
// R = rows of W that can be non-zero (members * rank of the group; 64 = all)
int ta_i_lora_skinny_nt(const void* X, int K, const void* W, void* out, int M, int R, hipStream_t st) {
  if (K % 32 || K <= 0 || R <= 0 || R > 64) return TA_ERR_ARG;
  if (M <= 0) return TA_OK;
  const int ncb = (R + 15) / 16;                   // only the 16-column blocks that hold an adapter (r03: 12.5 -> 9.3 us per launch)
  // 32 rows per workgroup (16 -- twice the workgroups -- measured equal to slightly slower, profiles/r03_y_ab_lora_nt_rows.txt)
  const dim3 grid(ta_cdiv(M, 32)), blk(SK_WAVES * 64);
#define SKNT(NCB_, RB_) TA_LAUNCH((lora_skinny_nt_kernel<NCB_, RB_>), grid, blk, 0, st, (const bf16_t*)X, K, (const bf16_t*)W, (bf16_t*)out, M)
  if (ncb == 1) SKNT(1, 2); else if (ncb == 2) SKNT(2, 2); else if (ncb == 3) SKNT(3, 2); else SKNT(4, 2);
#undef SKNT
  TA_CHECK_LAUNCH();
  return TA_OK;
}
