// ta355 fused attention (flash-style, online softmax, MFMA 16x16x32 bf16), gfx950.
//
//   attn_fwd_kernel<64,false>   GLM-ASR encoder self-attention: non-causal, NO mask (padded frames are
//                               attended to, exactly as the reference does), 20 heads x 64
//                               (TF:models/glmasr/modeling_glmasr.py:187-217)
//   attn_fwd_kernel<128,true>   Qwen3 attention: causal + key-padding mask, GQA 16q/8kv x 128
//                               (TF:models/qwen3/modeling_qwen3.py:185-208,264-275)
//   attn_bwd_dq_kernel / attn_bwd_dkv_kernel   activation gradients of the latter (frozen LM, dX only)
//
// Formulation: everything is computed TRANSPOSED so that the softmax row (one query) lives in one
// lane column:  S^T[key,q] = K Q^T  -> lane holds q = lane&15 and 4 consecutive keys per 16-key
// sub-tile.  Row max / sum need only two cross-lane steps (xor 16, 32); the running rescale of
// O^T[d,q] is lane-local; P^T feeds the second MFMA as the B operand straight from registers (the
// MFMA k-slot permutation is applied identically to the A operand, read from a [d][key] image of V).
// That image comes from the producer (qkv_post kernels write V^T / K^T / Q^T / dO^T copies: HBM is
// plentiful, LDS transposes are not free), so every LDS tile here is a plain row-major copy:
//   "row tiles"  [64 rows][HD]   XOR-swizzled 16-B chunks -> conflict-free ds_read_b128 fragments
//   "col tiles"  [HD rows][64]   8-B granules XOR-swizzled by the row -> conflict-free ds_read_b64 halves
//                (first version padded rows to 144 B: SQ_LDS_BANK_CONFLICT showed 30 % conflict cycles)
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/ta355.h"

#define KV_TILE 64
#define CT_STRIDE 128          // bytes, [d][64 keys] tiles; 8-byte granule index XOR (row & 15)
#define LOG2E 1.4426950408889634f
#define NEG_BIG (-1.0e30f)

// Experiment builds only (-DTA355_ATTN_STAMPS, scripts/attn_stamps.py): s_memtime stamps at the phase boundaries of the LM attention
// kernels, 16 per workgroup, read back with ta_debug_attn_stamps.  The product library compiles none of it.
#ifdef TA355_ATTN_STAMPS
__device__ unsigned long long g_attn_stamps[8192 * 16];
#define ATTN_STAMP(blk, i, who) do { if (threadIdx.x == (who)) g_attn_stamps[(long)(blk) * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int ta_debug_attn_stamps(unsigned long long* host_out, int nblocks) {
  return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_stamps), (size_t)nblocks * 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 2;
}
#else
#define ATTN_STAMP(blk, i, who) do { } while (0)
#endif

// Row tiles are unpadded [64][HD] with the 16-B chunk index XOR-swizzled by the row: conflict-free ds_read_b128
// fragments under either lane grouping of the instruction (same scheme the GEMM uses; measured 0 conflicts there).
template <int HD> struct RowTile {
  static constexpr int STRIDE = HD * 2;
  static constexpr int BYTES = 64 * STRIDE;
  static __device__ __forceinline__ int swz(int r) { return HD == 64 ? ((r >> 1) & 7) : (r & 15); }
  // byte offset of 16-B chunk c of row r
  static __device__ __forceinline__ int off(int r, int c) { return r * STRIDE + ((c ^ swz(r)) << 4); }
  // Round 6: tiles that are ALSO read transposed (ds_read_b64_tr_b16: V in the forward; K, Q, dO in the backward) take this swizzle.
  // A transposed read's 32-lane group covers 8 consecutive rows x 32 B (chunks ca, ca ^ 1 of every row); under `r & 15` rows r and
  // r ^ 1 put (ca ^ r) and (ca ^ 1 ^ (r ^ 1)) in the SAME 16-B bank slot -- a 2-way conflict on every transposed read (PMC r05:
  // SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.27 in attn_bwd_kernel).  XOR by 2 * (r & 7) spreads the 16 (row, chunk) pairs over
  // all 16 slots, and the ds_read_b128 fragment pattern (16 rows x one chunk column per lane group) stays conflict-free under it:
  // rows {0-3, 12-15} at chunk c and rows {4-11} at chunk c + 1 land on the even and the odd slots.  (Same idea as attention_enc's vswz.)
  static __device__ __forceinline__ int swz_tr(int r) { return HD == 64 ? ((r >> 1) & 7) : ((r & 7) << 1); }
  static __device__ __forceinline__ int off_tr(int r, int c) { return r * STRIDE + ((c ^ swz_tr(r)) << 4); }
};
template <int HD> struct ColTile { static constexpr int BYTES = HD * CT_STRIDE; };

// ---- staging helpers: global -> registers (issued early) -> LDS (written after the barrier)
template <int HD>
struct RowStage {   // 64 rows x HD bf16, 256 threads
  static constexpr int N = HD / 32;   // 16-byte chunks per thread
  uint4 v[N];
  __device__ __forceinline__ void load(const bf16_t* base, long row_stride, int row0, int nrows_valid, int tid) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int ch = tid + i * 256;
      const int r = ch / (HD / 8), c = ch % (HD / 8);
      int gr = row0 + r; if (gr > nrows_valid - 1) gr = nrows_valid - 1; if (gr < 0) gr = 0;
      v[i] = *(const uint4*)(base + (long)gr * row_stride + c * 8);
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int ch = tid + i * 256;
      const int r = ch / (HD / 8), c = ch % (HD / 8);
      *(uint4*)(lds + RowTile<HD>::off(r, c)) = v[i];
    }
  }
};
template <int HD>
struct ColStage {   // HD rows x 64 bf16 (128 B per row) from a [.., HD, Lp] image
  static constexpr int N = HD / 32;
  uint4 v[N];
  // al: 0 = every chunk 16-B aligned, 1 = 8-B aligned (e.g. clip offsets b * 500 columns), 2 = element aligned
  __device__ __forceinline__ void load(const bf16_t* base, long Lp, int col0, int tid, int al = 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int ch = tid + i * 256;
      const int r = ch >> 3, c = ch & 7;
      const bf16_t* src = base + (long)r * Lp + col0 + c * 8;
      if (al == 0) {
        v[i] = *(const uint4*)src;
      } else if (al == 1) {
        const uint2 a = *(const uint2*)src, b = *(const uint2*)(src + 4);
        v[i] = make_uint4(a.x, a.y, b.x, b.y);
      } else {
        uint32_t u[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) u[k] = (uint32_t)src[2 * k] | ((uint32_t)src[2 * k + 1] << 16);
        v[i] = make_uint4(u[0], u[1], u[2], u[3]);
      }
    }
  }
  __device__ __forceinline__ void store(char* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const int ch = tid + i * 256;
      const int r = ch >> 3, c = ch & 7;
      // granule g8 of row r lives at g8 ^ (r & 15): the 16-B chunk moves to c ^ ((r & 15) >> 1) and its two
      // 8-B halves swap when r is odd
      const uint4 x = v[i];
      const uint4 y = (r & 1) ? make_uint4(x.z, x.w, x.x, x.y) : x;
      *(uint4*)(lds + r * CT_STRIDE + ((c ^ ((r & 15) >> 1)) << 4)) = y;
    }
  }
};

// XCD-aware block order.  Workgroup b runs on XCD b % 8 and each XCD has a private L2, so all workgroups that
// stream the SAME K/V (one batch element x one kv head: every query tile of every query head in the GQA group) are
// given ids with the same residue mod 8:  id = (j * gsz + member) * 8 + xcd,  group = j * 8 + xcd.
// (Measured before this: 775 MB fetched per encoder-attention launch for 82 MB of K/V -- each of the 8 query tiles
// of a head ran on a different XCD and pulled its own copy.)  Grid = gsz * round_up(ngroups, 8); tail groups exit.
__device__ __forceinline__ bool decode_group(int id, int gsz, int ngroups, int& group, int& member) {
  const int xcd = id & 7, w = id >> 3;
  member = w % gsz;
  group = (w / gsz) * 8 + xcd;
  return group < ngroups;
}
static inline int grouped_grid(int gsz, int ngroups) { return gsz * ((ngroups + 7) / 8 * 8); }

__device__ __forceinline__ bf16x8 pack_p(const f32x4& a, const f32x4& b) {
  union { bf16x8 v; uint32_t u[4]; } r;
  r.u[0] = pack2bf(a[0], a[1]); r.u[1] = pack2bf(a[2], a[3]);
  r.u[2] = pack2bf(b[0], b[1]); r.u[3] = pack2bf(b[2], b[3]);
  return r.v;
}
__device__ __forceinline__ bf16x8 read_colfrag(const char* tile, int row, int c0, int c1) {
  // two 8-byte halves: columns [c0, c0+4) and [c1, c1+4) of row `row`
  union { bf16x8 v; uint2 h[2]; } r;
  r.h[0] = *(const uint2*)(tile + row * CT_STRIDE + (((c0 >> 2) ^ (row & 15)) << 3));
  r.h[1] = *(const uint2*)(tile + row * CT_STRIDE + (((c1 >> 2) ^ (row & 15)) << 3));
  return r.v;
}

// The same fragment (row d = dt * 16 + l15 of the [d][64] image, columns [c0, c0+4) and [c1, c1+4)) read TRANSPOSED out of the
// ROW tile ([64 rows][HD], RowTile layout) with ds_read_b64_tr_b16 (gfx950): in each group of 16 lanes, lane L receives element
// (L & 3) of the 8-byte chunks addressed by lanes (L >> 2) + 4 j, j = 0..3.  Lane n of a group addresses (row c + (n >> 2),
// columns dt * 16 + 4 (n & 3) ..): lane L then holds rows c .. c+3 of column dt * 16 + L.  The backward kernels use it for the
// K^T / Q^T / dO^T operands, so the producers need not write (and the backward need not stage) transposed images.
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
template <int HD>
__device__ __forceinline__ bf16x8 read_colfrag_tr(const char* rowtile, int dt, int l15, int c0, int c1) {
  const int dcol = dt * 16 + 4 * (l15 & 3);                     // first of the 4 columns of this lane's 8-byte chunk
  const int chunk = dcol >> 3, half = (dcol >> 2) & 1;
  const int r0 = c0 + (l15 >> 2), r1 = c1 + (l15 >> 2);
  const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4_t*)(rowtile + RowTile<HD>::off_tr(r0, chunk) + half * 8));
  const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4_t*)(rowtile + RowTile<HD>::off_tr(r1, chunk) + half * 8));
  return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// ============================================================================ forward
// cross-lane combines over the 4 lane groups (g = lane>>4) that share a query column, on the VALU (no LDS round trip):
// permlane16_swap(x,x) leaves {own, partner(xor 16)} in the two results, permlane32_swap likewise for xor 32.
// 3-input max: the compiler fuses the two maxnum calls into one v_max3_f32.  (An inline-asm v_max3_f32 here was WRONG: the
// hazard recognizer does not see inside asm, and once no other VALU work separated it from the QK^T MFMAs it read their
// accumulators before the matrix pipe had written them -- a stale, smaller row maximum, i.e. run-to-run 1-ulp noise.)
__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
// the same for a sum; and sums / exchanges inside a 16-lane row on the DPP path (no LDS crossbar: __shfl_xor lowers to ds_bpermute)
__device__ __forceinline__ float group_sum(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
  return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row_sum8(float v) {                 // over the 8 lanes of a half row: quad, quad, half-row mirror
  v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v);
  return v;
}
__device__ __forceinline__ float row_sum16(float v) { v = row_sum8(v); v += dpp_mov<0x140>(v); return v; }   // + row mirror
__device__ __forceinline__ float row_xor8(float v) { return dpp_mov<0x128>(v); }                              // row_ror:8 = lane ^ 8
__device__ __forceinline__ float group_max(float x) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}

// One workgroup = 128 query rows of one head (4 waves x 2 sub-tiles of 16 rows): every K / V^T fragment read from LDS
// feeds two MFMAs.  The softmax denominator is NOT summed on the VALU: the V^T image carries 16 extra rows whose
// first is all ones, so one extra MFMA block per k-step accumulates l = sum_k P[q,k] next to O (and is rescaled
// with it).  Per score element that leaves: 1 FMA + 1 v_exp + 1/2 v_max3 + 1/2 v_cvt_pk on the VALU.
template <int HD, bool CAUSAL, int QSUB>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                       const bf16_t* __restrict__ VT, bf16_t* __restrict__ O,
                                                       float* __restrict__ LSE, const int* __restrict__ kmask,
                                                       int B, int Hq, int Hkv, int L, int Lp, float scale,
                                                       const ta_attn_layout lay, const int valign) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ND = HD / 16;                        // O^T row blocks; block ND is the row-sum block
  char* Ks = smem;
  char* Vs = smem + RowTile<HD>::BYTES;              // (HD + 16) rows x 144 B
  int* Ms = (int*)(Vs + (HD + 16) * CT_STRIDE);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  constexpr int QROWS = 64 * QSUB;
  const int nq = (L + QROWS - 1) / QROWS, grp = Hq / Hkv;
  int group, member;
  if (!decode_group(blockIdx.x, grp * nq, B * Hkv, group, member)) return;
  const int b = group / Hkv, hk = group % Hkv;
  const int h = hk * grp + member / nq, qt = member % nq;
  const int q0 = qt * QROWS + wave * 16 * QSUB;         // first query row of this wave
  const bf16_t* Qb = Q + b * lay.q_bs + h * lay.q_hs;
  const bf16_t* Kb = K + b * lay.k_bs + hk * lay.k_hs;
  const bf16_t* Vb = VT + b * lay.v_bs + hk * lay.v_hs;
  const float sl2 = scale * LOG2E;

  // rows HD .. HD+15 of the V^T image: [1 1 1 ...] then zeros (written once, never restaged)
  for (int i = tid; i < 16 * (CT_STRIDE / 4); i += 256) {
    const int r = i / (CT_STRIDE / 4);
    ((uint32_t*)(Vs + HD * CT_STRIDE))[i] = r == 0 ? 0x3f803f80u : 0u;
  }
  bf16x8 qf[QSUB][HD / 32];
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) {
    int qr = q0 + sub * 16 + l15; if (qr > L - 1) qr = L - 1;
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) qf[sub][ks] = *(const bf16x8*)(Qb + (long)qr * lay.q_rs + ks * 32 + g * 8);
  }
  f32x4 o[QSUB][ND + 1];
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub)
#pragma unroll
    for (int i = 0; i <= ND; ++i) o[sub][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[QSUB];
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) m_run[sub] = NEG_BIG;

  int ntiles = (L + KV_TILE - 1) / KV_TILE;
  if (CAUSAL) { const int lim = (qt * QROWS + QROWS - 1) / KV_TILE + 1; if (lim < ntiles) ntiles = lim; }

  RowStage<HD> ks_reg; ColStage<HD> vs_reg; int mk_reg = 1;
  ks_reg.load(Kb, lay.k_rs, 0, L, tid);
  vs_reg.load(Vb, lay.v_rs, 0, tid, valign & 3);
  if (kmask && tid < 64) mk_reg = (tid < L) ? kmask[(long)b * L + tid] : 0;

  // The tile body exists twice: MASKED = false has no masking code at all (the compiler otherwise hoists the index
  // compares of the masked path in front of the branch: ~40 VALU per tile in a VALU-bound loop), MASKED = true is the
  // general one.  Tiles [0, nfull) are complete, unmasked and (causal) entirely below the diagonal for every query here.
  auto tile = [&](const int t, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const int key0 = t * KV_TILE;
    ks_reg.store(Ks, tid);
    vs_reg.store(Vs, tid);
    if (tid < 64) Ms[tid] = mk_reg;
    __syncthreads();
    if (t + 1 < ntiles) {
      ks_reg.load(Kb, lay.k_rs, key0 + KV_TILE, L, tid);
      vs_reg.load(Vb, lay.v_rs, key0 + KV_TILE, tid, valign & 3);
      if (kmask && tid < 64) { const int kk = key0 + KV_TILE + tid; mk_reg = (kk < L) ? kmask[(long)b * L + kk] : 0; }
    }
    // ---- S^T = K Q^T for both query sub-tiles
    f32x4 s[QSUB][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int sub = 0; sub < QSUB; ++sub) s[sub][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        const bf16x8 a = *(const bf16x8*)(Ks + RowTile<HD>::off(kt * 16 + l15, ks * 4 + g));
#pragma unroll
        for (int sub = 0; sub < QSUB; ++sub)
          s[sub][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[sub][ks], s[sub][kt], 0, 0, 0);
      }
    }
    // ---- online softmax numerators (the denominator rides in the MFMA below)
#pragma unroll
    for (int sub = 0; sub < QSUB; ++sub) {
      const int qrow = q0 + sub * 16 + l15;
      if constexpr (MASKED) {                        // selects, no per-element branches
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const int4 mk = *(const int4*)(Ms + kt * 16 + g * 4);
          const int mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + kt * 16 + g * 4 + r;
            bool v = key < L;
            if (CAUSAL) v = v & (key <= qrow);
            if (kmask) v = v & (mkv[r] != 0);
            s[sub][kt][r] = v ? s[sub][kt][r] : -INFINITY;
          }
        }
      }
      float mloc = max3(s[sub][0][0], s[sub][0][1], s[sub][0][2]);
      mloc = max3(mloc, s[sub][0][3], s[sub][1][0]);
      mloc = max3(mloc, s[sub][1][1], s[sub][1][2]);
      mloc = max3(mloc, s[sub][1][3], s[sub][2][0]);
      mloc = max3(mloc, s[sub][2][1], s[sub][2][2]);
      mloc = max3(mloc, s[sub][2][3], s[sub][3][0]);
      mloc = max3(mloc, s[sub][3][1], s[sub][3][2]);
      mloc = fmaxf(mloc, s[sub][3][3]);
      const float m_new = fmaxf(m_run[sub], group_max(mloc));
      if (__any(m_new != m_run[sub])) {              // rescale only when some row's running max moved
        const float alpha = __builtin_amdgcn_exp2f((m_run[sub] - m_new) * sl2);
#pragma unroll
        for (int i = 0; i <= ND; ++i) { o[sub][i][0] *= alpha; o[sub][i][1] *= alpha; o[sub][i][2] *= alpha; o[sub][i][3] *= alpha; }
        m_run[sub] = m_new;
      }
      const float mb = m_new * sl2;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[sub][kt][r] = __builtin_amdgcn_exp2f(fmaf(s[sub][kt][r], sl2, -mb));
    }
    // ---- [O^T ; l] += [V^T ; 1] P^T
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      bf16x8 pb[QSUB];
#pragma unroll
      for (int sub = 0; sub < QSUB; ++sub) pb[sub] = pack_p(s[sub][2 * kp], s[sub][2 * kp + 1]);
#pragma unroll
      for (int dt = 0; dt <= ND; ++dt) {
        const bf16x8 va = read_colfrag(Vs, dt * 16 + l15, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4);
#pragma unroll
        for (int sub = 0; sub < QSUB; ++sub)
          o[sub][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb[sub], o[sub][dt], 0, 0, 0);
      }
    }
    __syncthreads();
  };
  int nfull = (kmask == nullptr) ? L / KV_TILE : 0;
  if (CAUSAL) nfull = min(nfull, (qt * QROWS) / KV_TILE);      // keys of tiles below the first query row of the workgroup
  nfull = min(nfull, ntiles);
  if (valign & 4) nfull = 0;                                   // experiment (TA355_ATTN_NOPEEL=1): every tile takes the general body
  for (int t = 0; t < nfull; ++t) tile(t, std::false_type{});
  for (int t = nfull; t < ntiles; ++t) tile(t, std::true_type{});
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) {
    const int qrow = q0 + sub * 16 + l15;
    const float l_run = __shfl(o[sub][ND][0], l15, 64);     // row 0 of the sum block lives in lane group g = 0
    if (qrow < L) {
      const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
      bf16_t* orow = O + ((long)b * L + qrow) * ((long)Hq * HD) + (long)h * HD;
#pragma unroll
      for (int dt = 0; dt < ND; ++dt) {
        uint2 w;
        w.x = pack2bf(o[sub][dt][0] * inv, o[sub][dt][1] * inv);
        w.y = pack2bf(o[sub][dt][2] * inv, o[sub][dt][3] * inv);
        *(uint2*)(orow + dt * 16 + g * 4) = w;
      }
      if (LSE && g == 0) LSE[(long)(b * Hq + h) * L + qrow] = l_run > 0.f ? m_run[sub] * scale + __logf(l_run) : 1.0e30f;
    }
  }
}

// ============================================================================ forward, short sequences (LM, L <= 192)
// One workgroup = one (clip, kv head): ALL keys / values of the sequence are staged once (3 K row tiles + 3 V^T column
// tiles, 102 KB of LDS) and shared by every query of the GQA group -- 2 q heads x 192 queries = 12 waves x 2 sub-tiles
// of 16 rows.  After the single staging barrier each wave walks its causal key range alone: no barriers, no
// re-staging, and B * Hkv = 256 workgroups are one round of the 256 CUs.  (The tiled kernel above spent its time in
// 1536 small workgroups that each re-staged K / V for 64 queries: 42 us per layer for 4.8 GFLOP.)
template <int HD, int MAXT, int QSUB, int NW>
__global__ __launch_bounds__(NW * 64) void attn_fwd_gqa_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                           const bf16_t* __restrict__ VT, bf16_t* __restrict__ O,
                                                           float* __restrict__ LSE, const int* __restrict__ kmask,
                                                           int B, int Hq, int Hkv, int L, int Lp, float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ND = HD / 16, VBYTES = (HD + 16) * CT_STRIDE, NT_ = NW * 64;
  char* Ks = smem;                                   // MAXT row tiles
  char* Vs = smem + MAXT * RowTile<HD>::BYTES;       // MAXT column tiles, each with its ones-row block
  int* Ms = (int*)(Vs + MAXT * VBYTES);              // key mask, MAXT * 64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.x / Hkv, hk = blockIdx.x % Hkv;
  // query chunks of 16 * QSUB rows: chunk c belongs to q head c / cph; a wave takes chunk `wave` and then its mirror
  // image (nchunk - 1 - wave), which pairs a short causal key range with a long one
  const int grp = Hq / Hkv, cph = (L + 16 * QSUB - 1) / (16 * QSUB), nchunk = grp * cph;
  const bf16_t* Kb = K + ((long)(b * Hkv + hk) * L) * HD;
  const bf16_t* Vb = VT + ((long)(b * Hkv + hk) * HD) * Lp;
  const float sl2 = scale * LOG2E;
  const int ntiles = (L + KV_TILE - 1) / KV_TILE;
  // ---- stage everything once.  All global loads are issued before the first LDS store (MAXT * 1024 16-B chunks of K
  // and of V^T = 4 + 4 per thread): a load -> store loop per tile serialises 12 round trips to memory (measured: the
  // kernel took 38 us that way, no faster than the tiled one).
  constexpr int PER = (MAXT * 64 * (HD / 8) + NT_ - 1) / NT_;
  uint4 kreg[PER], vreg[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * NT_, t = ch / (64 * (HD / 8)), w = ch % (64 * (HD / 8));
    if (t < ntiles) {
      const int r = w / (HD / 8), c = w % (HD / 8);
      int gr = t * KV_TILE + r; if (gr > L - 1) gr = L - 1;
      kreg[i] = *(const uint4*)(Kb + (long)gr * HD + c * 8);
      const int vr = w >> 3, vc = w & 7;
      vreg[i] = *(const uint4*)(Vb + (long)vr * Lp + t * KV_TILE + vc * 8);
    }
  }
  for (int i = tid; i < ntiles * 16 * (CT_STRIDE / 4); i += NT_) {
    const int t = i / (16 * (CT_STRIDE / 4)), w = i % (16 * (CT_STRIDE / 4)), r = w / (CT_STRIDE / 4);
    ((uint32_t*)(Vs + t * VBYTES + HD * CT_STRIDE))[w] = r == 0 ? 0x3f803f80u : 0u;
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int ch = tid + i * NT_, t = ch / (64 * (HD / 8)), w = ch % (64 * (HD / 8));
    if (t < ntiles) {
      const int r = w / (HD / 8), c = w % (HD / 8);
      *(uint4*)(Ks + t * RowTile<HD>::BYTES + RowTile<HD>::off(r, c)) = kreg[i];
      const int vr = w >> 3, vc = w & 7;
      const uint4 x = vreg[i];
      const uint4 y = (vr & 1) ? make_uint4(x.z, x.w, x.x, x.y) : x;
      *(uint4*)(Vs + t * VBYTES + vr * CT_STRIDE + ((vc ^ ((vr & 15) >> 1)) << 4)) = y;
    }
  }
  for (int i = tid; i < ntiles * 64; i += NT_) Ms[i] = (i < L) ? (kmask ? kmask[(long)b * L + i] : 1) : 0;
  __syncthreads();
  for (int pass = 0; pass < 2; ++pass) {
  const int chunk = pass == 0 ? wave : nchunk - 1 - wave;
  if (chunk >= nchunk || (pass == 1 && chunk < NW)) continue;          // wave-uniform
  const int h = hk * grp + chunk / cph;
  const int q0 = (chunk % cph) * 16 * QSUB;
  const bf16_t* Qb = Q + ((long)(b * Hq + h) * L) * HD;
  bf16x8 qf[QSUB][HD / 32];
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) {
    int qr = q0 + sub * 16 + l15; if (qr > L - 1) qr = L - 1;
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) qf[sub][ks] = *(const bf16x8*)(Qb + (long)qr * HD + ks * 32 + g * 8);
  }
  f32x4 o[QSUB][ND + 1];
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub)
#pragma unroll
    for (int i = 0; i <= ND; ++i) o[sub][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[QSUB];
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) m_run[sub] = NEG_BIG;
  const int my_tiles = min(ntiles, (q0 + 16 * QSUB - 1) / KV_TILE + 1);      // causal: keys beyond the wave's last query never count
  for (int t = 0; t < my_tiles; ++t) {
    const int key0 = t * KV_TILE;
    const char* Kt = Ks + t * RowTile<HD>::BYTES;
    const char* Vt = Vs + t * VBYTES;
    f32x4 s[QSUB][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int sub = 0; sub < QSUB; ++sub) s[sub][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        const bf16x8 a = *(const bf16x8*)(Kt + RowTile<HD>::off(kt * 16 + l15, ks * 4 + g));
#pragma unroll
        for (int sub = 0; sub < QSUB; ++sub)
          s[sub][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[sub][ks], s[sub][kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);             // keep the K fragments of one key block live at a time (VGPR budget)
    }
#pragma unroll
    for (int sub = 0; sub < QSUB; ++sub) {
      const int qrow = q0 + sub * 16 + l15;
      const bool full = (key0 + KV_TILE <= L) && (kmask == nullptr) && (key0 + KV_TILE - 1 <= q0 + sub * 16);
      if (!full) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const int4 mk = *(const int4*)(Ms + key0 + kt * 16 + g * 4);
          const int mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + kt * 16 + g * 4 + r;
            const bool v = (key <= qrow) & (mkv[r] != 0);
            s[sub][kt][r] = v ? s[sub][kt][r] : -INFINITY;
          }
        }
      }
      float mloc = max3(s[sub][0][0], s[sub][0][1], s[sub][0][2]);
      mloc = max3(mloc, s[sub][0][3], s[sub][1][0]);
      mloc = max3(mloc, s[sub][1][1], s[sub][1][2]);
      mloc = max3(mloc, s[sub][1][3], s[sub][2][0]);
      mloc = max3(mloc, s[sub][2][1], s[sub][2][2]);
      mloc = max3(mloc, s[sub][2][3], s[sub][3][0]);
      mloc = max3(mloc, s[sub][3][1], s[sub][3][2]);
      mloc = fmaxf(mloc, s[sub][3][3]);
      const float m_new = fmaxf(m_run[sub], group_max(mloc));
      if (__any(m_new != m_run[sub])) {
        const float alpha = __builtin_amdgcn_exp2f((m_run[sub] - m_new) * sl2);
#pragma unroll
        for (int i = 0; i <= ND; ++i) { o[sub][i][0] *= alpha; o[sub][i][1] *= alpha; o[sub][i][2] *= alpha; o[sub][i][3] *= alpha; }
        m_run[sub] = m_new;
      }
      const float mb = m_new * sl2;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[sub][kt][r] = __builtin_amdgcn_exp2f(fmaf(s[sub][kt][r], sl2, -mb));
    }
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      bf16x8 pb[QSUB];
#pragma unroll
      for (int sub = 0; sub < QSUB; ++sub) pb[sub] = pack_p(s[sub][2 * kp], s[sub][2 * kp + 1]);
#pragma unroll
      for (int dt = 0; dt <= ND; ++dt) {
        const bf16x8 va = read_colfrag(Vt, dt * 16 + l15, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4);
#pragma unroll
        for (int sub = 0; sub < QSUB; ++sub)
          o[sub][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb[sub], o[sub][dt], 0, 0, 0);
        if (dt & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) {
    const int qrow = q0 + sub * 16 + l15;
    const float l_run = __shfl(o[sub][ND][0], l15, 64);
    if (qrow < L) {
      const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
      bf16_t* orow = O + ((long)b * L + qrow) * ((long)Hq * HD) + (long)h * HD;
      // adjacent 16-column blocks are exchanged between lane rows (v_permlane16_swap) so that a lane stores 8
      // consecutive columns: 64 contiguous bytes per row per store instead of 32 (see gemm.hip epilogue_strip)
#pragma unroll
      for (int dt = 0; dt < ND; dt += 2) {
        const uint32_t x0 = pack2bf(o[sub][dt][0] * inv, o[sub][dt][1] * inv), x1 = pack2bf(o[sub][dt][2] * inv, o[sub][dt][3] * inv);
        const uint32_t y0 = pack2bf(o[sub][dt + 1][0] * inv, o[sub][dt + 1][1] * inv), y1 = pack2bf(o[sub][dt + 1][2] * inv, o[sub][dt + 1][3] * inv);
        const auto a = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
        const auto c = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
        *(uint4*)(orow + 16 * (dt + (g & 1)) + 8 * (g >> 1)) = make_uint4(a[0], c[0], a[1], c[1]);
      }
      if (LSE && g == 0) LSE[(long)(b * Hq + h) * L + qrow] = l_run > 0.f ? m_run[sub] * scale + __logf(l_run) : 1.0e30f;
    }
  }
  }   // pass
}

// ---- round 3: global -> LDS DMA staging of [64][128] row tiles for the backward (double-buffered LDS, no staging registers).
// Inline assembly: behind the builtin the compiler drains vmcnt in front of every later transposing LDS read (it tracks the DMA as
// an LDS store that may alias them), which would wait for the NEXT tile right after issuing it.  Every wait for these loads is the
// explicit s_waitcnt vmcnt(0) at the top of an iteration.  `lds` = wave-uniform LDS byte address, the hardware adds lane * 16.
__device__ __forceinline__ void dma16_asm(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(g) : "memory");
}
// rows row0 .. row0 + 63 (clamped to nrows - 1) of a [.., 128] bf16 matrix with `row_stride` elements per row -> the RowTile<128>
// image at `lds_tile`: one wave instruction covers 4 rows x 256 B, wave w of 4 moves row blocks w, w + 4, w + 8, w + 12; the XOR
// swizzle of the image is applied to the SOURCE chunk (the LDS side of the DMA is lane-linear).
__device__ __forceinline__ void dma_rowtile128(const bf16_t* base, long row_stride, int row0, int nrows, unsigned lds_tile, int wave,
                                               int lane) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int blk = wave + 4 * i, r = blk * 4 + (lane >> 4);
    int gr = row0 + r; if (gr > nrows - 1) gr = nrows - 1; if (gr < 0) gr = 0;
    const int c = (lane & 15) ^ RowTile<128>::swz_tr(r);       // the backward's tiles: every one is read with both patterns
    dma16_asm(base + (long)gr * row_stride + c * 8, __builtin_amdgcn_readfirstlane(lds_tile + blk * 1024));   // (wave-uniform by construction)
  }
}
__device__ __forceinline__ unsigned lds_addr_of(const char* p) {
  return __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)p);
}
constexpr int BWD_BUF = 2 * 64 * 128 * 2 + 512;      // one staging buffer of the backward: two row tiles + 128 floats / ints

// ============================================================================ forward, short sequences, straight from the q|k|v GEMM output
// attn_fwd_gqa_kernel with ta_lm_qkv_post_fwd folded in: one workgroup = one (clip, kv head) reads the PRE-norm q | k | v rows of its
// GQA group from the token-major GEMM output, applies the per-head RMSNorm (q_norm / k_norm) and RoPE while staging
// (TF:models/qwen3/modeling_qwen3.py:50-64,211-240), keeps K and V as ROW tiles in LDS (the V^T operand of the second product is read
// transposed with ds_read_b64_tr_b16, the row of ones that accumulates the softmax denominator is a constant fragment) and writes
// what the backward needs -- normalised Q / K and V head-major, 1/rms per (token, head) -- on the way.  No separate post kernel, no
// V^T image.
//   staging: 16 lanes per key row (one 16-B chunk of 8 dims each; chunk c and c + 8 are RoPE partners, 8 lanes apart)
//   queries: in the MFMA fragment layout (lane = row l15, dims ks * 32 + g * 8 ..): the 4 g lanes share a row, dims d and d + 64 are
//            fragments ks and ks + 2 of the same lane
// PAIR (round 6, GQA groups of two query heads): a wave's two 16-row sub-tiles are the SAME 16 rows of the group's two heads instead of
// 32 consecutive rows of one head, so the rows' cos / sin table entries (512 B per row) are fetched once for both heads.  The phase
// stamps of this kernel (scripts/attn_stamps.py, profiles/r06_d_attn_stamps_f32.txt) show it moving ~784 KB per workgroup through the
// CU's vector-memory path at ~10 B per cycle for its whole life -- it is bound by that path, not by latency or MFMA -- and 196 KB of
// it were query-side table rows.  Results are bit-identical (each row's arithmetic and its order over the keys do not change).
template <int HD, int MAXT, int QSUB, int NW, bool PAIR = false>
__global__ __launch_bounds__(NW * 64) void attn_fwd_gqa_qkv_kernel(const bf16_t* __restrict__ qkv0, const float* __restrict__ qn_w,
                                                               const float* __restrict__ kn_w, const float* __restrict__ cosT,
                                                               const float* __restrict__ sinT, const int* __restrict__ pos,
                                                               bf16_t* __restrict__ Qo, bf16_t* __restrict__ Ko, bf16_t* __restrict__ Vo,
                                                               float* __restrict__ rq, float* __restrict__ rk, bf16_t* __restrict__ O,
                                                               float* __restrict__ LSE, const int* __restrict__ kmask, int B, int Hq,
                                                               int Hkv, int L, float scale, float eps) {
  static_assert(HD == 128, "Qwen3 head_dim");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int ND = HD / 16, NT_ = NW * 64;
  char* Ks = smem;                                   // MAXT row tiles of normalised, rotated K
  char* Vs = smem + MAXT * RowTile<HD>::BYTES;       // MAXT row tiles of V
  int* Ms = (int*)(Vs + MAXT * RowTile<HD>::BYTES);  // key mask, MAXT * 64
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int b = blockIdx.x / Hkv, hk = blockIdx.x % Hkv;
  const int grp = Hq / Hkv, cph = PAIR ? (L + 15) / 16 : (L + 16 * QSUB - 1) / (16 * QSUB), nchunk = PAIR ? cph : grp * cph;
  const long ld = (long)(Hq + 2 * Hkv) * HD;
  const float sl2 = scale * LOG2E;
  const int ntiles = (L + KV_TILE - 1) / KV_TILE;
  // ---- stage K and V once (round 3): the PRE-norm K rows and the V rows travel global -> LDS by DMA straight into the swizzled
  // row tiles (no staging registers: the r02 form held 64 VGPRs of chunks across the whole pass and spilled), K is then
  // normalised and rotated IN PLACE in LDS, with the cos / sin rows of the next pass requested while the current one is
  // computed.  (The r02 form also sat behind an s_waitcnt per 16-B load: each load lived under its own bounds branch.)
  constexpr int PER = (MAXT * 64 * (HD / 8) + NT_ - 1) / NT_;
  ATTN_STAMP(blockIdx.x + 6144, 0, 0);
  {
    const unsigned ldsK = lds_addr_of(Ks), ldsV = lds_addr_of(Vs);
    const bf16_t* kbase = qkv0 + (long)b * L * ld + (long)(Hq + hk) * HD;
    const bf16_t* vbase = kbase + (long)Hkv * HD;
    // 16 row blocks of 4 rows per tile, MAXT tiles; wave w moves blocks w, w + NW, ... (a block = one 1-KB DMA instruction)
    for (int blk = wave; blk < ntiles * 16; blk += NW) {
      const int t = blk >> 4, rb4 = (blk & 15) * 4, r = rb4 + (lane >> 4);
      int gr = t * KV_TILE + r; if (gr > L - 1) gr = L - 1;
      const int c = (lane & 15) ^ RowTile<HD>::swz(r), cv = (lane & 15) ^ RowTile<HD>::swz_tr(r);   // V is read transposed
      const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(t * RowTile<HD>::BYTES + rb4 * 256));
      dma16_asm(kbase + (long)gr * ld + c * 8, ldsK + off);
      dma16_asm(vbase + (long)gr * ld + cv * 8, ldsV + off);
    }
  }
  {
    const int c = tid & 15;                           // NT_ is a multiple of 16: a thread keeps its chunk column in every pass
    const float4 wa = *(const float4*)(kn_w + c * 8), wb = *(const float4*)(kn_w + c * 8 + 4);
    const float wk[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w};
    float4 cc[2][4];                                  // cos | sin rows of the current and the next pass
    auto rope_rows = [&](int i, float4* dst) {
      const int ch = tid + i * NT_, t = ch / (64 * (HD / 8)), w = ch % (64 * (HD / 8));
      const int krow = t * KV_TILE + w / (HD / 8), gr = krow > L - 1 ? L - 1 : krow;
      const int pp = pos ? pos[(long)b * L + gr] : gr;
      const float* cp = cosT + (long)pp * 64 + (c & 7) * 8;
      const float* sp = sinT + (long)pp * 64 + (c & 7) * 8;
      dst[0] = *(const float4*)cp; dst[1] = *(const float4*)(cp + 4); dst[2] = *(const float4*)sp; dst[3] = *(const float4*)(sp + 4);
    };
    rope_rows(0, cc[0]);
    for (int i = tid; i < ntiles * 64; i += NT_) Ms[i] = (i < L) ? (kmask ? kmask[(long)b * L + i] : 1) : 0;
    ATTN_STAMP(blockIdx.x + 6144, 1, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA instructions (and the loads above) have landed
    __syncthreads();                                  // ... everyone's
    ATTN_STAMP(blockIdx.x + 6144, 2, 0);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int ch = tid + i * NT_, t = ch / (64 * (HD / 8)), w = ch % (64 * (HD / 8));
      if (i + 1 < PER) rope_rows(i + 1, cc[(i + 1) & 1]);
      if (t < ntiles) {                               // whole 16-lane rows take the same branch
        const int r = w / (HD / 8);
        const int krow = t * KV_TILE + r, gr = krow > L - 1 ? L - 1 : krow;
        const long tok = (long)b * L + gr;
        char* kp = Ks + t * RowTile<HD>::BYTES + RowTile<HD>::off(r, c);
        const uint4 kin = *(const uint4*)kp;
        const uint4 vin = *(const uint4*)(Vs + t * RowTile<HD>::BYTES + RowTile<HD>::off_tr(r, c));
        const uint32_t u[4] = {kin.x, kin.y, kin.z, kin.w};
        float x[8], ss = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) { x[2 * k] = bf2f((bf16_t)(u[k] & 0xffff)); x[2 * k + 1] = bf2f((bf16_t)(u[k] >> 16)); }
#pragma unroll
        for (int e = 0; e < 8; ++e) ss += x[e] * x[e];
        ss = row_sum16(ss);
        const float rr = rsqrtf(ss / (float)HD + eps);
        const float4* q4 = cc[i & 1];
        const float cs[8] = {q4[0].x, q4[0].y, q4[0].z, q4[0].w, q4[1].x, q4[1].y, q4[1].z, q4[1].w};
        const float sn[8] = {q4[2].x, q4[2].y, q4[2].z, q4[2].w, q4[3].x, q4[3].y, q4[3].z, q4[3].w};
        float y[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float n = x[e] * rr * wk[e];
          const float np = row_xor8(n);               // the RoPE partner (dims d <-> d + 64) sits 8 lanes away
          y[e] = c < 8 ? n * cs[e] - np * sn[e] : n * cs[e] + np * sn[e];
        }
        const uint4 kv = make_uint4(pack2bf(y[0], y[1]), pack2bf(y[2], y[3]), pack2bf(y[4], y[5]), pack2bf(y[6], y[7]));
        *(uint4*)kp = kv;                             // in place: this thread is the only reader and writer of its chunk
        if (krow < L) {
          *(uint4*)(Ko + ((long)(b * Hkv + hk) * L + krow) * HD + c * 8) = kv;
          if (Vo) *(uint4*)(Vo + ((long)(b * Hkv + hk) * L + krow) * HD + c * 8) = vin;
          if (c == 0) rk[tok * Hkv + hk] = rr;
        }
      }
    }
  }
  ATTN_STAMP(blockIdx.x + 6144, 3, 0);
  __syncthreads();
  ATTN_STAMP(blockIdx.x + 6144, 4, 0);
  const bf16x8 ones = (bf16x8){0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};
  const bf16x8 zeros = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
  const bf16x8 one_row = l15 == 0 ? ones : zeros;     // row 0 of the extra block is all ones: it accumulates l = sum_k P[q, k]
  for (int pass = 0; pass < 2; ++pass) {
  const int chunk = pass == 0 ? wave : nchunk - 1 - wave;
  if (chunk >= nchunk || (pass == 1 && chunk < NW)) continue;          // wave-uniform
  const int h0 = PAIR ? hk * grp : hk * grp + chunk / cph;             // head of sub-tile 0 (PAIR: sub-tile s is head h0 + s)
  const int q0 = PAIR ? chunk * 16 : (chunk % cph) * 16 * QSUB;
  constexpr int RSTEP = PAIR ? 0 : 16;                                  // row offset between the sub-tiles
  bf16x8 qf[QSUB][HD / 32];
  float4 tc[2][2], ts[2][2];                                            // cos / sin rows of the current sub-tile (PAIR: of both)
  // (Round 3: the loads of a query sub-tile -- its 4 row chunks and the cos / sin rows of both dim halves -- are issued as ONE batch;
  // the r02 form waited for the row, then per dim half for eight dependent table loads.)
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) {
    const int h = PAIR ? h0 + sub : h0;
    const int qrow = q0 + sub * RSTEP + l15, qr = qrow > L - 1 ? L - 1 : qrow;
    const long tok = (long)b * L + qr;
    const bf16_t* src = qkv0 + tok * ld + (long)h * HD;
    uint4 raw[HD / 32];
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) raw[ks] = *(const uint4*)(src + ks * 32 + g * 8);
    if (!PAIR || sub == 0) {
      const int p = pos ? pos[tok] : qr;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int d0 = ks * 32 + g * 8;
        tc[ks][0] = *(const float4*)(cosT + (long)p * 64 + d0); tc[ks][1] = *(const float4*)(cosT + (long)p * 64 + d0 + 4);
        ts[ks][0] = *(const float4*)(sinT + (long)p * 64 + d0); ts[ks][1] = *(const float4*)(sinT + (long)p * 64 + d0 + 4);
      }
    }
    float x[HD / 32][8], ss = 0.f;
#pragma unroll
    for (int ks = 0; ks < HD / 32; ++ks) {
      const uint32_t u[4] = {raw[ks].x, raw[ks].y, raw[ks].z, raw[ks].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) { x[ks][2 * k] = bf2f((bf16_t)(u[k] & 0xffff)); x[ks][2 * k + 1] = bf2f((bf16_t)(u[k] >> 16)); }
#pragma unroll
      for (int e = 0; e < 8; ++e) ss += x[ks][e] * x[ks][e];
    }
    ss = group_sum(ss);
    const float rr = rsqrtf(ss / (float)HD + eps);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {                   // dims d = ks * 32 + g * 8 + e < 64 and their partners d + 64 (fragment ks + 2)
      const int d0 = ks * 32 + g * 8;
      const float4 c0 = tc[ks][0], c1 = tc[ks][1], s0 = ts[ks][0], s1 = ts[ks][1];
      const float4 wa = *(const float4*)(qn_w + d0), wb = *(const float4*)(qn_w + d0 + 4);
      const float4 wc = *(const float4*)(qn_w + 64 + d0), wd = *(const float4*)(qn_w + 64 + d0 + 4);
      const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float w1[8] = {wa.x, wa.y, wa.z, wa.w, wb.x, wb.y, wb.z, wb.w}, w2[8] = {wc.x, wc.y, wc.z, wc.w, wd.x, wd.y, wd.z, wd.w};
      float y1[8], y2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float n1 = x[ks][e] * rr * w1[e], n2 = x[ks + 2][e] * rr * w2[e];
        y1[e] = n1 * cs[e] - n2 * sn[e];
        y2[e] = n2 * cs[e] + n1 * sn[e];
      }
      union { bf16x8 v; uint4 u; } a, c;
      a.u = make_uint4(pack2bf(y1[0], y1[1]), pack2bf(y1[2], y1[3]), pack2bf(y1[4], y1[5]), pack2bf(y1[6], y1[7]));
      c.u = make_uint4(pack2bf(y2[0], y2[1]), pack2bf(y2[2], y2[3]), pack2bf(y2[4], y2[5]), pack2bf(y2[6], y2[7]));
      qf[sub][ks] = a.v; qf[sub][ks + 2] = c.v;
      if (qrow < L) {
        bf16_t* qdst = Qo + ((long)(b * Hq + h) * L + qrow) * HD;
        *(uint4*)(qdst + d0) = a.u;
        *(uint4*)(qdst + 64 + d0) = c.u;
      }
    }
    if (qrow < L && g == 0) rq[tok * Hq + h] = rr;
  }
  f32x4 o[QSUB][ND + 1];
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub)
#pragma unroll
    for (int i = 0; i <= ND; ++i) o[sub][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float m_run[QSUB];
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) m_run[sub] = NEG_BIG;
  const int my_tiles = min(ntiles, (q0 + (PAIR ? 16 : 16 * QSUB) - 1) / KV_TILE + 1);      // causal: keys beyond the wave's last query never count
  ATTN_STAMP(blockIdx.x + 6144, 5 + 3 * pass, 0);                                   // wave 0: queries staged (pass 0: 1 tile, pass 1: 3 tiles)
  for (int t = 0; t < my_tiles; ++t) {
    const int key0 = t * KV_TILE;
    const char* Kt = Ks + t * RowTile<HD>::BYTES;
    const char* Vt = Vs + t * RowTile<HD>::BYTES;
    f32x4 s[QSUB][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
      for (int sub = 0; sub < QSUB; ++sub) s[sub][kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        const bf16x8 a = *(const bf16x8*)(Kt + RowTile<HD>::off(kt * 16 + l15, ks * 4 + g));
#pragma unroll
        for (int sub = 0; sub < QSUB; ++sub)
          s[sub][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, qf[sub][ks], s[sub][kt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int sub = 0; sub < QSUB; ++sub) {
      const int qrow = q0 + sub * RSTEP + l15;
      const bool full = (key0 + KV_TILE <= L) && (kmask == nullptr) && (key0 + KV_TILE - 1 <= q0 + sub * RSTEP);
      if (!full) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
          const int4 mk = *(const int4*)(Ms + key0 + kt * 16 + g * 4);
          const int mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = key0 + kt * 16 + g * 4 + r;
            const bool v = (key <= qrow) & (mkv[r] != 0);
            s[sub][kt][r] = v ? s[sub][kt][r] : -INFINITY;
          }
        }
      }
      float mloc = max3(s[sub][0][0], s[sub][0][1], s[sub][0][2]);
      mloc = max3(mloc, s[sub][0][3], s[sub][1][0]);
      mloc = max3(mloc, s[sub][1][1], s[sub][1][2]);
      mloc = max3(mloc, s[sub][1][3], s[sub][2][0]);
      mloc = max3(mloc, s[sub][2][1], s[sub][2][2]);
      mloc = max3(mloc, s[sub][2][3], s[sub][3][0]);
      mloc = max3(mloc, s[sub][3][1], s[sub][3][2]);
      mloc = fmaxf(mloc, s[sub][3][3]);
      const float m_new = fmaxf(m_run[sub], group_max(mloc));
      if (__any(m_new != m_run[sub])) {
        const float alpha = __builtin_amdgcn_exp2f((m_run[sub] - m_new) * sl2);
#pragma unroll
        for (int i = 0; i <= ND; ++i) { o[sub][i][0] *= alpha; o[sub][i][1] *= alpha; o[sub][i][2] *= alpha; o[sub][i][3] *= alpha; }
        m_run[sub] = m_new;
      }
      const float mb = m_new * sl2;
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[sub][kt][r] = __builtin_amdgcn_exp2f(fmaf(s[sub][kt][r], sl2, -mb));
    }
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      bf16x8 pb[QSUB];
#pragma unroll
      for (int sub = 0; sub < QSUB; ++sub) pb[sub] = pack_p(s[sub][2 * kp], s[sub][2 * kp + 1]);
#pragma unroll
      for (int dt = 0; dt <= ND; ++dt) {
        const bf16x8 va = dt < ND ? read_colfrag_tr<HD>(Vt, dt, l15, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4) : one_row;
#pragma unroll
        for (int sub = 0; sub < QSUB; ++sub)
          o[sub][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb[sub], o[sub][dt], 0, 0, 0);
        if (dt & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  ATTN_STAMP(blockIdx.x + 6144, 6 + 3 * pass, 0);                                   // tile loop done
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) {
    const int h = PAIR ? h0 + sub : h0;
    const int qrow = q0 + sub * RSTEP + l15;
    const float l_run = __shfl(o[sub][ND][0], l15, 64);
    if (qrow < L) {
      const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
      bf16_t* orow = O + ((long)b * L + qrow) * ((long)Hq * HD) + (long)h * HD;
#pragma unroll
      for (int dt = 0; dt < ND; dt += 2) {
        const uint32_t x0 = pack2bf(o[sub][dt][0] * inv, o[sub][dt][1] * inv), x1 = pack2bf(o[sub][dt][2] * inv, o[sub][dt][3] * inv);
        const uint32_t y0 = pack2bf(o[sub][dt + 1][0] * inv, o[sub][dt + 1][1] * inv), y1 = pack2bf(o[sub][dt + 1][2] * inv, o[sub][dt + 1][3] * inv);
        const auto a = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
        const auto c = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
        *(uint4*)(orow + 16 * (dt + (g & 1)) + 8 * (g >> 1)) = make_uint4(a[0], c[0], a[1], c[1]);
      }
      if (LSE && g == 0) LSE[(long)(b * Hq + h) * L + qrow] = l_run > 0.f ? m_run[sub] * scale + __logf(l_run) : 1.0e30f;
    }
  }
  ATTN_STAMP(blockIdx.x + 6144, 7 + 3 * pass, 0);                                   // output stores issued
  }   // pass
#ifdef TA355_ATTN_STAMPS
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  ATTN_STAMP(blockIdx.x + 6144, 11, 0);                                             // ... and acknowledged
#endif
}

// ---- optional fused epilogue of the backward: the q|k|v post-processing backward (RoPE^T, per-head RMSNorm backward, head-major ->
// token-major) applied to the f32 accumulators, so dQ / dK / dV never travel through memory.  qkv0 == nullptr: plain head-major
// dQ / dK / dV as before (ta_lm_qkv_post_bwd then does this work; it is still needed when the q_norm / k_norm weights train).
// Math as lm_qkv_post_bwd_kernel (qkv_post.hip), TF:models/qwen3/modeling_qwen3.py:50-64,211-240.
struct QkvPostBwd {
  const bf16_t* qkv0;                // pre-norm q | k | v, token-major [B*L, (Hq + 2 Hkv) * 128]
  const float *rq, *rk;              // 1 / rms of every (token, head) row
  const float *qn_w, *kn_w;          // q_norm / k_norm weights [128]
  const float *cosT, *sinT;          // [positions][64]
  const int* pos;                    // position of token (b, l), or null = l
  bf16_t* dqkv;                      // out: token-major gradient of qkv0
};
// The workgroup's 64 rows x 128 columns of f32 accumulators are first laid out row-major in LDS (the tiles are free by then), then
// re-read with the producer kernels' mapping -- 8 lanes per row, lane j owning columns [8j, 8j+8) and [64+8j, 64+8j+8), RoPE partners
// in the same lane -- so that qkv0, the tables and the result move as 16-byte chunks of whole rows.  (A first version applied the
// same math in the accumulator layout, 8-byte accesses of 16 different rows per instruction: 20 us per layer instead of ~9.)
constexpr int QP_PITCH = 128 * 4 + 16;               // bytes per staged f32 row
__device__ __forceinline__ float oct_sum8(float v) { return row_sum8(v); }
// stage: lane (l15, g) of wave w writes row w * 16 + l15, columns dt * 16 + 4 g .. + 3 of acc[dt]
__device__ __forceinline__ void qkv_stage_f32(char* st, const f32x4* acc, int wave, int l15, int g) {
#pragma unroll
  for (int dt = 0; dt < 8; ++dt)
    *(float4*)(st + (wave * 16 + l15) * QP_PITCH + (dt * 16 + g * 4) * 4) = make_float4(acc[dt][0], acc[dt][1], acc[dt][2], acc[dt][3]);
}
// sec 0 / 1: RoPE^T + RMSNorm backward of q / k head `head`; sec 2: v, copied through.  row0 = first sequence row of the tile.
__device__ __forceinline__ void qkv_post_bwd_tile(const char* st, const QkvPostBwd& F, int sec, int b, int row0, int L, int head,
                                                  int Hq, int Hkv, int tid) {
  constexpr int HD = 128;
  const long ld = (long)(Hq + 2 * Hkv) * HD;
  const int Hs = sec == 0 ? Hq : Hkv, j = tid & 7;
  const long coff = (long)(sec == 0 ? head : (sec == 1 ? Hq + head : Hq + Hkv + head)) * HD;
  float w1[8], w2[8];
  if (sec < 2) {
    const float* nw = sec == 0 ? F.qn_w : F.kn_w;
    const float4 a = *(const float4*)(nw + 8 * j), b4 = *(const float4*)(nw + 8 * j + 4), c = *(const float4*)(nw + 64 + 8 * j), d = *(const float4*)(nw + 64 + 8 * j + 4);
    w1[0] = a.x; w1[1] = a.y; w1[2] = a.z; w1[3] = a.w; w1[4] = b4.x; w1[5] = b4.y; w1[6] = b4.z; w1[7] = b4.w;
    w2[0] = c.x; w2[1] = c.y; w2[2] = c.z; w2[3] = c.w; w2[4] = d.x; w2[5] = d.y; w2[6] = d.z; w2[7] = d.w;
  }
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int tl = (tid >> 3) + it * 32, l = row0 + tl;
    const bool live = l < L;
    const int lc = live ? l : L - 1;                  // clamped rows compute (the octet shuffles stay uniform) but never store
    const long tok = (long)b * L + lc;
    const float* sr = (const float*)(st + tl * QP_PITCH);
    float d1[8], d2[8];
    { const float4 a = *(const float4*)(sr + 8 * j), b4 = *(const float4*)(sr + 8 * j + 4), c = *(const float4*)(sr + 64 + 8 * j), d = *(const float4*)(sr + 64 + 8 * j + 4);
      d1[0] = a.x; d1[1] = a.y; d1[2] = a.z; d1[3] = a.w; d1[4] = b4.x; d1[5] = b4.y; d1[6] = b4.z; d1[7] = b4.w;
      d2[0] = c.x; d2[1] = c.y; d2[2] = c.z; d2[3] = c.w; d2[4] = d.x; d2[5] = d.y; d2[6] = d.z; d2[7] = d.w; }
    if (sec < 2) {
      const bf16_t* src = F.qkv0 + tok * ld + coff;
      const uint4 xa = *(const uint4*)(src + 8 * j), xb = *(const uint4*)(src + 64 + 8 * j);
      const uint32_t ua[4] = {xa.x, xa.y, xa.z, xa.w}, ub[4] = {xb.x, xb.y, xb.z, xb.w};
      float x1[8], x2[8];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        x1[2 * k] = bf2f((bf16_t)(ua[k] & 0xffff)); x1[2 * k + 1] = bf2f((bf16_t)(ua[k] >> 16));
        x2[2 * k] = bf2f((bf16_t)(ub[k] & 0xffff)); x2[2 * k + 1] = bf2f((bf16_t)(ub[k] >> 16));
      }
      const int p = F.pos ? F.pos[tok] : lc;
      const float4 c0 = *(const float4*)(F.cosT + (long)p * 64 + 8 * j), c1 = *(const float4*)(F.cosT + (long)p * 64 + 8 * j + 4);
      const float4 s0 = *(const float4*)(F.sinT + (long)p * 64 + 8 * j), s1 = *(const float4*)(F.sinT + (long)p * 64 + 8 * j + 4);
      const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sn[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float r = (sec == 0 ? F.rq : F.rk)[tok * Hs + head];
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dn1 = d1[e] * cs[e] + d2[e] * sn[e], dn2 = d2[e] * cs[e] - d1[e] * sn[e];     // RoPE^T
        x1[e] *= r; x2[e] *= r;                                                               // x-hat
        d1[e] = dn1 * w1[e]; d2[e] = dn2 * w2[e];
        dot += d1[e] * x1[e] + d2[e] * x2[e];
      }
      const float md = oct_sum8(dot) / (float)HD;
#pragma unroll
      for (int e = 0; e < 8; ++e) { d1[e] = r * (d1[e] - x1[e] * md); d2[e] = r * (d2[e] - x2[e] * md); }
    }
    if (live) {
      bf16_t* dst = F.dqkv + ((long)b * L + l) * ld + coff;
      *(uint4*)(dst + 8 * j) = make_uint4(pack2bf(d1[0], d1[1]), pack2bf(d1[2], d1[3]), pack2bf(d1[4], d1[5]), pack2bf(d1[6], d1[7]));
      *(uint4*)(dst + 64 + 8 * j) = make_uint4(pack2bf(d2[0], d2[1]), pack2bf(d2[2], d2[3]), pack2bf(d2[4], d2[5]), pack2bf(d2[6], d2[7]));
    }
  }
}


// ============================================================================ backward: dQ
// grid (q tiles, Hq, B).  dQ^T[d,q] = sum_key K^T[d,key] dS^T[key,q],  dS = P o (dP - Delta) * scale
template <int HD, bool CAUSAL>
__device__ __forceinline__ void attn_bwd_dq_body(char* smem, int block_id, const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                 const bf16_t* __restrict__ V, const bf16_t* __restrict__ KT,
                                                 const bf16_t* __restrict__ dO, long dO_stride,
                                                 const float* __restrict__ LSE, const float* __restrict__ Delta,
                                                 const int* __restrict__ kmask, bf16_t* __restrict__ dQ,
                                                 int B, int Hq, int Hkv, int L, int Lp, float scale, const QkvPostBwd& F) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int nq = (L + 63) / 64, grp = Hq / Hkv;
  int group, member;
  if (!decode_group(block_id, grp * nq, B * Hkv, group, member)) return;
  const int b = group / Hkv, hk = group % Hkv;
  const int h = hk * grp + member / nq, qt = member % nq;
  const int qrow = qt * 64 + wave * 16 + l15;
  const int qr = qrow < L ? qrow : L - 1;
  const bf16_t* Qb = Q + ((long)(b * Hq + h) * L) * HD;
  const bf16_t* Kb = K + ((long)(b * Hkv + hk) * L) * HD;
  // V == nullptr (round 6, the fused q|k|v form): V is read in place from the token-major q|k|v GEMM output (V is neither normalised
  // nor rotated, so the forward need not write a head-major copy: 49 KB less for each of its store-bound workgroups)
  const long ldq = (long)(Hq + 2 * Hkv) * HD;
  const bf16_t* Vb = V ? V + ((long)(b * Hkv + hk) * L) * HD : F.qkv0 + (long)b * L * ldq + (long)(Hq + Hkv + hk) * HD;
  const long v_rs = V ? HD : ldq;
  (void)KT; (void)Lp;
  const bf16_t* dOb = dO + (long)b * L * dO_stride + (long)h * HD;   // token-major rows
  const float sl2 = scale * LOG2E;
  bf16x8 qf[HD / 32], dof[HD / 32];
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks) {
    qf[ks] = *(const bf16x8*)(Qb + (long)qr * HD + ks * 32 + g * 8);
    dof[ks] = *(const bf16x8*)(dOb + (long)qr * dO_stride + ks * 32 + g * 8);
  }
  const float lse2 = LSE[(long)(b * Hq + h) * L + qr] * LOG2E;
  const float delta = Delta[(long)(b * Hq + h) * L + qr];
  f32x4 dq[HD / 16];
#pragma unroll
  for (int i = 0; i < HD / 16; ++i) dq[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int ntiles = (L + KV_TILE - 1) / KV_TILE;
  if (CAUSAL) { const int lim = qt + 1; if (lim < ntiles) ntiles = lim; }
  // Round 3: K / V tiles by DMA into a double buffer, the next tile issued before this tile's arithmetic (the kernel runs at
  // 2 waves per SIMD -- 216 VGPRs -- and load -> store -> barrier -> compute exposed the full fetch latency in every iteration)
  static_assert(HD == 128, "the DMA staging of the backward is written for head_dim 128");
  const unsigned lds0 = lds_addr_of(smem);
  int pm = 0;
  auto issue = [&](int t) {
    const unsigned dst = lds0 + (t & 1) * BWD_BUF;
    dma_rowtile128(Kb, HD, t * KV_TILE, L, dst, wave, lane);
    dma_rowtile128(Vb, v_rs, t * KV_TILE, L, dst + RowTile<HD>::BYTES, wave, lane);
    if (tid < 64) { const int kk = t * KV_TILE + tid; pm = (kk < L) ? (kmask ? kmask[(long)b * L + kk] : 1) : 0; }
  };
  ATTN_STAMP(gridDim.x - 1 - blockIdx.x + 4096, 0, 0);   // (dq blocks are stamped from 4096 up, by their distance from the grid's end)
  if (ntiles > 0) issue(0);
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * KV_TILE;
    char* Ks = smem + (t & 1) * BWD_BUF;
    char* Vs = Ks + RowTile<HD>::BYTES;
    int* Ms = (int*)(Vs + RowTile<HD>::BYTES);
    if (tid < 64) Ms[tid] = pm;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                   // tile t has landed; every wave is done with tile t - 1 (the other buffer)
    if (t == 0) ATTN_STAMP(gridDim.x - 1 - blockIdx.x + 4096, 1, 0);
    if (t + 1 < ntiles) issue(t + 1);
    f32x4 s[4], dp[4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      s[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[kt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < HD / 32; ++ks) {
        const int off = RowTile<HD>::off_tr(kt * 16 + l15, ks * 4 + g);
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Ks + off), qf[ks], s[kt], 0, 0, 0);
        dp[kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Vs + off), dof[ks], dp[kt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int4 mk = *(const int4*)(Ms + kt * 16 + g * 4);
      const int mkv[4] = {mk.x, mk.y, mk.z, mk.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = key0 + kt * 16 + g * 4 + r;
        bool v = (mkv[r] != 0) && (qrow < L);
        if (CAUSAL) v = v && (key <= qrow);
        const float p = v ? __builtin_amdgcn_exp2f(s[kt][r] * sl2 - lse2) : 0.f;
        s[kt][r] = p * (dp[kt][r] - delta) * scale;
      }
    }
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      const bf16x8 dsb = pack_p(s[2 * kp], s[2 * kp + 1]);
#pragma unroll
      for (int dt = 0; dt < HD / 16; ++dt) {
        const bf16x8 ka = read_colfrag_tr<HD>(Ks, dt, l15, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4);
        dq[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka, dsb, dq[dt], 0, 0, 0);
      }
    }
  }
  ATTN_STAMP(gridDim.x - 1 - blockIdx.x + 4096, 2, 0);
  __syncthreads();                                     // the tiles are free for the fused epilogue's image
  if (F.qkv0) {
    if constexpr (HD == 128) {
      qkv_stage_f32(smem, dq, wave, l15, g);                 // the last loop iteration ended with a barrier: the tiles are free
      __syncthreads();
      qkv_post_bwd_tile(smem, F, 0, b, qt * 64, L, h, Hq, Hkv, tid);
    }
#ifdef TA355_ATTN_STAMPS
    ATTN_STAMP(gridDim.x - 1 - blockIdx.x + 4096, 3, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_STAMP(gridDim.x - 1 - blockIdx.x + 4096, 4, 0);
    if (threadIdx.x == 0) { g_attn_stamps[(long)(gridDim.x - 1 - blockIdx.x + 4096) * 16 + 5] = ntiles; g_attn_stamps[(long)(gridDim.x - 1 - blockIdx.x + 4096) * 16 + 6] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20); }
#endif
    return;
  }
  if (qrow < L) {
    bf16_t* drow = dQ + ((long)(b * Hq + h) * L + qrow) * HD;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      uint2 w; w.x = pack2bf(dq[dt][0], dq[dt][1]); w.y = pack2bf(dq[dt][2], dq[dt][3]);
      *(uint2*)(drow + dt * 16 + g * 4) = w;
    }
  }
}

// ============================================================================ backward: dK, dV
// grid (key tiles, Hkv, B); loops over the Hq/Hkv query heads of the group and over query tiles.
//   dV^T[d,key] += dO^T[d,q] P[q,key]      dK^T[d,key] += Q^T[d,q] dS[q,key]
template <int HD, bool CAUSAL>
__device__ __forceinline__ void attn_bwd_dkv_body(char* smem, int block_id, const bf16_t* __restrict__ Q, const bf16_t* __restrict__ QT,
                                                  const bf16_t* __restrict__ K, const bf16_t* __restrict__ V,
                                                  const bf16_t* __restrict__ dO, long dO_stride,
                                                  const bf16_t* __restrict__ dOT,
                                                  const float* __restrict__ LSE, const float* __restrict__ Delta,
                                                  const int* __restrict__ kmask, bf16_t* __restrict__ dK,
                                                  bf16_t* __restrict__ dV, int B, int Hq, int Hkv, int L, int Lp,
                                                  float scale, const QkvPostBwd& F) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  const int grp = Hq / Hkv;
  int group, kt_idx;
  if (!decode_group(block_id, (L + 63) / 64, B * Hkv, group, kt_idx)) return;
  const int b = group / Hkv, hk = group % Hkv;
  const int krow = kt_idx * 64 + wave * 16 + l15;
  const int kr = krow < L ? krow : L - 1;
  const bf16_t* Kb = K + ((long)(b * Hkv + hk) * L) * HD;
  const long ldq = (long)(Hq + 2 * Hkv) * HD;              // V == nullptr: V rows in place in the q|k|v GEMM output (see the dQ body)
  const bf16_t* Vb = V ? V + ((long)(b * Hkv + hk) * L) * HD : F.qkv0 + (long)b * L * ldq + (long)(Hq + Hkv + hk) * HD;
  const long v_rs = V ? HD : ldq;
  const float sl2 = scale * LOG2E;
  bf16x8 kf[HD / 32], vf[HD / 32];
#pragma unroll
  for (int ks = 0; ks < HD / 32; ++ks) {
    kf[ks] = *(const bf16x8*)(Kb + (long)kr * HD + ks * 32 + g * 8);
    vf[ks] = *(const bf16x8*)(Vb + (long)kr * v_rs + ks * 32 + g * 8);
  }
  const bool kvalid = (krow < L) && (kmask ? kmask[(long)b * L + kr] != 0 : true);
  f32x4 dk[HD / 16], dv[HD / 16];
#pragma unroll
  for (int i = 0; i < HD / 16; ++i) { dk[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; dv[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
  const int nq = (L + 63) / 64;
  const int qt_begin = CAUSAL ? kt_idx : 0;
  // Round 3: Q / dO tiles by DMA into a double buffer; the (head, query tile) pairs of this key tile form ONE sequence of
  // iterations, the next one's tiles are in flight while this one computes
  static_assert(HD == 128, "the DMA staging of the backward is written for head_dim 128");
  const unsigned lds0 = lds_addr_of(smem);
  float pl = 1.0e30f, pd = 0.f;
  int it = 0;
  auto issue = [&](int hh, int qt, int buf) {
    const int h = hk * grp + hh, q0 = qt * 64;
    const unsigned dst = lds0 + buf * BWD_BUF;
    dma_rowtile128(Q + ((long)(b * Hq + h) * L) * HD, HD, q0, L, dst, wave, lane);
    dma_rowtile128(dO + (long)b * L * dO_stride + (long)h * HD, dO_stride, q0, L, dst + RowTile<HD>::BYTES, wave, lane);
    // (raw values only: any arithmetic on them here makes the compiler wait for the loads right behind the DMA issues -- and with
    // the in-order counter for every DMA of the tile just requested, which serialised the whole prefetch; round-3 ISA reading)
    pl = 1.0e30f; pd = 0.f;
    if (tid < 64 && q0 + tid < L) {
      pl = LSE[(long)(b * Hq + h) * L + q0 + tid];
      pd = Delta[(long)(b * Hq + h) * L + q0 + tid];
    }
  };
  ATTN_STAMP(blockIdx.x, 0, 0);
  if (grp > 0 && qt_begin < nq) issue(0, qt_begin, 0);
  for (int hh = 0; hh < grp; ++hh) {
    for (int qt = qt_begin; qt < nq; ++qt, ++it) {
      const int q0 = qt * 64;
      char* Qs = smem + (it & 1) * BWD_BUF;
      char* dOs = Qs + RowTile<HD>::BYTES;
      float* Ls = (float*)(dOs + RowTile<HD>::BYTES);  // [64] lse * log2e
      float* Ds = Ls + 64;                             // [64] delta
      if (tid < 64) { Ls[tid] = pl * LOG2E; Ds[tid] = pd; }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (it == 0) ATTN_STAMP(blockIdx.x, 1, 0);
      if (qt + 1 < nq) issue(hh, qt + 1, (it + 1) & 1);
      else if (hh + 1 < grp) issue(hh + 1, qt_begin, (it + 1) & 1);
      f32x4 s[4], dp[4];
#pragma unroll
      for (int qs = 0; qs < 4; ++qs) {
        s[qs] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dp[qs] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < HD / 32; ++ks) {
          const int off = RowTile<HD>::off_tr(qs * 16 + l15, ks * 4 + g);
          s[qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(Qs + off), kf[ks], s[qs], 0, 0, 0);
          dp[qs] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(dOs + off), vf[ks], dp[qs], 0, 0, 0);
        }
      }
      f32x4 ds[4];
#pragma unroll
      for (int qs = 0; qs < 4; ++qs) {
        const float4 ls = *(const float4*)(Ls + qs * 16 + g * 4);
        const float4 dl = *(const float4*)(Ds + qs * 16 + g * 4);
        const float lsv[4] = {ls.x, ls.y, ls.z, ls.w};
        const float dlv[4] = {dl.x, dl.y, dl.z, dl.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = q0 + qs * 16 + g * 4 + r;
          bool v = kvalid && (q < L);
          if (CAUSAL) v = v && (krow <= q);
          const float p = v ? __builtin_amdgcn_exp2f(s[qs][r] * sl2 - lsv[r]) : 0.f;
          s[qs][r] = p;
          ds[qs][r] = p * (dp[qs][r] - dlv[r]) * scale;
        }
      }
#pragma unroll
      for (int qp = 0; qp < 2; ++qp) {
        const bf16x8 pb = pack_p(s[2 * qp], s[2 * qp + 1]);
        const bf16x8 dsb = pack_p(ds[2 * qp], ds[2 * qp + 1]);
#pragma unroll
        for (int dt = 0; dt < HD / 16; ++dt) {
          const int c0 = (2 * qp) * 16 + g * 4, c1 = (2 * qp + 1) * 16 + g * 4;
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(read_colfrag_tr<HD>(dOs, dt, l15, c0, c1), pb, dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(read_colfrag_tr<HD>(Qs, dt, l15, c0, c1), dsb, dk[dt], 0, 0, 0);
        }
      }
    }
  }
  ATTN_STAMP(blockIdx.x, 2, 0);
  __syncthreads();                                     // the tiles are free for the fused epilogue's image
  if (F.qkv0) {
    if constexpr (HD == 128) {
      qkv_stage_f32(smem, dk, wave, l15, g);
      __syncthreads();
      qkv_post_bwd_tile(smem, F, 1, b, kt_idx * 64, L, hk, Hq, Hkv, tid);
      __syncthreads();
      qkv_stage_f32(smem, dv, wave, l15, g);
      __syncthreads();
      qkv_post_bwd_tile(smem, F, 2, b, kt_idx * 64, L, hk, Hq, Hkv, tid);
    }
#ifdef TA355_ATTN_STAMPS
    ATTN_STAMP(blockIdx.x, 3, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ATTN_STAMP(blockIdx.x, 4, 0);
    if (threadIdx.x == 0) { g_attn_stamps[(long)blockIdx.x * 16 + 5] = it; g_attn_stamps[(long)blockIdx.x * 16 + 6] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20); }
#endif
    return;
  }
  if (krow < L) {
    bf16_t* kro = dK + ((long)(b * Hkv + hk) * L + krow) * HD;
    bf16_t* vro = dV + ((long)(b * Hkv + hk) * L + krow) * HD;
#pragma unroll
    for (int dt = 0; dt < HD / 16; ++dt) {
      uint2 w; w.x = pack2bf(dk[dt][0], dk[dt][1]); w.y = pack2bf(dk[dt][2], dk[dt][3]);
      *(uint2*)(kro + dt * 16 + g * 4) = w;
      uint2 u; u.x = pack2bf(dv[dt][0], dv[dt][1]); u.y = pack2bf(dv[dt][2], dv[dt][3]);
      *(uint2*)(vro + dt * 16 + g * 4) = u;
    }
  }
}

// One launch for both halves of the backward: blocks [0, n_dkv) run the dK / dV body, the rest the dQ body.  The two
// are independent (both only read Q, K, V, dO), so a single grid lets the dQ workgroups fill the CUs while the longer
// dK / dV ones drain, without a second stream or a kernel boundary in between.  The heavier dK / dV blocks go first.
template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_bwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ QT,
                                                       const bf16_t* __restrict__ K, const bf16_t* __restrict__ KT,
                                                       const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO, long dO_stride,
                                                       const bf16_t* __restrict__ dOT, const float* __restrict__ LSE,
                                                       const float* __restrict__ Delta, const int* __restrict__ kmask,
                                                       bf16_t* __restrict__ dQ, bf16_t* __restrict__ dK, bf16_t* __restrict__ dV,
                                                       int B, int Hq, int Hkv, int L, int Lp, float scale, int n_dkv,
                                                       const QkvPostBwd F) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.x < n_dkv)
    attn_bwd_dkv_body<HD, CAUSAL>(smem, blockIdx.x, Q, QT, K, V, dO, dO_stride, dOT, LSE, Delta, kmask, dK, dV, B, Hq, Hkv, L, Lp, scale, F);
  else
    attn_bwd_dq_body<HD, CAUSAL>(smem, blockIdx.x - n_dkv, Q, K, V, KT, dO, dO_stride, LSE, Delta, kmask, dQ, B, Hq, Hkv, L, Lp, scale, F);
}


// ----------------------------------------------------------------------------- C-ABI
template <int HD> static size_t fwd_lds() { return RowTile<HD>::BYTES + (HD + 16) * CT_STRIDE + 64 * 4; }

extern "C" int ta_attention_fwd(const void* Q, const void* K, const void* VT, void* O, float* LSE, const int* kmask,
                                int B, int Hq, int Hkv, int L, int Lp, int head_dim, int causal, float scale,
                                hipStream_t st) {
  return ta_attention_fwd_ex(Q, K, VT, O, LSE, kmask, B, Hq, Hkv, L, Lp, head_dim, causal, scale, nullptr, st);
}

extern "C" int ta_attention_fwd_ex(const void* Q, const void* K, const void* VT, void* O, float* LSE, const int* kmask,
                                   int B, int Hq, int Hkv, int L, int Lp, int head_dim, int causal, float scale,
                                   const ta_attn_layout* lay_in, hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  if (Hq % Hkv || Lp % 64 || Lp < L) return TA_ERR_ARG;
  // LM with a short prompt: the whole sequence of a (clip, kv head) lives in LDS, one workgroup serves the GQA group
  if (!lay_in) {
    const int grp = Hq / Hkv;
    if (head_dim == 128 && causal && L <= 192 && grp * ((L + 31) / 32) <= 12) {
      constexpr int MAXT = 3, QS = 2, NW = 6;          // 6 waves x 2 passes x 32 queries = 384 query rows per workgroup
      const size_t lds = MAXT * (RowTile<128>::BYTES + (128 + 16) * CT_STRIDE) + MAXT * 64 * 4;
      static bool attr = false;
      if (!attr) { (void)hipFuncSetAttribute((const void*)attn_fwd_gqa_kernel<128, MAXT, QS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
      TA_LAUNCH((attn_fwd_gqa_kernel<128, MAXT, QS, NW>), dim3(B * Hkv), dim3(NW * 64), lds, st, (const bf16_t*)Q, (const bf16_t*)K,
                (const bf16_t*)VT, (bf16_t*)O, LSE, kmask, B, Hq, Hkv, L, Lp, scale);
      TA_CHECK_LAUNCH();
      return TA_OK;
    }
  }
  const long hd = head_dim;
  ta_attn_layout lay = {Hq * L * hd, L * hd, hd, Hkv * L * hd, L * hd, hd, Hkv * hd * Lp, hd * Lp, Lp};   // head-major images
  if (lay_in) lay = *lay_in;
  // Q / K rows are read as 16-B chunks; V^T chunks may sit at any element offset (narrower loads)
  if (((lay.q_bs | lay.q_hs | lay.q_rs | lay.k_bs | lay.k_hs | lay.k_rs) & 7) || (((uintptr_t)Q | (uintptr_t)K) & 15)) return TA_ERR_ARG;
  const long vor = lay.v_bs | lay.v_hs | lay.v_rs;
  const uintptr_t vp = (uintptr_t)VT;
  int valign = (!(vor & 7) && !(vp & 15)) ? 0 : ((!(vor & 3) && !(vp & 7)) ? 1 : 2);
  // encoder (hd 64, S = 500, non-causal): 128 query rows per workgroup; LM (hd 128, short causal L): 64
  const int qsub = (head_dim == 64) ? 2 : 1;
  dim3 grid(grouped_grid((Hq / Hkv) * ta_cdiv(L, 64 * qsub), B * Hkv)), blk(256);
#define FWD(HD_, C_)                                                                                              \
  TA_LAUNCH((attn_fwd_kernel<HD_, C_, (HD_ == 64 ? 2 : 1)>), grid, blk, fwd_lds<HD_>(), st, (const bf16_t*)Q, (const bf16_t*)K, \
                     (const bf16_t*)VT, (bf16_t*)O, LSE, kmask, B, Hq, Hkv, L, Lp, scale, lay, valign)
  if (head_dim == 64 && !causal) FWD(64, false);
  else if (head_dim == 64 && causal) FWD(64, true);
  else if (head_dim == 128 && causal) FWD(128, true);
  else if (head_dim == 128 && !causal) FWD(128, false);
  else return TA_ERR_ARG;
  TA_CHECK_LAUNCH();
  return TA_OK;
}

static int attention_bwd_launch(const void* Q, const void* K, const void* V, const void* dO, long dO_stride, const float* LSE,
                                const float* Delta, const int* kmask, void* dQ, void* dK, void* dV, int B, int Hq, int Hkv, int L,
                                int Lp, int head_dim, int causal, float scale, const QkvPostBwd& F, hipStream_t st) {
  if (!Delta) return TA_ERR_ARG;
  if (B <= 0 || L <= 0) return TA_OK;
  if (Hq % Hkv || Lp % 64 || Lp < L || head_dim != 128) return TA_ERR_ARG;
  constexpr int HD = 128;
  // K^T / Q^T / dO^T fragments are read transposed out of the row tiles; the fused epilogue re-uses the space for a [64][128] f32 image
  const size_t lds_kv = (2 * (size_t)BWD_BUF > 64 * (size_t)QP_PITCH) ? 2 * (size_t)BWD_BUF : 64 * (size_t)QP_PITCH;      // two staging buffers
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<HD, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    (void)hipFuncSetAttribute((const void*)attn_bwd_kernel<HD, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_kv);
    attr_done = true;
  }
  const int n_dq = grouped_grid((Hq / Hkv) * ta_cdiv(L, 64), B * Hkv), n_dkv = grouped_grid(ta_cdiv(L, 64), B * Hkv);
  dim3 grid(n_dq + n_dkv), blk(256);
  // One launch for both halves: the heavier dK / dV blocks first, the dQ blocks fill the CUs while they drain.  (Measured alternatives,
  // removed in round 5: two launches -- +0.47 ms per step, r02; the dK/dV and dQ workgroups of one (clip, kv head) interleaved on
  // one XCD -- 45.42 vs 45.29 ms per step: the heavier blocks first balance the tail better than the shared L2 lines help.)
  const bf16_t* nul = nullptr;
#define BWD(C_)                                                                                                                     \
  TA_LAUNCH((attn_bwd_kernel<HD, C_>), grid, blk, lds_kv, st, (const bf16_t*)Q, nul, (const bf16_t*)K, nul, (const bf16_t*)V,       \
            (const bf16_t*)dO, dO_stride, nul, LSE, Delta, kmask, (bf16_t*)dQ, (bf16_t*)dK, (bf16_t*)dV, B, Hq, Hkv, L, Lp, scale, n_dkv, F)
  if (causal) BWD(true);
  else BWD(false);
#undef BWD
  TA_CHECK_LAUNCH();
  return TA_OK;
}

// ta_lm_qkv_post_fwd + ta_attention_fwd in one launch for the LM's short causal sequences (head_dim 128, L <= 192, GQA group x
// ceil(L / 32) <= 12): reads the pre-norm q | k | v GEMM output, writes O, LSE and the backward's operands (Q, K, V head-major,
// rq, rk).  Returns TA_ERR_ARG when the shape is outside that envelope (callers fall back to the two-kernel path).
extern "C" int ta_attention_fwd_qkv(const void* qkv0, const float* qn_w, const float* kn_w, const float* cosT, const float* sinT,
                                    const int* pos, void* Q, void* K, void* V, float* rq, float* rk, void* O, float* LSE,
                                    const int* kmask, int B, int Hq, int Hkv, int L, float scale, float eps, hipStream_t st) {
  if (B <= 0 || L <= 0) return TA_OK;
  if (!qkv0 || !Q || !K || !rq || !rk || !O || Hkv <= 0 || Hq % Hkv) return TA_ERR_ARG;     // V may be NULL: no head-major copy of V
  const int grp = Hq / Hkv;
  if (L > 192 || grp * ((L + 31) / 32) > 12) return TA_ERR_ARG;
  constexpr int MAXT = 3, QS = 2, NW = 6;
  const size_t lds = 2 * MAXT * RowTile<128>::BYTES + MAXT * 64 * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute((const void*)attn_fwd_gqa_qkv_kernel<128, MAXT, QS, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    (void)hipFuncSetAttribute((const void*)attn_fwd_gqa_qkv_kernel<128, MAXT, QS, NW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr = true;
  }
  if (grp == 2)     // Qwen3 (16 / 8): both heads of the group on the same 16 rows per wave (one fetch of the rows' cos / sin entries)
    TA_LAUNCH((attn_fwd_gqa_qkv_kernel<128, MAXT, QS, NW, true>), dim3(B * Hkv), dim3(NW * 64), lds, st, (const bf16_t*)qkv0, qn_w, kn_w, cosT, sinT,
              pos, (bf16_t*)Q, (bf16_t*)K, (bf16_t*)V, rq, rk, (bf16_t*)O, LSE, kmask, B, Hq, Hkv, L, scale, eps);
  else
    TA_LAUNCH((attn_fwd_gqa_qkv_kernel<128, MAXT, QS, NW>), dim3(B * Hkv), dim3(NW * 64), lds, st, (const bf16_t*)qkv0, qn_w, kn_w, cosT, sinT,
              pos, (bf16_t*)Q, (bf16_t*)K, (bf16_t*)V, rq, rk, (bf16_t*)O, LSE, kmask, B, Hq, Hkv, L, scale, eps);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

extern "C" int ta_attention_bwd(const void* Q, const void* QT, const void* K, const void* KT, const void* V,
                                const void* dO, long dO_stride, const void* dOT, const float* LSE, const float* Delta,
                                const int* kmask, void* dQ, void* dK, void* dV, int B, int Hq, int Hkv, int L, int Lp,
                                int head_dim, int causal, float scale, hipStream_t st) {
  (void)QT; (void)KT; (void)dOT;
  QkvPostBwd F = {};
  return attention_bwd_launch(Q, K, V, dO, dO_stride, LSE, Delta, kmask, dQ, dK, dV, B, Hq, Hkv, L, Lp, head_dim, causal, scale, F, st);
}

// The same with the q|k|v post-processing backward fused into the epilogue: writes d(qkv0) token-major [B*L, (Hq + 2 Hkv) * 128]
// directly (RoPE^T, per-head RMSNorm backward of q and k, v copied through); no head-major dQ / dK / dV.  Frozen q_norm / k_norm only.
extern "C" int ta_attention_bwd_qkv(const void* Q, const void* K, const void* V, const void* dO, long dO_stride, const float* LSE,
                                    const float* Delta, const int* kmask, const void* qkv0, const float* rq, const float* rk,
                                    const float* qn_w, const float* kn_w, const float* cosT, const float* sinT, const int* pos,
                                    void* dqkv, int B, int Hq, int Hkv, int L, int Lp, int head_dim, int causal, float scale,
                                    hipStream_t st) {
  if (!qkv0 || !dqkv || !rq || !rk || !qn_w || !kn_w || !cosT || !sinT) return TA_ERR_ARG;
  QkvPostBwd F = {(const bf16_t*)qkv0, rq, rk, qn_w, kn_w, cosT, sinT, pos, (bf16_t*)dqkv};
  return attention_bwd_launch(Q, K, V, dO, dO_stride, LSE, Delta, kmask, nullptr, nullptr, nullptr, B, Hq, Hkv, L, Lp, head_dim, causal,
                              scale, F, st);
}
