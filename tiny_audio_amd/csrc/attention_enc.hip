// ta355 GLM-ASR encoder self-attention, round-3 form (gfx950): non-causal, no mask, head_dim 64
// (TF:models/glmasr/modeling_glmasr.py:187-217), straight from the token-major q|k|v GEMM output [M, 3H].
//
// What changed against attn_fwd_kernel<64,false,2> (attention.hip; 87 us per layer at B = 32, S = 500, 0.19 of the MFMA peak):
//   * K and V tiles travel global -> LDS by DMA (global_load_lds_dwordx4) into a double buffer: no staging registers (the old
//     kernel kept the next K tile in SCRATCH -- 158 VGPRs at 3 waves / SIMD -- and waited for its global loads right after
//     issuing them), ONE barrier per key tile instead of two.
//   * V is read as a ROW tile [key][d] with ds_read_b64_tr_b16, so the producer writes no V^T image: the encoder's V^T GEMM and
//     the slack zeroing behind it are gone, q | k | v come out of ONE GEMM (N = 3H = 12 x 320 tiles).
//   * the softmax runs in base 2 on PRE-SCALED scores: head_dim^-0.5 * log2(e) is folded into the q rows of the weight image,
//     and the running maximum enters the QK^T MFMAs as their C operand (s' = k.q - m), so the per-score VALU work is
//     1 v_exp + 1/2 v_max3 + 1/2 v_cvt_pk -- no multiply-subtract pass, no accumulator zeroing.  O and l are rescaled only when
//     a tile raises some row's maximum by more than 2^8 (deferred rescale); the row sums ride on an all-ones MFMA block.
//   * the output tile is staged through LDS and stored as whole 128-byte rows.
// Per 64-key x 32-query tile and wave: 36 MFMAs (16 QK^T, 16 PV, 4 row sums), 32 v_exp, 16 v_max3, 16 v_cvt_pk, 8 ds_read_b128,
// 16 ds_read_b64_tr_b16.
#include <cstdlib>
#include <type_traits>
#include "common.h"
#include "../../include/ta355.h"

namespace {
constexpr int HD = 64;                 // head dim
constexpr int KT = 64;                 // keys per tile
constexpr int TILE = KT * HD * 2;      // 8 KB: one K or V tile, rows of 128 B
constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_THR = 8.0f;

// K rows: 16-B chunk index XOR ((row >> 1) & 7): conflict-free ds_read_b128 fragments (the GEMM's scheme).
// V rows: chunk index XOR 2 * ((row >> 1) & 3): the transposing read of one 16-lane group covers 4 rows x 32 B, two groups
// (8 rows) are serviced together; the even XOR keeps each row's two chunks adjacent and moves the four row pairs apart.
__device__ __forceinline__ int kswz(int r) { return (r >> 1) & 7; }
__device__ __forceinline__ int vswz(int r) { return ((r >> 1) & 3) << 1; }

// 16 B per lane global -> LDS; `lds` is the wave-uniform LDS byte address (the hardware adds lane * 16).  Inline assembly on
// purpose: behind the builtin the compiler tracks the DMA as an LDS store that may alias every later LDS read and drains vmcnt
// in front of the transposing V reads of the SAME tile iteration -- i.e. it waits for the NEXT tile's DMA right after issuing it.
// Here every wait for these loads is the explicit s_waitcnt vmcnt(0) at the top of a tile.
__device__ __forceinline__ void dma16(const void* g, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds), "v"(g) : "memory");
}

// XCD-aware block order (see attention.hip): the query tiles of one (clip, head) share an XCD, i.e. an L2.
__device__ __forceinline__ bool decode_group(int id, int gsz, int ngroups, int& group, int& member) {
  const int xcd = id & 7, w = id >> 3;
  member = w % gsz;
  group = (w / gsz) * 8 + xcd;
  return group < ngroups;
}
inline int grouped_grid(int gsz, int ngroups) { return gsz * ((ngroups + 7) / 8 * 8); }

__device__ __forceinline__ float max3(float a, float b, float c) { return __builtin_fmaxf(__builtin_fmaxf(a, b), c); }
__device__ __forceinline__ float group_max(float x) {                    // over the 4 lane groups that share a query column
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  x = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
  auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(q[0]), __uint_as_float(q[1]));
}
__device__ __forceinline__ bf16x8 pack_p(const f32x4& a, const f32x4& b) {
  union { bf16x8 v; uint32_t u[4]; } r;
  r.u[0] = pack2bf(a[0], a[1]); r.u[1] = pack2bf(a[2], a[3]);
  r.u[2] = pack2bf(b[0], b[1]); r.u[3] = pack2bf(b[2], b[3]);
  return r.v;
}
// V^T fragment (row d = dt * 16 + l15, keys [c0, c0 + 4) and [c1, c1 + 4)) read transposed out of the V ROW tile: lane n of a
// 16-lane group addresses the 8-byte chunk (row c + (n >> 2), columns dt * 16 + 4 (n & 3) ..) and receives rows c .. c + 3 of
// column dt * 16 + n (instruction mapping measured with scripts/probe/tr_probe.hip, see attention.hip).
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
__device__ __forceinline__ bf16x8 read_vfrag_tr(const char* vt, int dt, int l15, int c0, int c1) {
  const int dcol = dt * 16 + 4 * (l15 & 3);
  const int chunk = dcol >> 3, half = (dcol >> 2) & 1;
  const int r0 = c0 + (l15 >> 2), r1 = c1 + (l15 >> 2);
  const bf16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4_t*)(vt + r0 * 128 + ((chunk ^ vswz(r0)) << 4) + half * 8));
  const bf16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) bf16x4_t*)(vt + r1 * 128 + ((chunk ^ vswz(r1)) << 4) + half * 8));
  return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

// One workgroup = 64 * QSUB query rows of one (clip, head): 4 waves x QSUB sub-tiles of 16 rows.
template <int QSUB>
__global__ __launch_bounds__(256) void attn_enc_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ O, int B, int NH,
                                                           int L, long rs) {
  __shared__ __attribute__((aligned(16))) char smem[4 * TILE];       // [buffer][K | V]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, l15 = lane & 15;
  constexpr int QROWS = 64 * QSUB;
  const int nq = (L + QROWS - 1) / QROWS;
  int group, qt;
  if (!decode_group(blockIdx.x, nq, B * NH, group, qt)) return;
  const int b = group / NH, h = group % NH;
  const int q0 = qt * QROWS + wave * 16 * QSUB;                       // first query row of this wave
  const int Hd = NH * HD;
  const bf16_t* Qb = qkv + (long)b * L * rs + h * HD;                 // q | k | v of this head: columns h*64 of each third
  const bf16_t* Kb = Qb + Hd;
  const int ntiles = (L + KT - 1) / KT;

  // DMA of key tile t into buffer `buf`: wave w moves rows 16 w .. 16 w + 15 of K and of V (two 1-KB instructions each);
  // lane -> (row, LDS chunk position), the XOR swizzle is applied to the SOURCE chunk (the LDS side is lane-linear).
  // src[] walks down the token rows one tile per call (K rows i = 0, 1, then V rows i = 0, 1); only a ragged last tile
  // recomputes its rows (clamped to the clip's last token: finite duplicates, masked in the tile body).
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem + wave * 2048);
  const bf16_t* src[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 8 + (lane >> 3);
    src[i] = Kb + (long)r * rs + (((lane & 7) ^ kswz(r)) << 3);
    src[2 + i] = Kb + Hd + (long)r * rs + (((lane & 7) ^ vswz(r)) << 3);
  }
  const long tstride = (long)KT * rs;
  auto dma = [&](int t, int buf) {
    const unsigned dst = lds0 + buf * (2 * TILE);
    if ((t + 1) * KT <= L) {
      dma16(src[0], dst); dma16(src[2], dst + TILE); dma16(src[1], dst + 1024); dma16(src[3], dst + TILE + 1024);
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int key = t * KT + (wave * 2 + i) * 8 + (lane >> 3);
        const long back = key < L ? 0 : (long)(L - 1 - key) * rs;       // <= 0: up to the clip's last row
        dma16(src[i] + back, dst + i * 1024); dma16(src[2 + i] + back, dst + TILE + i * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) src[j] += tstride;
  };
  dma(0, 0);

  bf16x8 qf[QSUB][2];
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) {
    int qr = q0 + sub * 16 + l15; qr = qr < L ? qr : L - 1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[sub][ks] = *(const bf16x8*)(Qb + (long)qr * rs + ks * 32 + g * 8);
  }
  f32x4 o[QSUB][5];                                                   // O^T blocks 0..3 (d), block 4 = row sums
  f32x4 negm[QSUB];                                                   // -running maximum, the C operand of the QK^T MFMAs
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) {
#pragma unroll
    for (int i = 0; i < 5; ++i) o[sub][i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    negm[sub] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const bf16x8 ones = {0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80, 0x3f80};

  auto tile = [&](const int t, auto masked_tag) {
    constexpr bool MASKED = decltype(masked_tag)::value;
    const char* Ks = smem + (t & 1) * (2 * TILE);
    const char* Vs = Ks + TILE;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's share of tile t has landed
    __syncthreads();                                                  // ... everyone's has; everyone is done with tile t-1
    if (t + 1 < ntiles) dma(t + 1, (t + 1) & 1);
    // ---- S'^T = K Q^T - m for the wave's query sub-tiles
    f32x4 s[QSUB][4];
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
      const int kr = kt * 16 + l15;
      const bf16x8 a0 = *(const bf16x8*)(Ks + kr * 128 + (((0 + g) ^ kswz(kr)) << 4));
      const bf16x8 a1 = *(const bf16x8*)(Ks + kr * 128 + (((4 + g) ^ kswz(kr)) << 4));
#pragma unroll
      for (int sub = 0; sub < QSUB; ++sub) {
        s[sub][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, qf[sub][0], negm[sub], 0, 0, 0);
        s[sub][kt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, qf[sub][1], s[sub][kt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int sub = 0; sub < QSUB; ++sub) {
      if constexpr (MASKED) {
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (t * KT + kt * 16 + g * 4 + r >= L) s[sub][kt][r] = NEG_BIG;
      }
      float mx = max3(s[sub][0][0], s[sub][0][1], s[sub][0][2]);
      mx = max3(mx, s[sub][0][3], s[sub][1][0]);
      mx = max3(mx, s[sub][1][1], s[sub][1][2]);
      mx = max3(mx, s[sub][1][3], s[sub][2][0]);
      mx = max3(mx, s[sub][2][1], s[sub][2][2]);
      mx = max3(mx, s[sub][2][3], s[sub][3][0]);
      mx = max3(mx, s[sub][3][1], s[sub][3][2]);
      mx = fmaxf(mx, s[sub][3][3]);
      mx = group_max(mx);
      // Deferred rescale: m is the maximum as of the last rescale, not the running maximum -- it moves only when some row of the
      // wave exceeds it by more than RESCALE_THR (base-2 units: P <= 2^8 in between; bf16 keeps its relative precision, the f32
      // accumulators have the headroom, and O / l are built from the SAME rounded P, so the normalisation stays consistent).
      // An exact running maximum moves for SOME of a wave's 16 rows in almost every tile -- 1 - (t / (t + 1))^16 on exchangeable
      // scores -- so "rescale when the maximum moved" ran its 45 extra VALU per sub-tile nearly always (PMC: 225 VALU per 36
      // MFMAs).  Tile 0 always takes the path (m starts at 0, not at the first maximum).
      if (t == 0 || __any(mx > RESCALE_THR)) {
        const float d = t == 0 ? mx : fmaxf(mx, 0.f);
        const float alpha = __builtin_amdgcn_exp2f(-d);
#pragma unroll
        for (int i = 0; i < 5; ++i) { o[sub][i][0] *= alpha; o[sub][i][1] *= alpha; o[sub][i][2] *= alpha; o[sub][i][3] *= alpha; }
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { s[sub][kt][0] -= d; s[sub][kt][1] -= d; s[sub][kt][2] -= d; s[sub][kt][3] -= d; }
        negm[sub][0] -= d; negm[sub][1] -= d; negm[sub][2] -= d; negm[sub][3] -= d;
      }
#pragma unroll
      for (int kt = 0; kt < 4; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[sub][kt][r] = __builtin_amdgcn_exp2f(s[sub][kt][r]);
    }
    // ---- [O^T ; l] += [V^T ; 1] P^T
#pragma unroll
    for (int kp = 0; kp < 2; ++kp) {
      bf16x8 pb[QSUB];
#pragma unroll
      for (int sub = 0; sub < QSUB; ++sub) pb[sub] = pack_p(s[sub][2 * kp], s[sub][2 * kp + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 va = read_vfrag_tr(Vs, dt, l15, (2 * kp) * 16 + g * 4, (2 * kp + 1) * 16 + g * 4);
#pragma unroll
        for (int sub = 0; sub < QSUB; ++sub) o[sub][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va, pb[sub], o[sub][dt], 0, 0, 0);
      }
#pragma unroll
      for (int sub = 0; sub < QSUB; ++sub) o[sub][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, pb[sub], o[sub][4], 0, 0, 0);
    }
  };
  const int nfull = L / KT;
  for (int t = 0; t < nfull; ++t) tile(t, std::false_type{});
  for (int t = nfull; t < ntiles; ++t) tile(t, std::true_type{});

  // ---- output: normalise, stage the wave's 16 QSUB rows x 128 B through LDS, store whole rows
  __syncthreads();                                                    // every wave is done with the last K / V tile
  char* stage = smem + wave * (QSUB * 2048);
#pragma unroll
  for (int sub = 0; sub < QSUB; ++sub) {
    const float inv = 1.0f / o[sub][4][0];                            // every row of the all-ones block holds l of query l15
    const int row = sub * 16 + l15;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      uint2 w;
      w.x = pack2bf(o[sub][dt][0] * inv, o[sub][dt][1] * inv);
      w.y = pack2bf(o[sub][dt][2] * inv, o[sub][dt][3] * inv);
      const int chunk = dt * 2 + (g >> 1);
      *(uint2*)(stage + row * 128 + ((chunk ^ kswz(row)) << 4) + (g & 1) * 8) = w;
    }
  }
#pragma unroll
  for (int i = 0; i < QSUB * 2; ++i) {
    const int row = i * 8 + (lane >> 3), p = lane & 7;
    const uint4 v = *(const uint4*)(stage + row * 128 + ((p ^ kswz(row)) << 4));
    const int qrow = q0 + row;
    if (qrow < L) *(uint4*)(O + ((long)b * L + qrow) * Hd + h * HD + p * 8) = v;
  }
}
}  // namespace

// qkv bf16 [B*S, 3*heads*64] token-major (q | k | v thirds; q pre-multiplied by head_dim^-0.5 * log2(e): the softmax runs in
// base 2) -> out bf16 [B*S, heads*64].  Non-causal, no mask (TF:models/glmasr/modeling_glmasr.py:204-217).
extern "C" int ta_attention_enc_fwd(const void* qkv, void* out, int B, int heads, int S, hipStream_t st) {
  if (B <= 0 || heads <= 0 || S <= 0) return TA_ERR_ARG;
  const long rs = 3L * heads * HD;
  constexpr int QSUB = 2;
  const int nq = (S + 64 * QSUB - 1) / (64 * QSUB);
  TA_LAUNCH((attn_enc_fwd_kernel<QSUB>), dim3(grouped_grid(nq, B * heads)), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)out, B,
            heads, S, rs);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
