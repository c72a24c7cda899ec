// Shared between the GEMM translation units (gemm.hip: the 8-wave / 4-wave VGPR-accumulator kernels; gemm_v7.hip: the one-wave-per-SIMD
// AGPR-accumulator kernel): launch arguments, the epilogue of a tile, the tile walk of the persistent kernels.
#pragma once
#include "common.h"
#include "../../include/ta355.h"
#include <type_traits>

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* W;
  void* C;
  const float* bias;   // [N] or null
  const float* res;    // f32 residual, same row map as C, or null
  int M, N, K;
  long lda; int a_rpb; long a_bs;
  long ldc; int c_rpb; long c_bs; long c_off;
  int tiles_m, tiles_n, splits;
  long slab_stride;    // elements between split-K slabs (C is f32 slabs when splits > 1)
  // grouped / routed GEMM (MoE experts) without a host round trip: all three are DEVICE pointers or null
  const int* a_idx;    // A row of logical row r is a_idx[seg_base + r] (gather); null = seg_base + r
  const int* seg;      // {row base, row count}: this launch covers rows [base, base+count) of A (via a_idx) and of C
  const int* krange;   // {first, last+1} K-tile (64-wide): contract only over that slice (per-expert dW over sorted slots)
  // K extension (LoRA): after the K columns of A / W the contraction continues over K2 more columns taken from
  // A2 [M, K2] (row stride lda2) and W2 [N, K2]:  C = A W^T + A2 W2^T  in one accumulator pass
  const bf16_t* A2; const bf16_t* W2; int K2; long lda2;
  int res_bf16;        // the residual is bf16 (same row map as C), not f32
  // act == 2: partial rotary embedding in the epilogue (GLM-ASR q|k projection).  Heads are 64 columns; the first 32
  // columns of every head hold the 16 rotation pairs INTERLEAVED (pair i = columns 2i, 2i+1), so both members of a pair
  // sit in one lane; rope_tab [rope_rows][16][2] = (cos, sin) of pair i at position (logical row % rope_rows)
  const float* rope_tab; int rope_rows; int rope_cols;   // rope on columns [0, rope_cols)
  int group_m;         // tile-order group height (L2 reuse of W panels inside a group of M-tiles)
  int wide;            // bf16 epilogue may use 16-B (8-column) stores: N, ldc, c_off, c_bs all multiples of 8
  int w_blocked;       // W is stored as [N/64][K/64][64][64] blocks (8 KB contiguous per 64 rows x one K tile)
  int a_plain, c_plain; // the row map is the identity (one batch): skips two integer divisions per row in prologue / epilogue
  int dbg;             // experiments only (TA355_GEMM_DEBUG): bit 0 = no epilogue stores, bit 1 = contract over ONE K tile only
  // Grouped launch (MoE experts in ONE launch, ta_gemm_bf16_nt_grouped):
  //   rows form    seg = int[2 * grp_n] {row base, row count}: M tile indices run over the concatenation of the groups' row
  //                tiles; group e multiplies by W + e * grp_w_stride and adds bias + e * N
  //   K-slice form krange = int[2 * grp_n] K-tile ranges, the launch's z index IS the group: (C + z * slab_stride) gets the
  //                product contracted over slice z (per-expert weight gradients over the slot-sorted token axis)
  int grp_n; long grp_w_stride;
};

#define BM 128
#define BN 128
#define BK 64
// per-CU throughput of the 256-row tile variants relative to the 128x128 kernel (two co-resident workgroups),
// measured with scripts/gemm_bench.py; used only by the launch-time variant choice
#ifndef TA355_RATE_256x256
#define TA355_RATE_256x256 0.9      /* simple double buffer: superseded by the ping-pong schedule */
#define TA355_RATE_256x128 0.5      /* measured slower than 128x128 on every shape */
#define TA355_RATE_256x256_PP 1.40  /* per unit tile area vs the 128x128 kernel, fitted on profiles/r01_f_gemm_variants.txt (lm_qkv 56 vs 62 us, sq8192 1330 vs 1050 TF/s) */
#define TA355_RATE_96x128 0.93      /* 3x4 instead of 4x4 MFMAs per fragment set; estimate, to be refitted */
#define TA355_RATE_192x128 1.1      /* v5 (one 192x128 tile per CU), cold operands, profiles/r02_gemm_v5_ab_cold.txt: 1.05-1.15 in one round (lm o / down / dX: 43.5 / 58.5 / 61.7 / 97.2 us vs 49.4 / 68.4 / 74.3 / 120.2 for 96x128), 0.9-1.03 over several rounds; in the step 1.0 / 1.1 / 1.25 are equal for Qwen3-0.6B and 1.1 is 1 ms better than 1.0 for the 1.7B widths (2-round N = 2048 shapes); 0 = never chosen */
#define TA355_RATE_192x256_PP 1.15  /* round 3: v4 on a 192-row tile (variant 12).  Cold, M = 6144 (profiles/r03_b_gemm_lm_cold.txt): d(attn-out) N = 2048 36.3 us vs 42.7 (256x256) / 39.4 (v5); q|k|v N = 4096 794 vs 764 TF/s; gate|up N = 6144 95.9 us vs 80.6 for 256x320 -- the rate must keep 3 rounds of 192x256 ABOVE 2 rounds of 256x320 there (r < 1.278) and below 256x256 for N = 2048 / 4096 (r > 1.05); at 1.30 the step LOST 0.2 ms (gate|up moved) */
#define TA355_RATE_256x320_PP 1.42  /* enc qkv 140 vs 147 us (256x256), fc2 1190 vs 870 TF/s, lm gate|up 73 vs 86 us, lm dact 39 vs 55 us */
#endif
#define TILE_BYTES (BM * BK * 2)   // 16 KiB
#include "gelu_lut.h"

// the same DMA in its scalar-base form: address = 64-bit uniform base (SGPR pair) + 32-bit per-lane byte offset; `lds` is the
// wave-uniform LDS byte address (the hardware adds lane * 16).  Inline assembly: the builtin keeps 64-bit per-lane pointers.
// The compiler does not count these loads: every wait for them is an explicit s_waitcnt vmcnt.
__device__ __forceinline__ void glds16_s(const char* base, unsigned off, unsigned lds) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds), "v"(off), "s"(base) : "memory");
}
__device__ __forceinline__ const char* uniform_ptr(const char* q) {   // pins a wave-uniform pointer into an SGPR pair
  const unsigned long v = (unsigned long)q;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const char*)(((unsigned long)hi << 32) | lo);
}
__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// ---- epilogue of one 16-row fragment strip: lane (l15, g) owns row l15 and, in each of the NT 16-column fragments,
// columns 4g .. 4g+3.  f32 outputs go out as float4 (the four g lanes of a row cover 64 contiguous bytes).  bf16
// outputs would be 8 B per lane = 32-B runs, which the L2 takes at the same request rate as 64-B ones (measured:
// the bf16 epilogue cost 14.5 us per round of 256 tiles vs 7.6 us of HBM time), so adjacent fragments are first
// exchanged between lane rows with v_permlane16_swap: afterwards lane g holds 8 consecutive columns
// (16 * (j + (g & 1)) + 8 * (g >> 1) ...) and one store covers 64 contiguous bytes per row.
// erf-GELU through the chord table of gelu_lut.h staged in LDS (`lut`): 3 VALU + 1 ds_read_b64 + 1 FMA per element instead of the
// 13 VALU + v_rcp + v_exp of gelu_erf_fast.  Round 3: in the encoder's fc1 (M = 16000, N = 5120, K = 1280) the arithmetic form
// cost ~25 k of the ~94 k cycles a CU spends per 256x320 tile -- un-overlapped VALU time in the epilogue.  |error| <= 2.5e-5.
__device__ __forceinline__ float gelu_lut(float x, const float2* lut) {
  float t = fmaf(x, GELU_LUT_SCALE, GELU_LUT_BIAS);
  t = __builtin_amdgcn_fmed3f(t, 0.f, (float)(GELU_LUT_N - 1));
  const float2 e = lut[(int)t];
  return fmaf(e.x, x, e.y);
}
// Round 3: the epilogue's loads are BATCHED.  The ISA of the r02 epilogue had an `s_waitcnt vmcnt(0)` behind every single load --
// bias, residual, rope table, each behind its own per-fragment bounds branch -- i.e. up to 80 serialised memory round trips per
// wave and tile; o_proj spent 35 of its 73 us outside the main loop.  Now (i) every load of a strip is unconditional (clamped
// column, the value is simply not used out of range) and sits in one gather phase in front of the arithmetic, (ii) the bias of
// the lane's NT column groups rides in the same batch (L1 hits; keeping it in registers across the strips spilled), and (iii) the
// bf16 residual of strip i + 1 is requested before strip i is stored (EpiPre, filled by epilogue_tile): the residual stream aliases C, so the compiler may not hoist those loads itself.
// (iv) ELS (the persistent ping-pong kernel): the tile's bias row and its rows of the rope table were DMA'd into the idle LDS
// stage during the LAST K tile of the main loop (EPI_LDS_* below), so the strip reads them with ds_read_b128 and -- with the bf16
// residual taken as the accumulators' start value -- the encoder's epilogues wait for no global load at all.
#define EPI_LDS_BIAS 0        /* float[BN2]: the tile's bias columns (clamped into the matrix) */
#define EPI_LDS_TAB 2048      /* GELU: the chord table (8 KB); rope: BM2 rows x 128 B, 16-B chunk c of local row r at slot c ^ (r & 7) */
// (Round 3, FIRST attempt at row-merged stores, built, measured and removed -- the form that stayed is epilogue_tile_full below.  The two 32-column pairs of a strip exchanged once more between lane l15
// and l15 ^ 8 with DPP moves, so that an instruction stores rows 0-7 resp. 8-15 of the strip with 128 contiguous bytes per row (8
// lines per instruction instead of 16 half lines), also with the 256x320 tile's columns re-mapped to 64 line-aligned columns + a
// 16-column tail per wave.  Bit-identical.  Against the pair form of the SAME build it looked like -10 % per launch (fc1 on 256x256
// tiles 236.6 -> 214.0 us); against the previous build of the library it is equal on the 256-column tiles and 0.3-0.8 ms per step
// slower on 256x320 (profiles/r03_ad_gemm_lib_probe.txt, r03_ad_ab_store_merge_libs.txt): what the first comparison measured was the
// slowdown of carrying both forms -- with per-strip row offsets and guards for every lane -- in one epilogue.)
#ifndef TA355_GELU_ALWAYS_LUT
#define TA355_GELU_ALWAYS_LUT 0     /* gemm_v7.hip: 1 -- its table is always staged, the arithmetic form is not compiled in */
#endif
__device__ __forceinline__ int opaque_sgpr(int v) { asm volatile("" : "+s"(v)); return v; }
template <int NT> struct EpiPre { uint2 r[NT]; };
// oret != nullptr (bf16 outputs): the strip's packed results are handed back instead of stored (epilogue_tile_full stores them)
template <int NT, int ACT, bool OUT_BF16, bool HAS_RES, bool ELS = false>
__device__ __forceinline__ void epilogue_strip(const f32x4* acc, const GemmArgs& p, char* Cb, long roff, int nb, int g, bool wide,
                                               int m, const float* bias, const float2* lut = nullptr, const EpiPre<NT>* pre = nullptr,
                                               bool pre_r = false, const char* els = nullptr, int ecol0 = 0, int erow = 0,
                                               uint2* oret = nullptr, const float4* pref = nullptr) {
  // ACT: 0 none, 1 erf-GELU, 2 partial rotary embedding.  (Rounds 1-4 also carried a folded-LayerNorm form (ACT 3 / 4 / 5) and a fused
  // SwiGLU-backward form (ACT 6) as separate instantiations; both measured slower in the step than the streaming kernels they
  // replaced -- DESIGN.md section 8 -- and left the library in round 5.)
  static_assert(ACT >= 0 && ACT <= 2, "epilogue form");
  constexpr int BASE = ACT;
  // ---- gather phase: every load of the strip, unconditional (column clamped into the matrix)
  float4 bq[NT], rt[NT], rf[NT];
  uint2 rb[NT];
  const long rope_row = BASE == 2 ? (long)(m % p.rope_rows) * 16 : 0;
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = nb + j * 16 + g * 4;
    const int nn = n < p.N ? n : p.N - 4;          // (tiles are column-aligned to 4; out-of-range values are never stored)
    if (bias) bq[j] = ELS ? *(const float4*)(els + EPI_LDS_BIAS + (nn - ecol0) * 4) : *(const float4*)(bias + nn);
    if (BASE == 2) {
      const int pc = nn & 63;
      if (ELS) rt[j] = *(const float4*)(els + EPI_LDS_TAB + erow * 128 + ((((pc < 32 ? pc : 0) >> 2) ^ (erow & 7)) << 4));
      else rt[j] = *(const float4*)(p.rope_tab + (rope_row + (pc < 32 ? (pc >> 1) : 0)) * 2);   // c0 s0 c1 s1
    }
    if (HAS_RES) {
      if (p.res_bf16) rb[j] = pre_r ? pre->r[j] : *(const uint2*)((const bf16_t*)p.res + roff + nn);
      else rf[j] = pref ? pref[j] : *(const float4*)(p.res + roff + nn);
    }
  }
  uint2 o[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = nb + j * 16 + g * 4;
    const bool in = n < p.N;
    f32x4 v = acc[j];
    if (bias) { v[0] += bq[j].x; v[1] += bq[j].y; v[2] += bq[j].z; v[3] += bq[j].w; }
    if (BASE == 1) {
      if (TA355_GELU_ALWAYS_LUT || lut) { v[0] = gelu_lut(v[0], lut); v[1] = gelu_lut(v[1], lut); v[2] = gelu_lut(v[2], lut); v[3] = gelu_lut(v[3], lut); }
      else { v[0] = gelu_erf_fast(v[0]); v[1] = gelu_erf_fast(v[1]); v[2] = gelu_erf_fast(v[2]); v[3] = gelu_erf_fast(v[3]); }
    }
    if (BASE == 2) {
      const int pc = n & 63;                                   // column inside the head; the lane holds pairs pc/2, pc/2+1
      if (pc < 32 && n < p.rope_cols) {
        const float4 t = rt[j];
        const float a0 = v[0], a1 = v[1], a2 = v[2], a3 = v[3];
        v[0] = a0 * t.x - a1 * t.y; v[1] = a1 * t.x + a0 * t.y;
        v[2] = a2 * t.z - a3 * t.w; v[3] = a3 * t.z + a2 * t.w;
      }
    }
    if (HAS_RES) {
      if (p.res_bf16) {
        // (Round 3: reading the residual in the STORE layout instead -- 16 B per lane after the lane exchange, bf16 + bf16 adds --
        // measured 0.19 ms per step SLOWER, profiles/r03_g_ab_res_wide.txt, and made the rounding depend on the tile width; removed.)
        const uint2 r = rb[j];
        v[0] += bf2f((bf16_t)(r.x & 0xffff)); v[1] += bf2f((bf16_t)(r.x >> 16));
        v[2] += bf2f((bf16_t)(r.y & 0xffff)); v[3] += bf2f((bf16_t)(r.y >> 16));
      } else {
        const float4 r = rf[j];
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      }
    }
    if (OUT_BF16) {
      o[j].x = pack2bf(v[0], v[1]);
      o[j].y = pack2bf(v[2], v[3]);
      if (oret) oret[j] = o[j];
      else if (in && (!wide || (j == NT - 1 && (NT & 1)))) *(uint2*)(Cb + (roff + n) * 2) = o[j];
    } else if (in) {
      *(float4*)(Cb + (roff + n) * 4) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  if (OUT_BF16 && wide && !oret) {
#pragma unroll
    for (int j = 0; j + 1 < NT; j += 2) {
      const auto a = __builtin_amdgcn_permlane16_swap(o[j].x, o[j + 1].x, false, false);
      const auto b = __builtin_amdgcn_permlane16_swap(o[j].y, o[j + 1].y, false, false);
      const int col = nb + 16 * (j + (g & 1)) + 8 * (g >> 1);
      if (col < p.N) *(uint4*)(Cb + (roff + col) * 2) = make_uint4(a[0], b[0], a[1], b[1]);
    }
  }
}
// FULL tiles of the persistent kernel (every row and column inside the matrix, identity row map, bf16 out, no residual left for the
// epilogue): ROW-MERGED stores, second attempt.  The first one (see above) spent what it saved on per-strip row offsets and guards;
// here the tile is known to be full, so there are no guards, the two row pointers of a lane are set up once per tile and stepped by
// 16 rows per strip, and what remains per strip is 8 DPP moves: the two 32-column pairs are exchanged between lane l15 and l15 ^ 8,
// one instruction then stores rows 0-7 of the strip and the next rows 8-15, 128 contiguous bytes per row.  Fragments beyond the
// first four (the 16-column tail of a 320-column tile) go out as before.  40.89 -> 40.60 ms per step against the previous library
// (profiles/r03_ag_ab_store_merge_full_tiles.txt).  Re-mapping the 320-column tile's waves to 64 LINE-ALIGNED columns + a tail (three of
// the four waves' 128-B runs straddle two lines here) added nothing on top: 40.02 vs 39.98 ms (r03_ah_..._cmap.txt); not kept.  The same
// full-tile form in the one-tile-per-CU kernel (v5) and in v2 (LoRA K extension): no measurable change (r03_am_...); not kept either.
template <int MI, int NT, int ACT, bool ELS>
__device__ __forceinline__ void epilogue_tile_full(const f32x4 (*acc)[NT], const GemmArgs& p, char* Cb, int ml0, int rbase, int nb, int g,
                                                   const float* bias, const float2* lut, const char* els, int ecol0, int erow0) {
  static_assert(NT >= 4, "two column pairs per strip");
  const int l15 = ml0 & 15;                                   // (tile and wave-group row offsets are multiples of 16)
  const long own0 = p.c_off + (long)(rbase + ml0) * p.ldc;
  const long step = 16L * p.ldc;
  const int colx = nb + (l15 >> 3) * 32 + 16 * (g & 1) + 8 * (g >> 1);
  char* px = Cb + (p.c_off + (long)(rbase + ml0 - (l15 & 8)) * p.ldc + colx) * 2;       // row base + (l15 & 7) of strip 0
  char* py = px + 16L * p.ldc;                                                            // 8 rows further (x 2 bytes)
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    uint2 o[NT];
    epilogue_strip<NT, ACT, true, false, ELS>(acc[i], p, Cb, own0 + i * step, nb, g, true, rbase + ml0 + i * 16, bias, lut, nullptr, false, els,
                                              ecol0, erow0 + i * 16, o);
    uint32_t q[2][4];
#pragma unroll
    for (int pp = 0; pp < 2; ++pp) {
      const auto a = __builtin_amdgcn_permlane16_swap(o[2 * pp].x, o[2 * pp + 1].x, false, false);
      const auto b = __builtin_amdgcn_permlane16_swap(o[2 * pp].y, o[2 * pp + 1].y, false, false);
      q[pp][0] = a[0]; q[pp][1] = b[0]; q[pp][2] = a[1]; q[pp][3] = b[1];
    }
    uint32_t x[4], y[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      x[d] = __builtin_amdgcn_update_dpp(q[0][d], q[1][d], 0x128, 0xf, 0xc, false);   // lanes 8-15 of a row: pair 1 of row l15 - 8
      y[d] = __builtin_amdgcn_update_dpp(q[1][d], q[0][d], 0x128, 0xf, 0x3, false);   // lanes 0-7: pair 0 of row l15 + 8
    }
    *(uint4*)(px + i * step * 2) = make_uint4(x[0], x[1], x[2], x[3]);
    *(uint4*)(py + i * step * 2) = make_uint4(y[0], y[1], y[2], y[3]);
#pragma unroll
    for (int j = 4; j < NT; ++j) *(uint2*)(Cb + (own0 + i * step + nb + j * 16 + g * 4) * 2) = o[j];
  }
}
// The strips of one wave's part of a tile: rows ml0 + 16 i (i < MI); the bf16 residual is requested one strip ahead.
// AHEAD: how the bf16 residual (which aliases C, so the compiler cannot move its loads over the stores) is requested.
//   0  every strip gathers its own loads: ONE round trip per strip (instead of one per fragment and operand in r02)
//   1  strips in PAIRS: the residual of strips i and i + 1 in one batch (4 round trips per 8 strips; +NT registers)
//   2  one strip ahead (the one-wave-per-SIMD kernel: registers to spare; in the 8-wave kernels at 256 VGPRs this form spilled
//      into the main loop)
// PIN (gemm_v7.hip): the first PIN fragments of a strip are AGPR accumulators and stay there until the strip is processed
template <int MI, int NT, int ACT, bool OUT_BF16, bool HAS_RES, int AHEAD = 0, bool ELS = false, int PIN = 0>
__device__ __forceinline__ void epilogue_tile(const f32x4 (*acc)[NT], const GemmArgs& p, char* Cb, int ml0, int Mact, int rbase, int nb,
                                              int g, bool wide, const float* bias, const float2* lut, const char* els = nullptr,
                                              int ecol0 = 0, int erow0 = 0) {
  // (the divisor is made opaque per call: left visible, the compiler hoists its reciprocal out of the persistent tile loop and keeps
  // it live across the main loop -- the one register the 256-VGPR o_proj / fc2 kernel spilled, reloaded at the top of every epilogue)
  const int c_rpb_o = opaque_sgpr(p.c_rpb);
  auto row_off = [&](int m) -> long {
    return p.c_off + (p.c_plain ? (long)m * p.ldc : (long)(m / c_rpb_o) * p.c_bs + (long)(m % c_rpb_o) * p.ldc);
  };
  int ncl[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = nb + j * 16 + g * 4;
    ncl[j] = n < p.N ? n : p.N - 4;
  }
  const bool pre_r = AHEAD > 0 && HAS_RES && p.res_bf16;
  auto fetch = [&](int i, uint2* dst) {                        // residual of strip i (rows past the tile's end: the last valid row)
    int ml = ml0 + i * 16; if (ml >= Mact) ml = Mact - 1;
    const long ro = row_off(rbase + ml);
#pragma unroll
    for (int j = 0; j < NT; ++j) dst[j] = *(const uint2*)((const bf16_t*)p.res + ro + ncl[j]);
  };
  auto strip = [&](int i, const EpiPre<NT>* pre) {
    const int ml = ml0 + i * 16;
    if constexpr (PIN > 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
        if (j < PIN) asm volatile("" : "+a"(const_cast<f32x4&>(acc[i][j])));
    }
    if (ml < Mact) {
      const int m = rbase + ml;
      epilogue_strip<NT, ACT, OUT_BF16, HAS_RES, ELS>(acc[i], p, Cb, row_off(m), nb, g, wide, m, bias, lut, pre, pre_r, els, ecol0, erow0 + i * 16);
    }
  };
  if constexpr (AHEAD == 3 && HAS_RES && !OUT_BF16) {
    // the one-wave-per-SIMD kernel (v5: registers to spare), f32 residual: every strip's residual in ONE batch of loads
    float4 r[MI][NT];
    if (!p.res_bf16) {
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        int ml = ml0 + i * 16; if (ml >= Mact) ml = Mact - 1;
        const long ro = row_off(rbase + ml);
#pragma unroll
        for (int j = 0; j < NT; ++j) r[i][j] = *(const float4*)(p.res + ro + ncl[j]);
      }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int ml = ml0 + i * 16;
      if (ml < Mact) {
        const int m = rbase + ml;
        epilogue_strip<NT, ACT, OUT_BF16, HAS_RES, ELS>(acc[i], p, Cb, row_off(m), nb, g, wide, m, bias, lut, nullptr, false, els, ecol0,
                                                        erow0 + i * 16, nullptr, r[i]);
      }
    }
  } else if constexpr (AHEAD == 1 && HAS_RES && !OUT_BF16) {
    // Round 6, the fp32-stream mode (f32 residual, f32 out, in place: x += A W^T + b): the residual of TWO strips in one batch of
    // loads (the residual aliases C, so the compiler cannot hoist a strip's loads over the
    // previous strip's stores by itself)
    static_assert(MI % 2 == 0, "pairs of strips");
    auto fetch_f = [&](int i, float4* dst) {
      int ml = ml0 + i * 16; if (ml >= Mact) ml = Mact - 1;
      const long ro = row_off(rbase + ml);
#pragma unroll
      for (int j = 0; j < NT; ++j) dst[j] = *(const float4*)(p.res + ro + ncl[j]);
    };
    auto strip_f = [&](int i, const float4* r) {
      const int ml = ml0 + i * 16;
      if (ml < Mact) {
        const int m = rbase + ml;
        epilogue_strip<NT, ACT, OUT_BF16, HAS_RES, ELS>(acc[i], p, Cb, row_off(m), nb, g, wide, m, bias, lut, nullptr, false, els, ecol0,
                                                        erow0 + i * 16, nullptr, r);
      }
    };
    // the first two strips alone: a pair's second batch of loads lives in the registers of accumulator strips that are already
    // stored (with all 8 strips live a batch of 2 x NT float4 spilled: 96 bytes of scratch per lane) -- 5 round trips instead of 8
    static_assert(MI >= 4, "two single strips, then pairs");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      float4 r0[NT];
      if (!p.res_bf16) fetch_f(i, r0);
      strip_f(i, r0);
    }
#pragma unroll
    for (int i = 2; i < MI; i += 2) {
      float4 r0[NT], r1[NT];
      if (!p.res_bf16) { fetch_f(i, r0); fetch_f(i + 1, r1); }
      strip_f(i, r0);
      strip_f(i + 1, r1);
    }
  } else if constexpr (AHEAD == 1) {
    static_assert(MI % 2 == 0, "pairs of strips");
#pragma unroll
    for (int i = 0; i < MI; i += 2) {
      EpiPre<NT> p0, p1;
      if (pre_r) { fetch(i, p0.r); fetch(i + 1, p1.r); }
      strip(i, &p0);
      strip(i + 1, &p1);
    }
  } else {
    EpiPre<NT> pre;
    uint2 nxt[NT];
    if (pre_r) fetch(0, pre.r);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      if (pre_r && i + 1 < MI) fetch(i + 1, nxt);
      strip(i, &pre);
      if (pre_r && i + 1 < MI) {
#pragma unroll
        for (int j = 0; j < NT; ++j) pre.r[j] = nxt[j];
      }
    }
  }
}
// A plain bf16 residual (ACT 0: x += A W^T + b, the encoder's o_proj / fc2 and the LM's o / down products) is the START VALUE of
// the accumulators in EVERY tile variant, not an addend of the epilogue: the sum is then fl(..fl(fl(r + a0 w0) + a1 w1)..) + b
// whatever tile the launch-time model picks (the B = 32 step and the same clips at B = 4 run different variants and are compared
// in the tests), and the kernels' epilogues have no residual load left to wait for.  Loads in batches of <= 4 strips.
// TA355_GEMM_RES_INIT=0 (p.dbg bit 20): the r02 form, residual added in the epilogue.
// (Round 6 tried the same for an f32 residual with an f32 output -- the fp32-stream mode: float4 loads straight into the accumulator
// registers behind the first K tile's DMA.  Same-box A/B, profiles/r06_b_ab_f32.txt: the encoder's o_proj / fc2 launches 133.8 us
// against 130.2, the LM's one-wave-per-SIMD tiles 65.2 against 43.2: the start values must have landed before the first MFMA, so
// the read is as exposed at the top of the tile as it was in the epilogue, and 96-160 registers of loads per lane queue in front
// of the first K tile.  Removed.)
template <int ACT, bool OUT_BF16, bool HAS_RES>
__device__ __forceinline__ bool residual_is_start(const GemmArgs& p) {
  return HAS_RES && ACT == 0 && OUT_BF16 && p.res_bf16 && p.splits == 1 && !(p.dbg & (1 << 20));
}
// HBMAX: strips per batch of loads (0: half the tile, at most 4); PIN (gemm_v7.hip): the first PIN fragments of a strip are AGPR
// accumulators -- each start value moves there as soon as it is converted
template <int MI, int NT, int HBMAX = 0, int PIN = 0>
__device__ __forceinline__ void residual_start(f32x4 (*acc)[NT], const GemmArgs& p, int ml0, int Mact, int rbase, int nb, int g) {
  const int c_rpb_o = opaque_sgpr(p.c_rpb);          // (see epilogue_tile)
  int ncl[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { const int n = nb + j * 16 + g * 4; ncl[j] = n < p.N ? n : p.N - 4; }
  constexpr int HB = HBMAX > 0 ? HBMAX : (MI > 4 ? (MI + 1) / 2 : MI);
#pragma unroll
  for (int i0 = 0; i0 < MI; i0 += HB) {
    uint2 t[HB][NT];
#pragma unroll
    for (int i = 0; i < HB; ++i)
      if (i0 + i < MI) {
        const int m = rbase + min(ml0 + (i0 + i) * 16, Mact - 1);
        const long ro = p.c_off + (p.c_plain ? (long)m * p.ldc : (long)(m / c_rpb_o) * p.c_bs + (long)(m % c_rpb_o) * p.ldc);
#pragma unroll
        for (int j = 0; j < NT; ++j) t[i][j] = *(const uint2*)((const bf16_t*)p.res + ro + ncl[j]);
      }
#pragma unroll
    for (int i = 0; i < HB; ++i)
      if (i0 + i < MI) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const uint2 r = t[i][j];
          acc[i0 + i][j] = (f32x4){bf2f((bf16_t)(r.x & 0xffff)), bf2f((bf16_t)(r.x >> 16)), bf2f((bf16_t)(r.y & 0xffff)), bf2f((bf16_t)(r.y >> 16))};
          if constexpr (PIN > 0) { if (j < PIN) asm volatile("" : "+a"(acc[i0 + i][j])); }
        }
      }
  }
}
template <int MI, int NT>
__device__ __forceinline__ void zero_acc(f32x4 (*acc)[NT]) {
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
}
// Every tile variant evaluates GELU through the same chord table (a GEMM's result must not depend on the tile the launch-time
// model picks: the B = 32 step and the same clips at B = 4 run different variants and are compared in the tests).  `smem` must be
// free: every wave past its last LDS read of the main loop.
template <int ACT, int NTHREADS>
__device__ __forceinline__ const float2* stage_gelu_lut(char* smem, const GemmArgs& p, int tid, bool barrier_first) {
  if constexpr (ACT == 1) {
    if (!(p.dbg & 8)) {
      if (barrier_first) __syncthreads();
      for (int i = tid; i < GELU_LUT_N * 8 / 16; i += NTHREADS) ((uint4*)smem)[i] = ((const uint4*)kGeluLut)[i];
      __syncthreads();
      return (const float2*)smem;
    }
  }
  return nullptr;
}
__device__ __forceinline__ bool epilogue_wide_ok(const GemmArgs& p) { return p.wide != 0; }


// Group of a tile in a grouped launch (see GemmArgs.grp_n).  rows form: walks the <= 8 segments; returns false for the
// surplus tiles of the (upper-bound) grid.
template <int BMT_>
__device__ __forceinline__ bool resolve_group(const GemmArgs& p, int& pm, int z, const int*& seg, const bf16_t*& W,
                                              const float*& bias, const int*& krange) {
  seg = p.seg; W = p.W; bias = p.bias; krange = p.krange;
  if (p.grp_n <= 0) return true;
  if (p.seg) {
    int e = 0, rem = pm;
    for (; e < p.grp_n; ++e) {
      const int te = (p.seg[2 * e + 1] + BMT_ - 1) / BMT_;
      if (rem < te) break;
      rem -= te;
    }
    if (e == p.grp_n) return false;
    pm = rem; seg = p.seg + 2 * e; W = p.W + (long)e * p.grp_w_stride;
    if (bias) bias += (long)e * p.N;
  } else if (p.krange) {
    krange = p.krange + 2 * z;
  }
  return true;
}

struct TileCtx { int ok, m0, n0, kb, ke, Mact, rbase, z; const bf16_t* Wp; const float* biasp; };
template <int BM2, int BN2, bool KEXT>
__device__ __forceinline__ TileCtx tile_ctx(const GemmArgs& p, int h, int total) {
  TileCtx c; c.ok = 0;
  int bid = h;
  {
    const int q = total >> 3, r = total & 7, xcd = bid & 7, within = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + within;
  }
  const int tiles = p.tiles_m * p.tiles_n;
  const int z = bid / tiles;
  const int t = bid - z * tiles;
  const int GROUP_M = p.group_m > 0 ? p.group_m : 4;
  const int width = GROUP_M * p.tiles_n;
  const int group = t / width;
  const int first_m = group * GROUP_M;
  const int gsize = min(p.tiles_m - first_m, GROUP_M);
  int pm = first_m + (t % width) % gsize;
  const int pn = (t % width) / gsize;
  const int* segp; const int* krp;
  if (!resolve_group<BM2>(p, pm, z, segp, c.Wp, c.biasp, krp)) return c;
  c.m0 = pm * BM2; c.n0 = pn * BN2; c.z = z;
  const int nkt = p.K / BK;
  c.kb = 0; c.ke = nkt;
  if (p.splits > 1) { c.kb = (nkt * z) / p.splits; c.ke = (nkt * (z + 1)) / p.splits; }   // 32-bit: the 64-bit quotients of v2 cost ~2 k cycles per tile
  if (krp) { c.kb = krp[0]; c.ke = krp[1]; }
  if (KEXT) c.ke = nkt + p.K2 / BK;
  if (p.dbg & 2) c.ke = min(c.ke, c.kb + 1);
  c.Mact = p.M; c.rbase = 0;
  if (segp) { c.rbase = segp[0]; c.Mact = segp[1]; if (c.m0 >= c.Mact) return c; }
  // wave-uniform by construction; values that came through a vector load (segment / K-range tables) are marked as such, so the K
  // loop's control and the stage parity derived from it stay in SGPRs
  c.kb = __builtin_amdgcn_readfirstlane(c.kb); c.ke = __builtin_amdgcn_readfirstlane(c.ke);
  c.rbase = __builtin_amdgcn_readfirstlane(c.rbase); c.Mact = __builtin_amdgcn_readfirstlane(c.Mact);
  c.ok = 1;
  return c;
}
