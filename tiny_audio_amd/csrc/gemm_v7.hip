// ta355 GEMM, round 4: ONE 4-wave workgroup per CU, one wave per SIMD with the whole 512-entry register file -- the accumulators live
// in AGPRs (256 of them; the 256x320 tile keeps its last two fragment columns in VGPRs), the VGPRs hold one set of operand fragments.
//
// Why a fourth kernel family.  The 8-wave ping-pong tiles (gemm.hip v2 / v4) run two waves per SIMD at 256 registers each: 160
// accumulator registers for a 128 x 80 wave tile, 13 ds_read_b128 per 40 MFMAs, four barriers per K tile; the one-192x128-tile-per-CU
// kernel (v5) has the register file to itself but stages 40 KB per K tile for 768 MFMA cycles -- 83 % of the 64 B/clk the CU's
// vector-memory path moves, issued as one burst by all four waves (measured there: ~48 idle cycles per DMA instruction).
// Here a wave owns (16 MI) x (16 NJ) outputs of a (32 MI) x (32 NJ) tile:
//     8 x 8   256 x 256   64 MFMAs per k-step of 32, 16 fragment reads,  8 DMA pieces per wave: 50 % of the vector-memory path
//     8 x 10  256 x 320   80                         18                  9                       45 %
//     6 x 8   192 x 256   48                         14                  7                       58 %
// * Operands travel HBM -> LDS by DMA (global_load_lds_dwordx4) in k-steps of 32 columns: a ring of FOUR stages of
//   (BM + BN) x 64 B, three k-steps ahead of the MFMAs, counted waits (s_waitcnt vmcnt(pieces per wave)): every DMA has two to
//   three k-steps (~2.5-3.5 k cycles) to land.  The ring does not stop at a tile boundary: the last k-steps of a tile already
//   request the first k-steps of the workgroup's next tile, and its first fragments are in registers before the epilogue starts.
// * ONE barrier per k-step (it publishes stage G+1 and retires stage G-1).  The fragments of k-step G+1 are read during k-step G,
//   each into the register its predecessor has just left: W fragment j after the last MFMA of column j, A fragment i after MFMA
//   (i, last column) -- one set of fragment registers, no double buffer.
// * The matrix pipe sees `v_mfma_f32_16x16x32_bf16 a[..], v[..], v[..], a[..]` from inline assembly ("+a": the compiler's own AGPR
//   form copies every accumulator through a scratch AGPR around each MFMA, and the library's other kernels are built with
//   -amdgpu-mfma-vgpr-form), one ds_read / DMA issue at most between two of them, the order pinned by sched_barrier.
// * A wave's DMA pieces of a k-step are SPREAD over the k-step's MFMA slots (one piece every SL / ND MFMAs) instead of issued as a
//   burst: the CU's vector-memory path takes one 1-KB piece per ~16 cycles, and a wave waits at the issue until its piece is taken.
//   (Per-wave slot offsets -- four instantiations of the k loop, one per wave -- were built first: with 256 pinned AGPRs the
//   register allocator split the accumulators' live ranges across the four copies and spilled them; one copy of the loop it is.)
// * Same accumulation order per output element as every other variant (k ascending, fp32), same epilogue code (gemm_common.h):
//   results are bit-identical to the other tiles.
// Not served here (the launch falls back): gathered A rows, grouped launches, the K extension, blocked W, K ranges, fewer than
// four k-steps per tile, the folded-LayerNorm / fused-SwiGLU epilogues.
#define TA355_GELU_ALWAYS_LUT 1
#include "gemm_common.h"
#include <type_traits>

#define V7_NS 4                     /* ring stages of the one-workgroup-per-CU form (template parameter NS; the two-per-CU form has 3) */
#define V7_LUT_BYTES (GELU_LUT_N * 8)
#define V7_BIAS_BYTES 768        /* per wave: the bias of its <= 160 columns (+ 32 floats the clamped reads of a ragged tile may touch) */

template <int N> __device__ __forceinline__ void v7_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// acc += W_frag x A_frag on the matrix pipe; AGPR (IN_A) or VGPR accumulator.  The operands are swapped as everywhere in this library
// (first = W rows, second = A rows), so lane (l15, g) owns row l15 and columns 4g..4g+3 of the 16x16 block.
template <bool IN_A> __device__ __forceinline__ void v7_mfma(f32x4& c, const bf16x8& w, const bf16x8& a) {
  if constexpr (IN_A) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(w), "v"(a));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(a));
}

// FULL tiles with bf16 output on the identity row map: row-merged stores (see epilogue_tile_full in gemm_common.h) for every group of
// four fragments of a strip -- 8 rows x 128 contiguous bytes per instruction -- and 64-B pair stores for a remainder of two.
// The accumulators of strip i stay in their AGPRs until HERE: without this the compiler copies all 256 of them into VGPRs right
// behind the main loop (the asm's outputs feed VALU code) and spills.
template <int NT, int PIN> __device__ __forceinline__ void v7_pin_strip(f32x4* a) {
#pragma unroll
  for (int j = 0; j < NT; ++j)
    if (j < PIN) asm volatile("" : "+a"(a[j]));
}
template <int MI, int NT, int ACT, int PIN, bool ELS>
__device__ __forceinline__ void v7_epilogue_full(f32x4 (*acc)[NT], const GemmArgs& p, char* Cb, int ml0, int rbase, int nb, int g,
                                                 const float* bias, const float2* lut, const char* els, int ecol0) {
  static_assert(NT % 2 == 0, "pairs of fragments");
  const int l15 = ml0 & 15;
  const long own0 = p.c_off + (long)(rbase + ml0) * p.ldc;
  const long step = 16L * p.ldc;
  const int colx = nb + (l15 >> 3) * 32 + 16 * (g & 1) + 8 * (g >> 1);
  char* px = Cb + (p.c_off + (long)(rbase + ml0 - (l15 & 8)) * p.ldc + colx) * 2;
  char* py = px + 16L * p.ldc;
#pragma unroll
  for (int i = 0; i < MI; ++i) {
    uint2 o[NT];
    v7_pin_strip<NT, PIN>(acc[i]);
    epilogue_strip<NT, ACT, true, false, ELS>(acc[i], p, Cb, own0 + i * step, nb, g, true, rbase + ml0 + i * 16, bias, lut, nullptr, false,
                                              els, ecol0, 0, o);
#pragma unroll
    for (int q4 = 0; q4 + 3 < NT; q4 += 4) {
      uint32_t q[2][4];
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const auto a = __builtin_amdgcn_permlane16_swap(o[q4 + 2 * pp].x, o[q4 + 2 * pp + 1].x, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(o[q4 + 2 * pp].y, o[q4 + 2 * pp + 1].y, false, false);
        q[pp][0] = a[0]; q[pp][1] = b[0]; q[pp][2] = a[1]; q[pp][3] = b[1];
      }
      uint32_t x[4], y[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        x[d] = __builtin_amdgcn_update_dpp(q[0][d], q[1][d], 0x128, 0xf, 0xc, false);
        y[d] = __builtin_amdgcn_update_dpp(q[1][d], q[0][d], 0x128, 0xf, 0x3, false);
      }
      *(uint4*)(px + i * step * 2 + q4 * 32) = make_uint4(x[0], x[1], x[2], x[3]);
      *(uint4*)(py + i * step * 2 + q4 * 32) = make_uint4(y[0], y[1], y[2], y[3]);
    }
    if constexpr (NT % 4 == 2) {
      constexpr int j = NT - 2;
      const auto a = __builtin_amdgcn_permlane16_swap(o[j].x, o[j + 1].x, false, false);
      const auto b = __builtin_amdgcn_permlane16_swap(o[j].y, o[j + 1].y, false, false);
      const int col = nb + 16 * (j + (g & 1)) + 8 * (g >> 1);
      *(uint4*)(Cb + (own0 + i * step + col) * 2) = make_uint4(a[0], b[0], a[1], b[1]);
    }
  }
}

// Tile `hh` of the launch: the XCD-aware grouped order of tile_ctx (gemm_common.h) without the grouped-launch / segment / K-range forms
// this kernel does not serve (their table walk is a loop of global loads -- here it would sit inside the k loop).
template <int BM2, int BN2>
__device__ __forceinline__ TileCtx v7_tile_ctx(const GemmArgs& p, int hh, int total) {
  TileCtx c; c.ok = 1;
  int bid = hh;
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
  const int pm = first_m + (t % width) % gsize;
  const int pn = (t % width) / gsize;
  c.Wp = p.W; c.biasp = p.bias;
  c.m0 = pm * BM2; c.n0 = pn * BN2; c.z = z;
  const int nkt = p.K / BK;
  c.kb = 0; c.ke = nkt;
  if (p.splits > 1) { c.kb = (nkt * z) / p.splits; c.ke = (nkt * (z + 1)) / p.splits; }
  c.Mact = p.M; c.rbase = 0;
  return c;
}

// MI x NJ: 16 x 16 fragments per wave (waves 2 x 2).
//
// Round 5, WPC = 2: TWO such workgroups per CU on half-size tiles (MI x NJ = 4 x 8: 128 x 256, or 8 x 4: 256 x 128) -- 128 accumulator
// AGPRs + <= 128 VGPRs per lane, a THREE-stage ring (72 KB) per workgroup.  The two workgroups of a CU run out of phase on their own
// tile sequences, so one's epilogue (stores, the next tile's start values) sits beside the other's k loop on the same SIMDs instead of
// leaving the matrix pipe idle: what the step's short contractions (K = 1024 ... 2048: 32 ... 64 k-steps per tile) lack with one
// workgroup per CU.  NS = 3 keeps the three-k-step DMA lead: group G + 3 goes to the stage k-step G is multiplying FROM REGISTERS --
// its fragment reads were issued during k-step G - 1 and are waited for (lgkmcnt(0)) in front of the barrier that opens k-step G.
template <int MI, int NJ, int ACT, bool OUT_BF16, bool HAS_RES, int NS = V7_NS, int WPC = 1>
__global__ __launch_bounds__(256, WPC) void gemm_nt_kernel_v7(GemmArgs p) {
  constexpr int BM7 = 32 * MI, BN7 = 32 * NJ;
  constexpr int PA = BM7 / 16, PW = BN7 / 16;            // 1-KB pieces (16 rows x 64 B) of A / of W per k-step
  constexpr int NDA = PA / 4, NDW = PW / 4, ND = NDA + NDW;   // pieces per wave: wave w stages pieces w, w + 4, ...
  static_assert(PA % 4 == 0 && PW % 4 == 0, "pieces divide over the four waves");
  constexpr int STAGE = (PA + PW) * 1024;
  constexpr int SL = MI * NJ;                            // MFMA slots per k-step
  constexpr int STRIDE = SL / ND;                        // a wave's DMA pieces are STRIDE slots apart
  constexpr int NJ0 = 256 / (4 * MI) < NJ ? 256 / (4 * MI) : NJ;   // fragment columns whose accumulators are AGPRs (256 of them) ...
  constexpr int NJ1 = NJ - NJ0;                                    // ... and the rest (256x320: two columns), in VGPRs
  constexpr int NQ = NJ0 / 4;                                      // the AGPR columns in groups of four (64 output columns = one 128-B line of bf16)
  static_assert(NJ0 % 4 == 0, "quads");
  constexpr bool GELU = ACT == 1;
  constexpr int LUT_BYTES = GELU ? V7_LUT_BYTES : 0;
  constexpr bool ELS = ACT != 2;                                   // bias through LDS (rope keeps its table in global memory: 32 KB per tile)
  static_assert(NS == 3 || NS == 4, "ring stages");
  static_assert(NS * STAGE + LUT_BYTES + 4 * V7_BIAS_BYTES <= 160 * 1024 / WPC, "LDS");
  __shared__ __attribute__((aligned(16))) char smem[NS * STAGE + LUT_BYTES + 4 * V7_BIAS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, l15 = lane & 15;
  const int total = p.tiles_m * p.tiles_n * p.splits;
  const int nkt = p.K / BK;

  // ---- LDS image of a stage: pieces [A 0..PA) [W 0..PW) of 1 KB; inside a piece row r (64 B) holds its 16-B chunk c at slot
  //      c ^ m(r >> 2), m = (0, 2, 3, 1): the four lane groups of a ds_read_b128 (16 rows, one chunk each) hit all 64 banks once
  //      (the layout of gemm_nt_kernel_v3).  The DMA writes lane-linear, so the XOR is applied to the SOURCE chunk.
  const int rd = l15 * 64 + ((g ^ ((0x78 >> (2 * ((l15 >> 2) & 3))) & 3)) << 4);
  const int a_rd = wm * (MI * 1024) + rd;
  const int b_rd = PA * 1024 + wn * (NJ * 1024) + rd;
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem) + wave * 1024;

  if (GELU) {                                                    // the chord table of the erf-GELU epilogue, once per workgroup
    for (int i = tid; i < V7_LUT_BYTES / 16; i += 256) ((uint4*)(smem + NS * STAGE))[i] = ((const uint4*)kGeluLut)[i];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (published by the first barrier below)
  }
  const float2* lut = GELU ? (const float2*)(smem + NS * STAGE) : nullptr;
  char* bias_lds = smem + NS * STAGE + LUT_BYTES + wave * V7_BIAS_BYTES;   // wave-private: no barrier around it

  // ---- DMA side: THREE k-steps ahead of the MFMAs.  Inside a tile the two base pointers step by 64 B per k-step; when the MFMAs
  //      stand three k-steps before a tile's end, the DMA side moves to the workgroup's next tile (dma_setup) -- host: every tile has
  //      at least four k-steps, so it is never more than one tile ahead.  Past the last tile the pointers stay where they are: the
  //      surplus groups re-load the last k-step into retired stages, so that every k-step issues exactly ND pieces per wave and every
  //      wait is the same count.
  const char* a_base; const char* w_base;                        // uniform
  unsigned a_off[NDA], w_off[NDW];                               // per-lane byte offsets of this wave's pieces
  auto fresh_lane = [&]() -> int {                               // (opaque: keeps per-tile index arithmetic out of the k loop's live ranges)
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  auto dma_setup = [&](const TileCtx& c) {
    const int ln = fresh_lane();
    const int drow = ln >> 2;
    const int dchk = (ln & 3) ^ ((0x78 >> (2 * ((ln >> 4) & 3))) & 3);
    const int ld2 = (int)(p.lda * 2), k2 = p.K * 2;
    if (p.a_plain) {
#pragma unroll
      for (int d = 0; d < NDA; ++d) {
        const int r = min(16 * (wave + 4 * d) + drow, c.Mact - 1 - c.m0);
        a_off[d] = (unsigned)r * (unsigned)ld2 + dchk * 16;
      }
      a_base = uniform_ptr((const char*)(p.A + (long)(c.rbase + c.m0) * p.lda) + (long)c.kb * (BK * 2));
    } else {                                                     // affine row map (conv, frame stack): addressed from the start of A (host: < 4 GB)
#pragma unroll
      for (int d = 0; d < NDA; ++d) {
        const int gm = c.rbase + min(c.m0 + 16 * (wave + 4 * d) + drow, c.Mact - 1);
        a_off[d] = (unsigned)(((long)(gm / p.a_rpb) * p.a_bs + (long)(gm % p.a_rpb) * p.lda) * 2 + dchk * 16);
      }
      a_base = uniform_ptr((const char*)p.A + (long)c.kb * (BK * 2));
    }
#pragma unroll
    for (int d = 0; d < NDW; ++d) {
      const int r = min(16 * (wave + 4 * d) + drow, p.N - 1 - c.n0);
      w_off[d] = (unsigned)r * (unsigned)k2 + dchk * 16;
    }
    w_base = uniform_ptr((const char*)(c.Wp + (long)c.n0 * p.K) + (long)c.kb * (BK * 2));
  };
  // piece d of the k-step the DMA side stands at, into ring stage `st`
  auto dma_piece = [&](int d, int st) {
    const unsigned base = lds_w + st * STAGE;
    if (d < NDA) glds16_s(a_base, a_off[d < NDA ? d : 0], base + d * 4096);
    else glds16_s(w_base, w_off[d >= NDA ? d - NDA : 0], base + PA * 1024 + (d - NDA) * 4096);
  };

  int h = blockIdx.x;
  TileCtx cur = v7_tile_ctx<BM7, BN7>(p, h, total);          // (host: no groups / segments, so every tile index is a tile)
  dma_setup(cur);
  for (int q = 0; q < 3; ++q) {                                   // groups 0..2 into stages 0..2
#pragma unroll
    for (int d = 0; d < ND; ++d) dma_piece(d, q);
    a_base += 64; w_base += 64;
  }

  // ---- the first k-step's fragments
  bf16x8 af[MI], bfr[NJ];
  v7_wait_vm<2 * ND>();                                           // group 0 landed
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int j = 0; j < NJ; ++j) bfr[j] = *(const bf16x8*)(smem + b_rd + j * 1024);
#pragma unroll
  for (int i = 0; i < MI; ++i) af[i] = *(const bf16x8*)(smem + a_rd + i * 1024);
  // SYNC of k-step G: group G + 1 has landed in every wave's share (group G + 2 may stay in flight), and every wave is past its reads
  // of stage G - 1, which group G + 3 -- requested during k-step G -- overwrites
  auto sync = [&]() {
    v7_wait_vm<ND>();
    if constexpr (NS == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of the stage group G + 3 will overwrite
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  sync();
  int rs = 0;                                                     // ring stage of the k-step being multiplied; group G + 3 goes to stage rs + 3

  const bool res_init = residual_is_start<ACT, OUT_BF16, HAS_RES>(p);
  for (;;) {
    f32x4 acc0[NQ][MI][4], acc1[MI][NJ1 > 0 ? NJ1 : 1];
    {
      const int ln = fresh_lane();
      const int row0 = cur.m0 + wm * (BM7 / 2) + (ln & 15), col0 = cur.n0 + wn * (BN7 / 2);
      if (res_init) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) residual_start<MI, 4, 4, 4>(acc0[q], p, row0, cur.Mact, cur.rbase, col0 + 64 * q, ln >> 4);
        if constexpr (NJ1 > 0) residual_start<MI, NJ1>(acc1, p, row0, cur.Mact, cur.rbase, col0 + 16 * NJ0, ln >> 4);
      } else {
#pragma unroll
        for (int q = 0; q < NQ; ++q) zero_acc<MI, 4>(acc0[q]);
        if constexpr (NJ1 > 0) zero_acc<MI, NJ1>(acc1);
      }
    }
    const int nk = __builtin_amdgcn_readfirstlane(2 * (cur.ke - cur.kb));
    const bool more = h + (int)gridDim.x < total;
    TileCtx nx = cur;
    asm volatile("s_nop 4" ::: "memory");                         // accumulator start values (VALU writes) -> first MFMA

    // (the SYNC of a k-step sits at the END of its predecessor -- for a tile's first k-step that is the previous tile's last one, in
    // front of the epilogue: every wave then enters a tile with its first fragments in registers and stage 1 published)
    for (int ks = 0; ks < nk; ++ks) {
      if (ks == nk - 3 && more) {                                 // from here on the DMA side requests the next tile's first k-steps
        nx = v7_tile_ctx<BM7, BN7>(p, h + gridDim.x, total);
        dma_setup(nx);
      }
      // one k-step: SL MFMAs, between them the fragment reads of k-step + 1 and the DMA pieces of k-step + 3
      const char* Sn = smem + (rs + 1 == NS ? 0 : rs + 1) * STAGE;
      const int is = NS == 3 ? rs : ((rs + 3) & 3);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int s = j * MI + i;
          if (j < NJ0) v7_mfma<true>(acc0[j < NJ0 ? j >> 2 : 0][i][j & 3], bfr[j], af[i]);
          else v7_mfma<false>(acc1[i][j >= NJ0 ? j - NJ0 : 0], bfr[j], af[i]);
          if (j == NJ - 1) af[i] = *(const bf16x8*)(Sn + a_rd + i * 1024);
          if (i == MI - 1) bfr[j] = *(const bf16x8*)(Sn + b_rd + j * 1024);
          if (s % STRIDE == STRIDE / 2 && s / STRIDE < ND) dma_piece(s / STRIDE, is);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      {                                                           // (past the very last k-step: stay, see above)
        const long st = (more || ks + 4 < nk) ? 64 : 0;
        a_base = uniform_ptr(a_base + st); w_base = uniform_ptr(w_base + st);
      }
      rs = rs + 1 == NS ? 0 : rs + 1;
      if (more || ks + 1 < nk) sync();
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");            // last MFMA results -> VALU reads of the accumulators

    // ---- epilogue of `cur`; the ring keeps filling for the next tile
    if (!(p.dbg & 1)) {
      const int ln = fresh_lane();
      const int e_g = ln >> 4;
      const int row0 = cur.m0 + wm * (BM7 / 2) + (ln & 15), col0 = cur.n0 + wn * (BN7 / 2);
      char* Cb = (char*)p.C;
      if (p.splits > 1) Cb += (long)cur.z * p.slab_stride * 4;
      const bool wide = epilogue_wide_ok(p);
      // One wave per SIMD: nobody hides a load's latency here.  The wave's bias columns come from global memory ONCE per tile (one
      // round trip) into its private LDS row; the strips read them with ds_read_b128 (the per-strip bias loads of the shared epilogue
      // cost 24 L2 round trips per tile: o_proj 12.6 us of epilogue against 7 for the ping-pong kernel)
      if (ELS && cur.biasp) {
        if (ln < BN7 / 2 / 4 + 8) {
          const int n = col0 + 4 * ln;
          *(float4*)(bias_lds + ln * 16) = *(const float4*)(cur.biasp + (n < p.N ? n : p.N - 4));
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      const bool full = OUT_BF16 && wide && p.c_plain && (!HAS_RES || res_init) && cur.m0 + BM7 <= cur.Mact && cur.n0 + BN7 <= p.N &&
                        !(p.dbg & 2048);
      auto part = [&](auto* acc, auto nt_tag, auto pin_tag, int nb) {   // one column group of the wave's tile
        constexpr int NT = decltype(nt_tag)::value, PIN = decltype(pin_tag)::value;
        if constexpr (OUT_BF16) {
          if (full) v7_epilogue_full<MI, NT, ACT, PIN, ELS>(acc, p, Cb, row0, cur.rbase, nb, e_g, cur.biasp, lut, bias_lds, col0);
        }
        if (full) { }
        else if (res_init) epilogue_tile<MI, NT, ACT, OUT_BF16, false, 0, ELS, PIN>(acc, p, Cb, row0, cur.Mact, cur.rbase, nb, e_g, wide, cur.biasp, lut, bias_lds, col0, 0);
        else epilogue_tile<MI, NT, ACT, OUT_BF16, HAS_RES, 0, ELS, PIN>(acc, p, Cb, row0, cur.Mact, cur.rbase, nb, e_g, wide, cur.biasp, lut, bias_lds, col0, 0);
      };
      if constexpr (NJ1 > 0) part(acc1, std::integral_constant<int, NJ1>{}, std::integral_constant<int, 0>{}, col0 + 16 * NJ0);   // (first: frees its VGPRs)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        part(acc0[q], std::integral_constant<int, 4>{}, std::integral_constant<int, 4>{}, col0 + 64 * q);
      }
    }
    if (!more) break;
    h += gridDim.x;
    cur = nx;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // no DMA may be in flight into LDS when the workgroup ends
}

// ----------------------------------------------------------------------------- host side
static bool v7_geometry(int variant, int& bm, int& bn) {
  if (variant == 13) { bm = 256; bn = 256; return true; }
  if (variant == 14) { bm = 256; bn = 320; return true; }
  if (variant == 15) { bm = 192; bn = 256; return true; }
  if (variant == 16) { bm = 128; bn = 256; return true; }         // two workgroups per CU (round 5)
  if (variant == 17) { bm = 256; bn = 128; return true; }
  return false;
}
bool gemm_v7_two_per_cu(int variant) { return variant == 16 || variant == 17; }

bool gemm_v7_serves(int variant, int act, bool out_bf16, bool has_res, const GemmArgs& a) {
  int bm, bn;
  if (!v7_geometry(variant, bm, bn)) return false;
  if (a.a_idx || a.seg || a.krange || a.grp_n > 0 || a.A2 || a.w_blocked || a.sw_gu || a.lnf_mode) return false;
  if (act < 0 || act > 2) return false;
  if (act != 0 && (!out_bf16 || has_res)) return false;           // GELU / rope: bf16 out, no residual (what the step launches)
  if (gemm_v7_two_per_cu(variant) && act == 1) return false;      // (the GELU table does not fit beside two 72-KB rings)
  if ((a.K / BK) / a.splits < 2) return false;                    // >= 4 k-steps of 32 per tile
  if (a.lda * 2 * 256 >= (1L << 32) || (long)a.K * 2 * 320 >= (1L << 32)) return false;
  if (!a.a_plain && ((long)(a.M / a.a_rpb + 1) * a.a_bs + a.lda * a.a_rpb) * 2 >= (1L << 32)) return false;
  return true;
}

template <int MI, int NJ, int ACT, bool OUT_BF16, bool HAS_RES, int NS = V7_NS, int WPC = 1>
static int v7_launch_one(const GemmArgs& a, int pgrid, hipStream_t st) {
  TA_LAUNCH((gemm_nt_kernel_v7<MI, NJ, ACT, OUT_BF16, HAS_RES, NS, WPC>), dim3(pgrid), dim3(256), 0, st, a);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

template <int ACT, bool OUT_BF16, bool HAS_RES>
int launch_gemm_v7(int variant, GemmArgs a, int pgrid, hipStream_t st) {
  a.dbg &= ~2;                                                    // (the one-K-tile experiment bit does not exist here)
  if constexpr (ACT > 2 || (ACT != 0 && (!OUT_BF16 || HAS_RES))) {
    return TA_ERR_ARG;
  } else {
    if (variant == 13) return v7_launch_one<8, 8, ACT, OUT_BF16, HAS_RES>(a, pgrid, st);
    if (variant == 14) return v7_launch_one<8, 10, ACT, OUT_BF16, HAS_RES>(a, pgrid, st);
    if (variant == 15) return v7_launch_one<6, 8, ACT, OUT_BF16, HAS_RES>(a, pgrid, st);
    if constexpr (ACT != 1) {
      if (variant == 16) return v7_launch_one<4, 8, ACT, OUT_BF16, HAS_RES, 3, 2>(a, pgrid, st);
      if (variant == 17) return v7_launch_one<8, 4, ACT, OUT_BF16, HAS_RES, 3, 2>(a, pgrid, st);
    }
    return TA_ERR_ARG;
  }
}

#define V7_INST(ACT, OB, HR) template int launch_gemm_v7<ACT, OB, HR>(int, GemmArgs, int, hipStream_t);
V7_INST(0, true, true) V7_INST(0, true, false) V7_INST(0, false, true) V7_INST(0, false, false)
V7_INST(1, true, true) V7_INST(1, true, false) V7_INST(1, false, true) V7_INST(1, false, false)
V7_INST(2, true, false)
V7_INST(3, true, false) V7_INST(4, true, false) V7_INST(5, true, false) V7_INST(6, true, false)
