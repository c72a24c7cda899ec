// ta355 log-mel front end (Whisper-style, as tiny-audio feeds GLM-ASR):
// TF:models/whisper/feature_extraction_whisper.py:135-168 -- reflect-pad 200, frames of 400 / hop 160,
// periodic Hann, |rDFT-400|^2 (last frame dropped), slaney mel bank [201 x n_mels], clamp 1e-10, log10,
// per-clip max(x, max-8), (x+4)/4 -- and :330-339 for the frame mask.
//
// n_fft = 400 = 16 x 25: a mixed-radix complex FFT ((4.4) x (5.5)), two real frames per transform, f32 throughout (round 2; round
// 1's exact-f32 DFT-as-GEMM on v_mfma_f32_16x16x4_f32 -- 262-278 us per 32 clips against 42 -- left the library in round 5).
// The clip maximum: every persistent workgroup records the maximum of its tiles per clip (plain stores into its own scratch records,
// round 5); the finalize kernel reduces a clip's records and applies floor/scale (the only second pass over the 512 KB/clip output).
#include <cstdlib>
#include "common.h"

#define NFFT 400
#define HOP 160
#define NBIN 201


typedef __attribute__((ext_vector_type(4))) float lm_f32x4;


// ============================================================================ FFT form (default)
// The 400-point transform as a mixed-radix FFT, 400 = 16 x 25 = (4 x 4) x (5 x 5), in f32 on the VALU: 17 MFLOP per clip
// instead of the 322 MFLOP of round 1's DFT-as-GEMM, which makes the stage what the north star asks of it -- bound by its
// 1.15 MB of HBM traffic per clip, not by arithmetic.  (f32 FFT error ~1e-7 of the largest bin, below the O(N) error of a
// direct f32 summation; same 5e-4 tolerance against the reference's torch.stft as before.)
//   X[k1 + 16 k2] = sum_{n2} W25^{n2 k2} * ( W400^{n2 k1} * sum_{n1} W16^{n1 k1} z[25 n1 + n2] )
// Two REAL frames ride in one complex transform, z = w * (x_a + i x_b):  X_a[k] = (Z[k] + conj Z[400-k]) / 2,
// X_b[k] = (Z[k] - conj Z[400-k]) / 2i.  One workgroup = 64 frames of one clip (the waveform span of those frames is staged
// once: 10 480 samples instead of 64 x 400 windowed copies); each wave transforms 4 frame pairs at a time:
//   stage A  lane = (pair, n2): 16 samples of both frames from the span, windowed; 16-point FFT in registers (two radix-4
//            passes); twiddle; Y[pair][k1][n2] to the wave's LDS scratch
//   stage B  lane = (pair, k1): 25-point FFT in registers (two radix-5 passes); Z[pair][k] back to the same scratch
//   combine  lane = (pair, k <= 200): the two frames' powers -> pw[frame][k]
// then, as before, the mel bank (each filter over its own bin range only), log10, clip maximum, coalesced write-out.
#include "logmel_twiddles.h"
struct cpx { float re, im; };
__device__ __forceinline__ cpx cmul(cpx a, float wr, float wi) { return {a.re * wr - a.im * wi, a.re * wi + a.im * wr}; }
__device__ __forceinline__ void dft4(const cpx a0, const cpx a1, const cpx a2, const cpx a3, cpx& o0, cpx& o1, cpx& o2, cpx& o3) {
  const cpx s02 = {a0.re + a2.re, a0.im + a2.im}, d02 = {a0.re - a2.re, a0.im - a2.im};
  const cpx s13 = {a1.re + a3.re, a1.im + a3.im}, d13 = {a1.re - a3.re, a1.im - a3.im};
  o0 = {s02.re + s13.re, s02.im + s13.im};
  o2 = {s02.re - s13.re, s02.im - s13.im};
  o1 = {d02.re + d13.im, d02.im - d13.re};          // d02 - i d13
  o3 = {d02.re - d13.im, d02.im + d13.re};          // d02 + i d13
}
__device__ __forceinline__ void dft5(const cpx x0, const cpx x1, const cpx x2, const cpx x3, const cpx x4, cpx& o0, cpx& o1, cpx& o2,
                                     cpx& o3, cpx& o4) {
  const cpx t1 = {x1.re + x4.re, x1.im + x4.im}, t2 = {x2.re + x3.re, x2.im + x3.im};
  const cpx t3 = {x1.re - x4.re, x1.im - x4.im}, t4 = {x2.re - x3.re, x2.im - x3.im};
  o0 = {x0.re + t1.re + t2.re, x0.im + t1.im + t2.im};
  const cpx m1 = {x0.re + R5_C1 * t1.re + R5_C2 * t2.re, x0.im + R5_C1 * t1.im + R5_C2 * t2.im};
  const cpx m2 = {x0.re + R5_C2 * t1.re + R5_C1 * t2.re, x0.im + R5_C2 * t1.im + R5_C1 * t2.im};
  const cpx n1 = {R5_S1 * t3.re + R5_S2 * t4.re, R5_S1 * t3.im + R5_S2 * t4.im};
  const cpx n2 = {R5_S2 * t3.re - R5_S1 * t4.re, R5_S2 * t3.im - R5_S1 * t4.im};
  o1 = {m1.re + n1.im, m1.im - n1.re};              // m1 - i n1
  o4 = {m1.re - n1.im, m1.im + n1.re};              // m1 + i n1
  o2 = {m2.re + n2.im, m2.im - n2.re};
  o3 = {m2.re - n2.im, m2.im + n2.re};
}
// in-register 16-point FFT: n1 = 4p + q, k1 = r + 4s
__device__ __forceinline__ void fft16(cpx* a) {
  cpx u[4][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    dft4(a[q], a[4 + q], a[8 + q], a[12 + q], u[q][0], u[q][1], u[q][2], u[q][3]);
#pragma unroll
    for (int r = 1; r < 4; ++r)
      if (q > 0) u[q][r] = cmul(u[q][r], W16Q_RE[q][r], W16Q_IM[q][r]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) dft4(u[0][r], u[1][r], u[2][r], u[3][r], a[r], a[r + 4], a[r + 8], a[r + 12]);
}
// in-register 25-point FFT: n2 = 5p + q, k2 = r + 5s
__device__ __forceinline__ void fft25(cpx* y) {
  cpx u[5][5];
#pragma unroll
  for (int q = 0; q < 5; ++q) {
    dft5(y[q], y[5 + q], y[10 + q], y[15 + q], y[20 + q], u[q][0], u[q][1], u[q][2], u[q][3], u[q][4]);
#pragma unroll
    for (int r = 1; r < 5; ++r)
      if (q > 0) u[q][r] = cmul(u[q][r], W25Q_RE[q][r], W25Q_IM[q][r]);
  }
#pragma unroll
  for (int r = 0; r < 5; ++r) dft5(u[0][r], u[1][r], u[2][r], u[3][r], u[4][r], y[r], y[r + 5], y[r + 10], y[r + 15], y[r + 20]);
}

#define LF_PWS 201                          // power row stride (odd: the mel stage's reads spread over the banks)
// LFT frames per workgroup (4 waves), LF_G frame pairs per wave at once.  <64, 4> fills every lane of stage B but needs
// 150 KB of LDS (one workgroup, i.e. one wave per SIMD, per CU); <32, 2> needs 78 KB: two workgroups per CU.
// MFMA (round 4): the two sub-transforms as real matrix products on v_mfma_f32_16x16x4_f32 (exact f32 FMA chains) instead of register
// FFTs on 25 / 16 lanes per frame pair.  Stage A: [Re Y; Im Y] (32 x cols) = DFT16_A (32 x 32) [zr; zi], columns = (pair, n2) -- the
// two pairs of an iteration are 50 columns = 4 column blocks; the B operand is ONE windowed sample per lane per k-step, straight from
// the span.  Twiddle in registers (a lane holds Re and Im of the same (k1, column)), Y' to the wave's scratch as [pair][Re|Im][n2][k1]
// so that stage B's B operand (columns = k1, k = (Yr n2 | Yi n2)) is 16 consecutive floats per lane group.  Stage B: 4 row blocks
// (Re / Im x k2 < 16 / k2 >= 16) x 13 k-steps per pair; Z lands in the scratch in the interleaved [k][re, im] form the combine reads.
// Per frame: 21 + 26 MFMAs of 32 cycles on the CU's four matrix cores (~310 cycles) against ~1 150 cycles of partially filled VALU
// waves; the mel stage and everything around it are unchanged.
template <int LFT, int LF_G, bool MFMA = false>
__global__ __launch_bounds__(256) void logmel_fft_kernel(const float* __restrict__ wav, int Ls, const float* __restrict__ dft,
                                                         const float* __restrict__ window, const float* __restrict__ melfb,
                                                         int n_mels, float* __restrict__ out, float* __restrict__ wg_rec, int T, int dbg,
                                                         const int* __restrict__ mrange_g, int nseg,
                                                         const long* __restrict__ lens, int* __restrict__ mask, int nblk, int ntiles) {
  constexpr int LF_SPAN = (LFT - 1) * HOP + NFFT, LF_SCR = LF_G * NFFT * 2, LF_MELS = LFT + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* span = lds;                              // [LF_SPAN]            (later: melbuf [n_mels][LF_MELS])
  float* scr = span + LF_SPAN;                    // [4 waves][LF_SCR]
  float* pw = scr + 4 * LF_SCR;                   // [FT][LF_PWS]
  float* win = pw + LFT * LF_PWS;                  // [NFFT]
  float* tw = win + NFFT;                         // [NFFT][2] = W400^j
  int* mrange = (int*)(tw + 2 * NFFT);            // [n_mels][2] first / end bin of every mel filter
  // round 4: a workgroup is PERSISTENT over tiles (tile = LFT frames of one clip; 1-D grid of about two workgroups per CU) and
  // requests the NEXT tile's span -- six 16-byte loads per thread, one round trip -- before it transforms the current one.  Measured
  // (rocprofv3, profiles/r04_za_logmel_kernel_times.txt): the kernel took 48.7 us with neither the transform nor the mel stage in it
  // (68.2 with both): a workgroup that lives for one tile spends its life in four serialized round trips (three span batches, the
  // tables) and the store drain, two resident per CU, two rounds.
  const int tid = threadIdx.x;
  constexpr bool legacy = false;                   // (rounds 2-3 launched one workgroup per tile on a 2-D grid)
  // a persistent workgroup takes CONSECUTIVE tiles (the same clip, mostly): its clip maximum is then ONE device-scope atomic at the
  // end -- the kernel spent 25 of its 65 us waiting on 4 096 atomicMax to 32 addresses, one per wave and tile, each a memory-side
  // round trip that the next barrier waited for (profiles/r04_zc_logmel_floor.txt: 41.1 us without transform and mel taps, 16.3
  // without the mel stage and its atomic)
  const int per = legacy ? 1 : (ntiles + (int)gridDim.x - 1) / (int)gridDim.x;
  const int tile0 = legacy ? (int)(blockIdx.y * gridDim.x + blockIdx.x) : (int)blockIdx.x * per;
  constexpr int tstride = 1;
  const int tend = legacy ? tile0 + 1 : (tile0 + per < ntiles ? tile0 + per : ntiles);
  if (legacy) nblk = gridDim.x;
  float run_max = -INFINITY;                       // the workgroup's maximum over the tiles of clip run_b so far
  int run_b = -1;
  constexpr int PF = (LF_SPAN / 4 + 255) / 256;
  float4 pre[PF];
  // a tile whose span lies inside the clip and on a 16-byte boundary (t0 * 160 - 200 is a multiple of 8 samples) is loaded as float4
  const bool al = (Ls % 4 == 0) && ((((size_t)wav) & 15) == 0);
  auto fast = [&](int tile) { const int s0 = (tile % nblk) * LFT * HOP - NFFT / 2; return al && s0 >= 0 && s0 + LF_SPAN <= Ls; };
#define LM_REQUEST(tile_)                                                                                             \
  do {                                                                                                                \
    const float* src_ = wav + (long)((tile_) / nblk) * Ls + ((tile_) % nblk) * LFT * HOP - NFFT / 2;                  \
    _Pragma("unroll") for (int u = 0; u < PF; ++u) {                                                                  \
      const int q = tid + u * 256;                                                                                    \
      pre[u] = q < LF_SPAN / 4 ? *(const float4*)(src_ + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);                    \
    }                                                                                                                 \
  } while (0)
  if (tile0 < tend && fast(tile0)) LM_REQUEST(tile0);
  for (int j = tid; j < NFFT; j += 256) {
    win[j] = window[j];
    // W400^j = cos(2 pi j / 400) - i sin(2 pi j / 400) from row n = 1 of the host's table (columns k <= 200; mirrored above)
    // (MFMA form: entry j = n2 * 16 + k1 holds W400^(n2 k1), so a lane's four k1 are 32 contiguous bytes)
    const int je = MFMA ? (j >> 4) * (j & 15) : j;
    const int jj = je <= 200 ? je : NFFT - je;
    const float c = dft[2 * NBIN + jj], sn = dft[2 * NBIN + NBIN + jj];
    tw[2 * j] = c; tw[2 * j + 1] = je <= 200 ? -sn : sn;
  }
  for (int i = tid; i < 2 * n_mels; i += 256) mrange[i] = mrange_g[i];     // per-filter bin ranges (logmel_init_kernel)
  const int wave = tid >> 6, lane = tid & 63;
  float* my = scr + wave * LF_SCR;
  [[maybe_unused]] const int li = lane & 15, lg = lane >> 4;
  [[maybe_unused]] float a16[2][8], a25[4][13];
  if constexpr (MFMA) {
    static_assert(LF_G == 2, "the MFMA form works on two frame pairs (50 columns = 4 column blocks) at a time");
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) a16[rb][ks] = DFT16_A[rb][4 * ks + lg][li];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int ks = 0; ks < 13; ++ks) a25[rb][ks] = DFT25_A[rb][4 * ks + lg][li];
  }
  for (int tile = tile0; tile < tend; tile += tstride) {
  const int b = tile / nblk, t0 = (tile % nblk) * LFT;
  const float* w = wav + (long)b * Ls;
  __syncthreads();                                                // the previous tile's write-out is done with melbuf (= span)
  if (fast(tile)) {
#pragma unroll
    for (int u = 0; u < PF; ++u) { const int q = tid + u * 256; if (q < LF_SPAN / 4) *(float4*)(span + 4 * q) = pre[u]; }
  } else {
    // first / last tile of a clip (reflect padding, zero fill) or an unaligned clip: scalar, in batches of 8 independent loads
    for (int j0 = tid; j0 < LF_SPAN; j0 += 256 * 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u * 256;
        int sidx = t0 * HOP + j - NFFT / 2;         // index into the (virtually) reflect-padded signal
        if (sidx < 0) sidx = -sidx;
        if (sidx >= Ls) sidx = 2 * (Ls - 1) - sidx;
        v[u] = (j < LF_SPAN && sidx >= 0 && sidx < Ls) ? w[sidx] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int j = j0 + u * 256; if (j < LF_SPAN) span[j] = v[u]; }
    }
  }
  __syncthreads();
  if (tile + tstride < tend && fast(tile + tstride)) LM_REQUEST(tile + tstride);     // in flight under this tile's transform
  for (int it = 0; it < ((dbg & 1) ? 0 : LFT / 2 / 4 / LF_G); ++it) {            // LFT / 2 pairs per workgroup, a quarter per wave, LF_G at a time
    const int pair0 = wave * (LFT / 8) + it * LF_G;
    if constexpr (MFMA) {
      // ---- stage A: two column blocks at a time (four independent accumulator chains).  Per lane and column block everything that
      // depends on the column is ONE base offset into the span and one into the window; the k-steps are immediate offsets from them.
#pragma unroll
      for (int cb2 = 0; cb2 < 4; cb2 += 2) {
        lm_f32x4 acc[2][2];
        const float* sp[2]; const float* wp[2]; float ma[2], mb[2]; int n2v[2], pv[2]; bool okc[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          acc[q][0] = (lm_f32x4){0.f, 0.f, 0.f, 0.f}; acc[q][1] = (lm_f32x4){0.f, 0.f, 0.f, 0.f};
          const int c = (cb2 + q) * 16 + li;
          okc[q] = c < 50;
          pv[q] = c >= 25 ? 1 : 0;
          n2v[q] = okc[q] ? c - 25 * pv[q] : 0;
          const int fa = 2 * (pair0 + pv[q]);
          sp[q] = span + fa * HOP + 25 * lg + n2v[q];
          wp[q] = win + 25 * lg + n2v[q];
          ma[q] = (okc[q] && t0 + fa < T) ? 1.f : 0.f;                          // frame a valid (real part)
          mb[q] = (okc[q] && t0 + fa + 1 < T) ? 1.f : 0.f;                      // frame b valid (imaginary part)
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            // k = 4 ks + lg: k < 16 -> zr[n1 = k] (frame a), else zi[n1 = k - 16] (frame b); sample 25 n1 + n2 of the frame
            const float wv = wp[q][100 * (ks & 3)] * (ks < 4 ? ma[q] : mb[q]);
            const float b = sp[q][100 * (ks & 3) + (ks < 4 ? 0 : HOP)] * wv;
            acc[q][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a16[0][ks], b, acc[q][0], 0, 0, 0);
            acc[q][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a16[1][ks], b, acc[q][1], 0, 0, 0);
          }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (okc[q]) {
            const float4 t0v = *(const float4*)(tw + (n2v[q] * 16 + 4 * lg) * 2), t1v = *(const float4*)(tw + (n2v[q] * 16 + 4 * lg) * 2 + 4);
            const float tr[4] = {t0v.x, t0v.z, t1v.x, t1v.z}, ti[4] = {t0v.y, t0v.w, t1v.y, t1v.w};
            float4 yr, yi;
            float* pr = &yr.x; float* pi = &yi.x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              pr[j] = acc[q][0][j] * tr[j] - acc[q][1][j] * ti[j];
              pi[j] = acc[q][0][j] * ti[j] + acc[q][1][j] * tr[j];
            }
            float* dst = my + ((pv[q] * 2) * 25 + n2v[q]) * 16 + 4 * lg;
            *(float4*)dst = yr;
            *(float4*)(dst + 25 * 16) = yi;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      // ---- stage B: both pairs interleaved (eight accumulator chains)
      lm_f32x4 z[2][4];
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) z[p][rb] = (lm_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 13; ++ks) {
        const int k = 4 * ks + lg;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const float b = k < 50 ? my[p * 800 + k * 16 + li] : 0.f;
#pragma unroll
          for (int rb = 0; rb < 4; ++rb) z[p][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a25[rb][ks], b, z[p][rb], 0, 0, 0);
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      __builtin_amdgcn_wave_barrier();                            // every Y' has been read: Z takes its place
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int hb = 0; hb < 2; ++hb)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k2 = hb * 16 + 4 * lg + j;
            if (k2 < 25) *(float2*)(my + (p * NFFT + li + 16 * k2) * 2) = make_float2(z[p][hb][j], z[p][hb + 2][j]);
          }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    } else {
    // ---- stage A
#pragma unroll 1
    for (int pass = 0; pass < (LF_G * 25 + 63) / 64; ++pass) {
      const int id = lane + 64 * pass;
      if (id < LF_G * 25) {
        const int g = id / 25, n2 = id - g * 25;
        const int fa = 2 * (pair0 + g);
        const bool va = t0 + fa < T, vb = t0 + fa + 1 < T;
        const float* xa = span + fa * HOP + n2;
        cpx a[16];
#pragma unroll
        for (int n1 = 0; n1 < 16; ++n1) {
          const float wv = win[25 * n1 + n2];
          a[n1].re = va ? xa[25 * n1] * wv : 0.f;
          a[n1].im = vb ? xa[HOP + 25 * n1] * wv : 0.f;
        }
        fft16(a);
        float* dst = my + ((g * 16) * 25 + n2) * 2;
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
          cpx v = a[k1];
          if (k1 > 0) {
            const int j = n2 * k1;                                // < 400
            v = cmul(v, tw[2 * j], tw[2 * j + 1]);
          }
          *(float2*)(dst + k1 * 50) = make_float2(v.re, v.im);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // ---- stage B: lane = (pair g, k1)
    if (lane < LF_G * 16) {
      const int g = lane >> 4, k1 = lane & 15;
      const float* src = my + ((g * 16 + k1) * 25) * 2;
      cpx y[25];
#pragma unroll
      for (int n2 = 0; n2 < 25; ++n2) { const float2 v = *(const float2*)(src + 2 * n2); y[n2] = {v.x, v.y}; }
      fft25(y);                                                   // (every active lane holds its inputs: in-order LDS per wave)
      float* dst = my + (g * NFFT + k1) * 2;
#pragma unroll
      for (int k2 = 0; k2 < 25; ++k2) *(float2*)(dst + 32 * k2) = make_float2(y[k2].re, y[k2].im);     // k = k1 + 16 k2
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    }   // (register FFT form)
    // ---- combine: powers of the two real frames of every pair
    for (int id = lane; id < LF_G * NBIN; id += 64) {
      const int g = id / NBIN, k = id - g * NBIN;
      const float2 z = *(const float2*)(my + (g * NFFT + k) * 2);
      const float2 c = *(const float2*)(my + (g * NFFT + (k == 0 ? 0 : NFFT - k)) * 2);
      const float ar = 0.5f * (z.x + c.x), ai = 0.5f * (z.y - c.y);
      const float br = 0.5f * (z.y + c.y), bi = 0.5f * (c.x - z.x);
      const int fa = 2 * (pair0 + g);
      pw[fa * LF_PWS + k] = ar * ar + ai * ai;
      pw[(fa + 1) * LF_PWS + k] = br * br + bi * bi;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();                              // the next iteration overwrites the scratch
  }
  __syncthreads();
  // ---- mel + log10: thread = (mel bin m, group of frames), every filter over its own bins only
  float* melbuf = span;                                           // [n_mels][LF_MELS]
  float lmax = -INFINITY;
  const int fgroups = 256 / n_mels;
  const int m = tid % n_mels, fg = tid / n_mels, nf = LFT / fgroups;
  const int klo = mrange[2 * m], khi = mrange[2 * m + 1];
  for (int fb = 0; fb < ((dbg & 16) ? 0 : nf); fb += 16) {
    float am[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) am[q] = 0.f;
    const int f0 = fg * nf + fb;
    for (int kk = klo; kk < ((dbg & 2) ? klo : khi); ++kk) {
      const float wgt = melfb[kk * n_mels + m];
#pragma unroll
      for (int q = 0; q < 16; ++q) am[q] = fmaf(wgt, pw[(f0 + q) * LF_PWS + kk], am[q]);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const float v = log10f(fmaxf(am[q], 1e-10f));
      melbuf[m * LF_MELS + f0 + q] = v;
      if (t0 + f0 + q < T) lmax = fmaxf(lmax, v);
    }
  }
  lmax = wave_max(lmax);
  float* wred = (float*)(mrange + 2 * n_mels) + 1;                // [4] behind the mel ranges and the rendezvous float (sized by the host)
  if ((tid & 63) == 0) wred[wave] = lmax;
  __syncthreads();
  const float wm = fmaxf(fmaxf(wred[0], wred[1]), fmaxf(wred[2], wred[3]));
  // Round 5: the workgroup's maximum over its tiles of a clip goes to ITS OWN record -- a plain store, no atomic, nothing to
  // initialise (round 4 published it with one device-scope atomicMax per workgroup into a per-clip word that a separate launch had to
  // reset).  A workgroup's tiles are consecutive, so its records are its clips in order: record j belongs to clip (tile0 / nblk) + j;
  // the finalize kernel reduces the records of the workgroups that touched a clip (logmel_clip_floor).
  if (run_b != b) {                                               // (uniform) a new clip
    if (tid == 0 && run_b >= 0) wg_rec[(long)blockIdx.x * nseg + (run_b - tile0 / nblk)] = run_max;
    run_b = b; run_max = -INFINITY;
  }
  run_max = fmaxf(run_max, wm);
  if (tile + 1 >= tend && tid == 0) wg_rec[(long)blockIdx.x * nseg + (b - tile0 / nblk)] = run_max;
  const float floorv = -INFINITY, add = 0.f, mul = 1.f;           // (floor and scale are applied by the finalize pass)
  for (int i = tid; i < ((dbg & 8) ? 0 : n_mels * LFT); i += 256) {                 // one mel row = LFT consecutive frames per store
    const int r = i / LFT, f = i - r * LFT;
    if (t0 + f < T) out[((long)b * n_mels + r) * T + t0 + f] = (fmaxf(melbuf[r * LF_MELS + f], floorv) + add) * mul;
  }
  }   // tile loop
}

// {first, end} bin of every mel filter (its non-zero taps): the FFT kernel applies a filter over its own ~4 (low) to ~13 (high) bins
// instead of all 201.  Depends on the filter bank only: computed ONCE at table upload (ta_logmel_mel_ranges).
__global__ __launch_bounds__(256) void logmel_ranges_kernel(const float* __restrict__ melfb, int n_mels, int* __restrict__ mrange) {
  __shared__ int lo_s, hi_s;
  if (threadIdx.x == 0) { lo_s = NBIN; hi_s = 0; }
  __syncthreads();
  const int k = threadIdx.x, m = blockIdx.x;
  if (k < NBIN && melfb[k * n_mels + m] != 0.f) { atomicMin(&lo_s, k); atomicMax(&hi_s, k + 1); }
  __syncthreads();
  if (threadIdx.x == 0) { mrange[2 * m] = lo_s; mrange[2 * m + 1] = hi_s; }
}
// x = (max(x, clipmax - 8) + 4) / 4 in place; also the frame mask [B, T]: 1 iff t*160 < len[b].
// The (max - 8) floor of clip b comes from the records of the workgroups that worked on it (logmel_fft_kernel): workgroup g covers tiles
// [g * per, (g + 1) * per), clip b the tiles [b * nblk, (b + 1) * nblk); record j of workgroup g is its j-th clip.  One wave loads the
// records (one per lane, independent loads) and reduces them.
__global__ void logmel_finalize_kernel(float* __restrict__ out, const float* __restrict__ wg_rec, const long* __restrict__ lens,
                                       int* __restrict__ mask, int n_mels, int T, int nblk, int per, int nseg, int ntiles) {
  const int b = blockIdx.y;
  __shared__ float floor_s;
  if (threadIdx.x < 64) {
    const int g_lo = (b * nblk) / per, last = min((b + 1) * nblk, ntiles) - 1, g_hi = last / per;
    float m = -INFINITY;
    for (int g = g_lo + (int)threadIdx.x; g <= g_hi; g += 64) m = fmaxf(m, wg_rec[(long)g * nseg + (b - (g * per) / nblk)]);
    m = wave_max(m);
    if (threadIdx.x == 0) floor_s = m - 8.0f;
  }
  __syncthreads();
  const float floorv = floor_s;
  const long n = (long)n_mels * T;
  float* o = out + (long)b * n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    o[i] = (fmaxf(o[i], floorv) + 4.0f) * 0.25f;
  if (mask && blockIdx.x == 0)
    for (int t = threadIdx.x; t < T; t += blockDim.x) mask[(long)b * T + t] = ((long)t * HOP < lens[b]) ? 1 : 0;
}

extern "C" int ta_logmel_mel_ranges(const float* melfb, int n_mels, int* mel_ranges, hipStream_t st) {
  if (!melfb || !mel_ranges || n_mels <= 0) return TA_ERR_ARG;
  TA_LAUNCH(logmel_ranges_kernel, dim3(n_mels), dim3(256), 0, st, melfb, n_mels, mel_ranges);
  TA_CHECK_LAUNCH();
  return TA_OK;
}

// launch geometry of ta_logmel_f32 for (B, Ls, n_mels): tiles of LFT frames, persistent workgroups over consecutive tiles
namespace {
struct LogmelGeom { int T, wide, lft, nblk, ntiles, grid, per, nseg; };
LogmelGeom logmel_geom(int B, int Ls, int n_mels) {
  static const int ncu = [] { int dev = 0, n = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev); return n > 0 ? n : 256; }();
  LogmelGeom g;
  g.T = Ls / HOP;
  // the mel stage works on groups of 16 frames per thread: LFT / (256 / n_mels) >= 16 holds for <32, 2> only at n_mels = 128
  g.wide = n_mels != 128; g.lft = g.wide ? 64 : 32;
  g.nblk = ta_cdiv(g.T > 0 ? g.T : 1, g.lft);
  g.ntiles = g.nblk * B;
  const int resident = ncu * (g.wide ? 1 : 2);
  g.grid = g.ntiles < resident ? g.ntiles : resident;
  g.per = ta_cdiv(g.ntiles, g.grid > 0 ? g.grid : 1);
  g.nseg = (g.per + g.nblk - 2) / g.nblk + 2;          // clips a run of `per` consecutive tiles can touch (upper bound)
  return g;
}
}  // namespace
// floats of scratch ta_logmel_f32 needs for this shape (per-workgroup clip maxima; no initial contents required)
extern "C" long ta_logmel_scratch_floats(int B, int Ls, int n_mels) {
  if (B <= 0 || Ls <= 0) return 1;
  const LogmelGeom g = logmel_geom(B, Ls, n_mels);
  return (long)g.grid * g.nseg;
}

// wav f32 [B, Ls] (zero-padded to the longest clip), lens int64 [B] -> feats f32 [B, n_mels, T], mask i32 [B, T],
// T = Ls / 160.  scratch: float[ta_logmel_scratch_floats(B, Ls, n_mels)] (per-workgroup clip maxima: no initial contents required, no
// atomics, TWO launches -- round 4: int[2 B] behind an init launch, one device-scope atomicMax per workgroup);
// mel_ranges: int[2 * n_mels] from ta_logmel_mel_ranges (computed once per filter bank).
extern "C" int ta_logmel_f32(const float* wav, const long* lens, int B, int Ls, const float* dft, const float* window,
                             const float* melfb, int n_mels, float* feats, int* mask, float* scratch, const int* mel_ranges,
                             hipStream_t st) {
  if (B <= 0) return TA_OK;
  const int T = Ls / HOP;
  if (T <= 0 || Ls <= NFFT / 2 || (n_mels != 64 && n_mels != 128 && n_mels != 256) || !scratch || !mel_ranges) return TA_ERR_ARG;
  const LogmelGeom g = logmel_geom(B, Ls, n_mels);
  {
    auto lds_of = [&](int lft, int gq) { return (size_t)((lft - 1) * HOP + NFFT + 4 * gq * NFFT * 2 + lft * LF_PWS + 3 * NFFT) * 4 + (size_t)n_mels * 2 * 4 + 32; };
    static bool attr = false;
    if (!attr) {
      (void)hipFuncSetAttribute((const void*)logmel_fft_kernel<64, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(64, 4));
      (void)hipFuncSetAttribute((const void*)logmel_fft_kernel<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_of(32, 2));
      attr = true;
    }
    // persistent workgroups over the tiles of all clips (round 4).  Measured and removed in round 5: one workgroup per tile (rounds 2-3:
    // 68 us against 42), a single-pass form whose workgroups rendezvous per clip (113.7 us against 80.9, profiles/r03_f_*), the
    // sub-transforms on the f32 matrix cores (76.4 against 68.2, profiles/r04_za_*; experiment builds only).
    const dim3 grid(g.grid);
    if (g.wide)
      TA_LAUNCH((logmel_fft_kernel<64, 4>), grid, dim3(256), lds_of(64, 4), st, wav, Ls, dft, window, melfb, n_mels, feats, scratch, T, 0, mel_ranges, g.nseg, lens, mask, g.nblk, g.ntiles);
    else
      TA_LAUNCH((logmel_fft_kernel<32, 2>), grid, dim3(256), lds_of(32, 2), st, wav, Ls, dft, window, melfb, n_mels, feats, scratch, T, 0, mel_ranges, g.nseg, lens, mask, g.nblk, g.ntiles);
  }
  {
    int gx = ta_cdiv((long)n_mels * T, 256 * 4); if (gx < 1) gx = 1;
    TA_LAUNCH(logmel_finalize_kernel, dim3(gx, B), dim3(256), 0, st, feats, (const float*)scratch, lens, mask, n_mels, T, g.nblk, g.per, g.nseg, g.ntiles);
  }
  TA_CHECK_LAUNCH();
  return TA_OK;
}
