// ta355 log-mel front end (Whisper-style, as tiny-audio feeds GLM-ASR):
// TF:models/whisper/feature_extraction_whisper.py:135-168 -- reflect-pad 200, frames of 400 / hop 160,
// periodic Hann, |rDFT-400|^2 (last frame dropped), slaney mel bank [201 x n_mels], clamp 1e-10, log10,
// per-clip max(x, max-8), (x+4)/4 -- and :330-339 for the frame mask.
//
// n_fft = 400 is not a power of two and the whole stage is < 0.1 % of the step's flops, so the DFT is
// evaluated as an exact-f32 matrix product against a host-built (float64-rounded) twiddle matrix instead of
// an FFT (see logmel_power_kernel).  The clip maximum is an atomicMax on an
// order-preserving integer image of the float; a second tiny kernel applies floor/scale (the only
// second pass over the 512 KB/clip output).
#include "common.h"

#define NFFT 400
#define HOP 160
#define NBIN 201

__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// dft [NFFT][2*NBIN] f32: column k = cos(2 pi k n / 400), column NBIN + k = sin(..); window [NFFT]; melfb [NBIN][n_mels]
//
// The DFT is a GEMM  [frames x 400] x [400 x 402]  in EXACT f32 on the matrix cores: v_mfma_f32_16x16x4_f32 is a
// fused-multiply-add chain over k, i.e. the same arithmetic as the scalar FMA loop it replaces (5.5x faster: the
// scalar loop ran at 18 TFLOP/s of the 157 TFLOP/s f32 matrix peak).  One workgroup = FT frames of one clip:
// windowed frames staged in LDS (the A operand, one ds_read per lane per k-step), the twiddle table streams from
// L2 (B operand, 64-B coalesced rows); wave w owns column blocks w, w+4, ...; re / im land back in LDS over the
// frames, become powers in place, and the same workgroup applies the mel bank.
#define FT 64            // frames per block
#define FRS 420          // LDS row stride (floats): >= 26 column blocks of 16 for re|im, and 400 samples of a frame
#define NCB 26           // ceil(402 / 16)
typedef __attribute__((ext_vector_type(4))) float lm_f32x4;

__global__ __launch_bounds__(256) void logmel_power_kernel(const float* __restrict__ wav, int Ls, const float* __restrict__ dft,
                                                           const float* __restrict__ window, const float* __restrict__ melfb,
                                                           int n_mels, float* __restrict__ out, int* __restrict__ clip_max,
                                                           int T) {
  __shared__ __attribute__((aligned(16))) float fr[FT * FRS];    // frames -> re|im -> powers
  const int b = blockIdx.y, t0 = blockIdx.x * FT, tid = threadIdx.x;
  const float* w = wav + (long)b * Ls;
  for (int i = tid; i < FT * NFFT; i += 256) {
    const int f = i / NFFT, n = i - f * NFFT;
    int j = (t0 + f) * HOP + n - NFFT / 2;          // index into the (virtually) reflect-padded signal
    if (j < 0) j = -j;
    if (j >= Ls) j = 2 * (Ls - 1) - j;
    float v = 0.f;
    if (t0 + f < T && j >= 0 && j < Ls) v = w[j] * window[n];
    fr[f * FRS + n] = v;
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
  lm_f32x4 acc[4][7];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int c = 0; c < 7; ++c) acc[rb][c] = (lm_f32x4){0.f, 0.f, 0.f, 0.f};
  float bv[7], bn[7];
  auto load_b = [&](int k0, float* dst) {
    const float* drow = dft + (long)(k0 + lg) * (2 * NBIN);
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      const int col = (wave + 4 * c) * 16 + li;
      dst[c] = col < 2 * NBIN ? drow[col] : 0.f;                        // (column blocks >= NCB read 0 and are never stored)
    }
  };
  load_b(0, bv);
  for (int k0 = 0; k0 < NFFT; k0 += 4) {
    if (k0 + 4 < NFFT) load_b(k0 + 4, bn);                              // next twiddle rows fly under this step's MFMAs
    float a[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) a[rb] = fr[(rb * 16 + li) * FRS + k0 + lg];
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      if (wave + 4 * c < NCB) {                                         // wave-uniform
#pragma unroll
        for (int rb = 0; rb < 4; ++rb) acc[rb][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[rb], bv[c], acc[rb][c], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < 7; ++c) bv[c] = bn[c];
  }
  __syncthreads();                                                      // every wave is done with the frames
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int c = 0; c < 7; ++c) {
      const int cb = wave + 4 * c;
      if (cb < NCB) {
#pragma unroll
        for (int r = 0; r < 4; ++r) fr[(rb * 16 + lg * 4 + r) * FRS + cb * 16 + li] = acc[rb][c][r];
      }
    }
  __syncthreads();
  // |X|^2, re-laid out bin-major pw[k][frame] over the same LDS (every thread first reads all of its re / im values)
  constexpr int PER = (FT * NBIN + 255) / 256;
  float pwv[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * 256;
    if (i < FT * NBIN) {
      const int k = i / FT, f = i - k * FT;
      const float re = fr[f * FRS + k], im = fr[f * FRS + NBIN + k];
      pwv[q] = re * re + im * im;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = tid + q * 256;
    if (i < FT * NBIN) fr[i] = pwv[q];                                  // i = k * FT + f
  }
  __syncthreads();
  // mel + log10: thread = (mel bin m, group of FT / fgroups frames); the filter weight is loaded once per bin and the
  // powers of 4 frames come as one broadcast 16-B LDS read.  Same fmaf order over the bins as a scalar loop.
  float lmax = -INFINITY;
  const int fgroups = 256 / n_mels;   // n_mels in {64, 128, 256}: validated on the host
  const int m = tid % n_mels, fg = tid / n_mels, nf = FT / fgroups;      // nf in {16, 32, 64}
  for (int fb = 0; fb < nf; fb += 16) {
    float am[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) am[q] = 0.f;
    const int f0 = fg * nf + fb;
    for (int kk = 0; kk < NBIN; ++kk) {
      const float wgt = melfb[kk * n_mels + m];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 pv = *(const float4*)(fr + kk * FT + f0 + 4 * q);
        am[4 * q] = fmaf(wgt, pv.x, am[4 * q]); am[4 * q + 1] = fmaf(wgt, pv.y, am[4 * q + 1]);
        am[4 * q + 2] = fmaf(wgt, pv.z, am[4 * q + 2]); am[4 * q + 3] = fmaf(wgt, pv.w, am[4 * q + 3]);
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int t = t0 + f0 + q;
      if (t < T) {
        const float v = log10f(fmaxf(am[q], 1e-10f));
        out[((long)b * n_mels + m) * T + t] = v;
        lmax = fmaxf(lmax, v);
      }
    }
  }
  lmax = wave_max(lmax);
  if ((tid & 63) == 0 && lmax > -INFINITY) atomicMax(clip_max + b, f2ord(lmax));
}

__global__ void logmel_init_kernel(int* clip_max, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) clip_max[i] = f2ord(-INFINITY);
}

// x = (max(x, clipmax - 8) + 4) / 4 in place; also the frame mask [B, T]: 1 iff t*160 < len[b]
__global__ void logmel_finalize_kernel(float* __restrict__ out, const int* __restrict__ clip_max, const long* __restrict__ lens,
                                       int* __restrict__ mask, int n_mels, int T) {
  const int b = blockIdx.y;
  const float floorv = ord2f(clip_max[b]) - 8.0f;
  const long n = (long)n_mels * T;
  float* o = out + (long)b * n;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    o[i] = (fmaxf(o[i], floorv) + 4.0f) * 0.25f;
  if (mask && blockIdx.x == 0)
    for (int t = threadIdx.x; t < T; t += blockDim.x) mask[(long)b * T + t] = ((long)t * HOP < lens[b]) ? 1 : 0;
}

// wav f32 [B, Ls] (zero-padded to the longest clip), lens int64 [B] -> feats f32 [B, n_mels, T], mask i32 [B, T],
// T = Ls / 160.  clip_max_ws: int[B] workspace.
extern "C" int ta_logmel_f32(const float* wav, const long* lens, int B, int Ls, const float* dft, const float* window,
                             const float* melfb, int n_mels, float* feats, int* mask, int* clip_max_ws, hipStream_t st) {
  if (B <= 0) return TA_OK;
  const int T = Ls / HOP;
  if (T <= 0 || Ls <= NFFT / 2 || (n_mels != 64 && n_mels != 128 && n_mels != 256)) return TA_ERR_ARG;
  TA_LAUNCH(logmel_init_kernel, dim3(ta_cdiv(B, 256)), dim3(256), 0, st, clip_max_ws, B);
  TA_LAUNCH(logmel_power_kernel, dim3(ta_cdiv(T, FT), B), dim3(256), 0, st, wav, Ls, dft, window, melfb, n_mels,
                     feats, clip_max_ws, T);
  int gx = ta_cdiv((long)n_mels * T, 256 * 4); if (gx < 1) gx = 1;
  TA_LAUNCH(logmel_finalize_kernel, dim3(gx, B), dim3(256), 0, st, feats, clip_max_ws, lens, mask, n_mels, T);
  TA_CHECK_LAUNCH();
  return TA_OK;
}
