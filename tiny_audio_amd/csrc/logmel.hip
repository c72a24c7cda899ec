// ta355 log-mel front end (Whisper-style, as tiny-audio feeds GLM-ASR):
// TF:models/whisper/feature_extraction_whisper.py:135-168 -- reflect-pad 200, frames of 400 / hop 160,
// periodic Hann, |rDFT-400|^2 (last frame dropped), slaney mel bank [201 x n_mels], clamp 1e-10, log10,
// per-clip max(x, max-8), (x+4)/4 -- and :330-339 for the frame mask.
//
// n_fft = 400 is not a power of two and the whole stage is < 0.1 % of the step's flops, so the DFT is
// evaluated as an exact-f32 FMA chain against a host-built (float64-rounded) twiddle matrix instead of
// an FFT: each block stages FT windowed frames in LDS (coalesced waveform reads, reflect handled at load),
// thread k owns frequency bin k for all FT frames (twiddle rows read coalesced, L2-resident), powers go
// back to LDS and the same block applies the mel bank.  The clip maximum is an atomicMax on an
// order-preserving integer image of the float; a second tiny kernel applies floor/scale (the only
// second pass over the 512 KB/clip output).
#include "common.h"

#define NFFT 400
#define HOP 160
#define NBIN 201
#define FT 32            // frames per block
#define FRS 404          // LDS row stride (floats) for a frame

__device__ __forceinline__ int f2ord(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ __forceinline__ float ord2f(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }

// dft [NFFT][2*NBIN] f32: column k = cos(2 pi k n / 400), column NBIN + k = sin(..); window [NFFT]; melfb [NBIN][n_mels]
__global__ __launch_bounds__(256) void logmel_power_kernel(const float* __restrict__ wav, int Ls, const float* __restrict__ dft,
                                                           const float* __restrict__ window, const float* __restrict__ melfb,
                                                           int n_mels, float* __restrict__ out, int* __restrict__ clip_max,
                                                           int T) {
  __shared__ __attribute__((aligned(16))) float fr[FT * FRS];    // frames, later reused for powers [FT][NBIN(+pad)]
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
  float re[FT], im[FT];
#pragma unroll
  for (int f = 0; f < FT; ++f) { re[f] = 0.f; im[f] = 0.f; }
  const int k = tid < NBIN ? tid : NBIN - 1;
  for (int n4 = 0; n4 < NFFT; n4 += 4) {
    float c[4], s[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { c[j] = dft[(long)(n4 + j) * (2 * NBIN) + k]; s[j] = dft[(long)(n4 + j) * (2 * NBIN) + NBIN + k]; }
#pragma unroll
    for (int f = 0; f < FT; ++f) {
      const float4 x = *(const float4*)(fr + f * FRS + n4);
      re[f] = fmaf(x.x, c[0], re[f]); im[f] = fmaf(x.x, s[0], im[f]);
      re[f] = fmaf(x.y, c[1], re[f]); im[f] = fmaf(x.y, s[1], im[f]);
      re[f] = fmaf(x.z, c[2], re[f]); im[f] = fmaf(x.z, s[2], im[f]);
      re[f] = fmaf(x.w, c[3], re[f]); im[f] = fmaf(x.w, s[3], im[f]);
    }
  }
  __syncthreads();
  if (tid < NBIN) {
#pragma unroll
    for (int f = 0; f < FT; ++f) fr[f * FRS + tid] = re[f] * re[f] + im[f] * im[f];
  }
  __syncthreads();
  // mel + log10: thread handles mel bin m = tid % n_mels for frames f = tid / n_mels, step 256 / n_mels
  float lmax = -INFINITY;
  const int fstep = 256 / n_mels;     // n_mels in {64, 128, 256}: validated on the host
  const int m = tid % n_mels;
  for (int f = tid / n_mels; f < FT; f += fstep) {
    if (t0 + f >= T) break;
    float acc = 0.f;
    for (int kk = 0; kk < NBIN; ++kk) acc = fmaf(melfb[kk * n_mels + m], fr[f * FRS + kk], acc);
    const float v = log10f(fmaxf(acc, 1e-10f));
    out[((long)b * n_mels + m) * T + t0 + f] = v;
    lmax = fmaxf(lmax, v);
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
