// Where a fused decode-step kernel (csrc/decode_fused.hip) spends its time at B = 32 on Qwen3-0.6B shapes: every kernel is launched
// back-to-back over 28 weight sets (353 MB of gate|up > the 256 MB infinity cache, as in a real step), with the debug variants that
// drop the W loads, the X loads, the reduction + epilogue, or everything.  Prints us per launch.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I tiny_audio_amd/csrc -o scripts/probe/dec_probe scripts/probe/dec_probe.hip
#include "../../tiny_audio_amd/csrc/decode_fused.hip"
#include <cstdio>
#include <vector>

template <typename F> float time_us(F&& launch, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 28; ++i) launch(i);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch(i);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main() {
  const int B = 32, D = 1024, F = 3072, NQKV = 4096, BQ = 2048, NL = 28, Hq = 16, Hkv = 8, Lmax = 256, slot = 184;
  bf16_t *wqkv, *wo, *wgu, *wd, *xb, *kc, *vc, *qkv0, *ao;
  float *x, *x1, *lnw, *qn, *kn, *cosT, *sinT;
  int *pos, *slotp, *kmask;
  hipMalloc(&wqkv, (size_t)NL * NQKV * D * 2); hipMalloc(&wo, (size_t)NL * D * BQ * 2); hipMalloc(&wgu, (size_t)NL * 2 * F * D * 2);
  hipMalloc(&wd, (size_t)NL * D * F * 2); hipMalloc(&xb, (size_t)32 * F * 2); hipMalloc(&x, (size_t)32 * D * 4); hipMalloc(&x1, (size_t)32 * D * 4);
  hipMalloc(&lnw, D * 4); hipMalloc(&qn, 512); hipMalloc(&kn, 512); hipMalloc(&cosT, 4096 * 64 * 4); hipMalloc(&sinT, 4096 * 64 * 4);
  hipMalloc(&kc, (size_t)NL * B * Hkv * Lmax * 128 * 2); hipMalloc(&vc, (size_t)NL * B * Hkv * Lmax * 128 * 2);
  hipMalloc(&qkv0, (size_t)B * NQKV * 2); hipMalloc(&ao, (size_t)B * BQ * 2);
  hipMalloc(&pos, B * 4); hipMalloc(&slotp, 4); hipMalloc(&kmask, (size_t)B * Lmax * 4);
  hipMemset(wqkv, 0x11, (size_t)NL * NQKV * D * 2); hipMemset(wo, 0x11, (size_t)NL * D * BQ * 2); hipMemset(wgu, 0x11, (size_t)NL * 2 * F * D * 2);
  hipMemset(wd, 0x11, (size_t)NL * D * F * 2); hipMemset(xb, 0x11, (size_t)B * F * 2); hipMemset(x, 0x11, (size_t)B * D * 4); hipMemset(x1, 0, (size_t)B * D * 4);
  hipMemset(lnw, 0x11, D * 4); hipMemset(qn, 0x11, 512); hipMemset(kn, 0x11, 512); hipMemset(cosT, 0, 4096 * 64 * 4); hipMemset(sinT, 0, 4096 * 64 * 4);
  hipMemset(kc, 0x11, (size_t)NL * B * Hkv * Lmax * 128 * 2); hipMemset(vc, 0x11, (size_t)NL * B * Hkv * Lmax * 128 * 2);
  hipMemset(qkv0, 0x11, (size_t)B * NQKV * 2); hipMemset(pos, 0, B * 4);
  std::vector<int> km((size_t)B * Lmax, 1); hipMemcpy(kmask, km.data(), km.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(slotp, &slot, 4, hipMemcpyHostToDevice);
  const int R = 280;
#define LIN(NORM_, EPI_, UN_, COLS_, BLK_, DBG_, X_, W_, WSTRIDE_, OUT_, RES_, N_, K_)                                                  \
  time_us([&](int i) { hipLaunchKernelGGL((dec_linear_kernel<NORM_, EPI_, UN_, COLS_, BLK_, DBG_>), dim3((N_) / (COLS_)), dim3(512), 0, 0, \
                                          (const void*)(X_), lnw, 1e-6f, (W_) + (size_t)(i % NL) * (WSTRIDE_), (void*)(OUT_), (const float*)(RES_), B, N_, K_, (N_) / (COLS_), DecPf{nullptr, nullptr, 0, nullptr, 0, 0}); }, R)
#define ROW(name, NORM_, EPI_, UN_, COLS_, BLK_, X_, W_, WS_, OUT_, RES_, N_, K_)                                                        \
  printf("%-26s cols %2d %s  %6.2f %6.2f %6.2f %6.2f\n", name, COLS_, BLK_ ? "blocked  " : "row-major", LIN(NORM_, EPI_, UN_, COLS_, BLK_, 0, X_, W_, WS_, OUT_, RES_, N_, K_), \
         LIN(NORM_, EPI_, UN_, COLS_, BLK_, 1, X_, W_, WS_, OUT_, RES_, N_, K_), LIN(NORM_, EPI_, UN_, COLS_, BLK_, 2, X_, W_, WS_, OUT_, RES_, N_, K_),                     \
         LIN(NORM_, EPI_, UN_, COLS_, BLK_, 8, X_, W_, WS_, OUT_, RES_, N_, K_))
  printf("kernel                                           full   noW    noX   empty   (us per launch, eager back-to-back)\n");
  ROW("norm + q|k|v   (8.4 MB)", true, EPI_BF16, 4, 16, false, x, wqkv, (size_t)NQKV * D, qkv0, nullptr, NQKV, D);
  ROW("norm + q|k|v   (8.4 MB)", true, EPI_BF16, 4, 16, true, x, wqkv, (size_t)NQKV * D, qkv0, nullptr, NQKV, D);
  ROW("norm + q|k|v   (8.4 MB)", true, EPI_BF16, 4, 32, true, x, wqkv, (size_t)NQKV * D, qkv0, nullptr, NQKV, D);
  ROW("norm + gate|up (12.6 MB)", true, EPI_SWIGLU, 4, 8, false, x, wgu, (size_t)2 * F * D, xb, nullptr, F, D);
  ROW("norm + gate|up (12.6 MB)", true, EPI_SWIGLU, 4, 8, true, x, wgu, (size_t)2 * F * D, xb, nullptr, F, D);
  ROW("norm + gate|up (12.6 MB)", true, EPI_SWIGLU, 4, 16, true, x, wgu, (size_t)2 * F * D, xb, nullptr, F, D);
  ROW("o_proj + res   (4.2 MB)", false, EPI_F32_RES, 4, 4, false, xb, wo, (size_t)D * BQ, x1, x, D, BQ);
  ROW("o_proj + res   (4.2 MB)", false, EPI_F32_RES, 4, 4, true, xb, wo, (size_t)D * BQ, x1, x, D, BQ);
  ROW("o_proj + res   (4.2 MB)", false, EPI_F32_RES, 4, 8, true, xb, wo, (size_t)D * BQ, x1, x, D, BQ);
  ROW("o_proj + res   (4.2 MB)", false, EPI_F32_RES, 4, 16, true, xb, wo, (size_t)D * BQ, x1, x, D, BQ);
  ROW("down + res     (6.3 MB)", false, EPI_F32_RES, 6, 4, false, xb, wd, (size_t)D * F, x1, x, D, F);
  ROW("down + res     (6.3 MB)", false, EPI_F32_RES, 6, 4, true, xb, wd, (size_t)D * F, x1, x, D, F);
  ROW("down + res     (6.3 MB)", false, EPI_F32_RES, 6, 8, true, xb, wd, (size_t)D * F, x1, x, D, F);
  ROW("down + res     (6.3 MB)", false, EPI_F32_RES, 6, 16, true, xb, wd, (size_t)D * F, x1, x, D, F);
  const size_t smem = dec_attn_smem(2, Lmax), le = (size_t)B * Hkv * Lmax * 128;
  const float ta = time_us([&](int i) { hipLaunchKernelGGL((dec_attn_kernel<2>), dim3(Hkv, B), dim3(256), smem, 0, qkv0, qn, kn, cosT, sinT, pos, slotp, kmask,
                                                           kc + (size_t)(i % NL) * le, vc + (size_t)(i % NL) * le, ao, Hq, Hkv, Lmax, Lmax, 1e-6f, 0.088f, 1, B, DecPf{nullptr, nullptr, 0, nullptr, 0, 0}); }, R);
  printf("attention, 185 keys (24 MB)  %6.2f\n", ta);
  // the five launches of a layer in sequence over 28 layers (a decode step without its head), with / without next-kernel prefetch
  for (int wgs : {0, 64, 128, 256}) {
    auto layer = [&](int l) {
      const int ln = (l + 1) % NL;
      const ta_i_dec_prefetch p_kv = {kc + (size_t)l * le, vc + (size_t)l * le, 0, slotp, (long)Lmax * 256, B * Hkv, wgs};
      const ta_i_dec_prefetch p_o = {wo + (size_t)l * D * BQ, nullptr, (long)D * BQ * 2, nullptr, 0, 0, wgs};
      const ta_i_dec_prefetch p_gu = {wgu + (size_t)l * 2 * F * D, nullptr, (long)2 * F * D * 2, nullptr, 0, 0, wgs};
      const ta_i_dec_prefetch p_d = {wd + (size_t)l * D * F, nullptr, (long)D * F * 2, nullptr, 0, 0, wgs};
      const ta_i_dec_prefetch p_n = {wqkv + (size_t)ln * NQKV * D, nullptr, (long)NQKV * D * 2, nullptr, 0, 0, wgs};
      const bool on = wgs > 0;
      ta_i_dec_norm_linear(x, lnw, 1e-6f, wqkv + (size_t)l * NQKV * D, qkv0, B, NQKV, D, false, on ? &p_kv : nullptr, 0);
      ta_i_dec_attn(qkv0, qn, kn, cosT, sinT, pos, slotp, kmask, kc + (size_t)l * le, vc + (size_t)l * le, ao, B, Hq, Hkv, Lmax, 1e-6f, 0.088f, on ? &p_o : nullptr, 0);
      ta_i_dec_linear_res(ao, wo + (size_t)l * D * BQ, x1, x, B, D, BQ, on ? &p_gu : nullptr, 0);
      ta_i_dec_norm_linear(x1, lnw, 1e-6f, wgu + (size_t)l * 2 * F * D, xb, B, F, D, true, on ? &p_d : nullptr, 0);
      ta_i_dec_linear_res(xb, wd + (size_t)l * D * F, x, x1, B, D, F, on ? &p_n : nullptr, 0);
    };
    const float tl = time_us([&](int i) { layer(i % NL); }, 280);
    printf("layer sequence (5 launches, eager), prefetch workgroups %3d: %6.2f us per layer\n", wgs, tl);
  }
  return 0;
}
