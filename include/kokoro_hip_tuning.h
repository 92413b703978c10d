/* Tuning hooks of libkokoro_hip.so for tools/ (tile thresholds, XCD-aware tile order, pipeline depth, k-slices of grouped
 * launches).  NOT part of the product ABI: they mutate process-global tile policy, so the product library does not export
 * them.  Build a library that does with `python -m kokoro_ruslan_amd.build --tuning` (defines KK_TUNING_HOOKS).  The product
 * library reads an optional one-time override from the environment instead (KK_GEMM16_TUNE, see csrc/kk_gemm16.hip). */
#ifndef KOKORO_HIP_TUNING_H
#define KOKORO_HIP_TUNING_H
#ifdef __cplusplus
extern "C" {
#endif
int kk_gemm_tune(int tm_threshold, int xcd_swizzle);
int kk_gemm_tune16(int enable, int thr128, int thr12864, int split_target);
int kk_gemm_tune_group(int split);
int kk_gemm_tune16x(int on, int force, int dbg);   /* large-tile family (csrc/kk_gemm16x.hip): enable bits, forced tile, probe bits */
#ifdef __cplusplus
}
#endif
#endif
