// Developer tool: do the matrix pipe and the vector ALU of a CDNA4 SIMD overlap — inside one wave, and across the waves of
// a SIMD? Times (wall, HIP events) loops of (a) MFMAs only, (b) vector-ALU instructions only, (c) both in one wave,
// independent of each other, (d) MFMA-only waves beside VALU-only waves on the same SIMDs.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_valu_overlap.hip -o tmp_ab/mfma_valu_overlap && tmp_ab/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define ITERS 2000
#define VPM 4     // dependent chains of vector instructions (5 per chain and iteration: 20 beside 2 MFMAs)

template <int MODE>   // 0: mfma only, 1: valu only, 2: both in every wave, 3: even waves mfma / odd waves valu
__global__ __launch_bounds__(256) void k(float* out, uint32_t seed) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t r[VPM];
  for (int i = 0; i < VPM; ++i) r[i] = seed * (lane + 3) + i;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(i + 1); }
  f32x16 acc0 = {0}, acc1 = {0};
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (wave & 1) == 0);
  const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  for (int it = 0; it < ITERS; ++it) {
    if (do_m) {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
    }
    if (do_v) {
#pragma unroll
        for (int i = 0; i < VPM; ++i) {   // and + sub-like + perm mix, one dependent chain per register, VPM chains
          const float f = __uint_as_float(r[i] & 0xFFFF0000u | 0x3F000000u);
          const float g = __uint_as_float(r[i] | 0x3F800000u) - f;
          r[i] = __builtin_amdgcn_perm(__float_as_uint(g), r[i], 0x07060302u) + 1u;
        }
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  uint32_t x = 0;
  for (int i = 0; i < VPM; ++i) x ^= r[i];
  if (s == 123.456f || x == 0xdeadbeefu) out[threadIdx.x] = s;
}

template <int MODE>
float run(int blocks, float* d) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, 12345u);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* d;
  hipMalloc(&d, 4096);
  for (int wps = 1; wps <= 4; wps *= 2) {      // waves per SIMD (workgroups of 4 waves: one per SIMD each)
    const int blocks = 256 * wps;
    const float m = run<0>(blocks, d), v = run<1>(blocks, d), both = run<2>(blocks, d), split = run<3>(blocks, d);
    printf("waves/SIMD %d: mfma-only %.1f us (%d MFMA per wave: %.1f cycles each at 2.4 GHz per SIMD-wave), valu-only %.1f us "
           "(%d chain steps of 5 instructions), both in one wave %.1f us (sum %.1f, max %.1f), mfma waves beside valu waves %.1f us\n",
           wps, m, 2 * ITERS, m * 2400.0f / (2 * ITERS) / wps, v, VPM * ITERS, both, m + v, m > v ? m : v, split);
  }
  return 0;
}
