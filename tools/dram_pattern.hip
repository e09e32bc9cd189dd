// Developer tool: what HBM gives a read pattern — every workgroup reads, for each of `rows` rows, `piece` contiguous bytes at
// row * stride + piece * workgroup (the critic pass's pattern: piece = 256, stride = environments per frame slab), against the
// same bytes read as one contiguous run per workgroup.   hipcc --offload-arch=gfx950 -O3 tools/dram_pattern.hip -o /tmp/dram_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// 256 threads: lane l of wave w reads dword l of row (4 r + w)'s piece (piece = 256 B) — 32 rows in flight per workgroup and step
template <bool STRIDED>
__global__ __launch_bounds__(256) void k_read(const uint32_t* __restrict__ src, int64_t stride_dw, int64_t rows, uint32_t* __restrict__ out) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t wg = blockIdx.x;
  src += (int64_t)blockIdx.y * rows * stride_dw;      // this workgroup's frame slab
  uint32_t acc = 0;
  for (int64_t r0 = 0; r0 < rows; r0 += 32) {
    uint32_t v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int64_t r = r0 + 4 * i + wave;
      const int64_t off = STRIDED ? r * stride_dw + 64 * wg + lane : (wg * rows + r) * 64 + lane;
      v[i] = src[off];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc += v[i];
  }
  if (acc == 0x12345678u) out[(wg * 256 + tid) & 0xFFFF] = acc;
}

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 32768;      // bytes per row (= environments)
  const int64_t rows = argc > 2 ? atoll(argv[2]) : 2496 * 64;  // rows in total (multiple of 32): 64 frames of 2 496 nodes
  const int64_t bytes = B * rows;
  uint32_t *src, *out;
  CHECK(hipMalloc(&src, bytes));
  CHECK(hipMalloc(&out, 1 << 20));
  CHECK(hipMemset(src, 1, bytes));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int64_t wgs = B / 256;
  // the critic's shape: a workgroup owns 256 bytes of EVERY row of one frame (2 496 rows); frames = rows / 2496 slabs
  const int64_t per_frame = 2496, frames = rows / per_frame;
  for (int strided = 1; strided >= 0; --strided) {
    for (int rep = 0; rep < 3; ++rep) {
      CHECK(hipEventRecord(e0));
      if (strided)
        hipLaunchKernelGGL(k_read<true>, dim3((unsigned)wgs, (unsigned)frames), dim3(256), 0, 0, src, B / 4, per_frame, out);
      else
        hipLaunchKernelGGL(k_read<false>, dim3((unsigned)wgs, (unsigned)frames), dim3(256), 0, 0, src, B / 4, per_frame, out);
      CHECK(hipEventRecord(e1));
      CHECK(hipEventSynchronize(e1));
      float ms = 0;
      CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("%s  %.2f GB in %.3f ms = %.2f TB/s\n", strided ? "256-byte pieces, rows B bytes apart " : "one contiguous run per workgroup     ",
             bytes / 1e9, ms, bytes / 1e9 / ms);
    }
  }
  return 0;
}
