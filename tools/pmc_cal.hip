// Developer tool (round 3): known-byte-count kernels in the ACCESS SHAPES of the frame kernels, to calibrate what rocprofv3's
// FETCH_SIZE / WRITE_SIZE (and the TCC_EA0_RDREQ / WRREQ request counters beside them) report per byte actually moved.
// The guide's "FETCH_SIZE reads half of a wide coalesced read" is calibrated on 16 B/lane copies only; the frame kernels
// issue 1-, 4- and 8-byte-per-lane coalesced loads (env-minor words: lane = environment) and scattered 12-byte slot
// triples at a 192-byte stride.
//
//   hipcc --offload-arch=gfx950 -O3 tools/pmc_cal.hip -o gpurun_out/pmc_cal
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/cal_f -o run -- gpurun_out/pmc_cal
//   (one pass per counter set: WRITE_SIZE; TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum; TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum)
//   python3 tools/pmc_cal_reduce.py gpurun_out/cal_* > profiles/r03_pmc_calibration.txt
//
// Every kernel touches each byte of its range exactly once (grid-stride over a 1 GiB buffer: four times the Infinity
// Cache), the program prints "name useful_read_bytes useful_write_bytes" per kernel for the reducer.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                         \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));       \
      exit(1);                                                                        \
    }                                                                                 \
  } while (0)

template <typename T>
__device__ __forceinline__ uint32_t fold(T v);
template <>
__device__ __forceinline__ uint32_t fold<uint8_t>(uint8_t v) { return (uint32_t)v * 0x01010101u; }
template <>
__device__ __forceinline__ uint32_t fold<uint32_t>(uint32_t v) { return v; }
template <>
__device__ __forceinline__ uint32_t fold<uint2>(uint2 v) { return v.x ^ v.y; }
template <>
__device__ __forceinline__ uint32_t fold<uint4>(uint4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// coalesced read of n elements of T, lane = consecutive element
template <typename T>
__global__ __launch_bounds__(256) void cal_read(const T* __restrict__ p, int64_t n, uint32_t* __restrict__ sink) {
  uint32_t acc = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc ^= fold<T>(p[i]);
  if (acc == sink[1]) sink[0] = acc;   // sink[1] is a runtime value the fold never equals in practice: keeps the loads alive
}

template <typename T>
__device__ __forceinline__ T mk(uint32_t v);
template <>
__device__ __forceinline__ uint8_t mk<uint8_t>(uint32_t v) { return (uint8_t)v; }
template <>
__device__ __forceinline__ uint32_t mk<uint32_t>(uint32_t v) { return v; }
template <>
__device__ __forceinline__ uint2 mk<uint2>(uint32_t v) { return make_uint2(v, v); }
template <>
__device__ __forceinline__ uint4 mk<uint4>(uint32_t v) { return make_uint4(v, v, v, v); }

template <typename T>
__global__ __launch_bounds__(256) void cal_write(T* __restrict__ p, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) p[i] = mk<T>((uint32_t)i);
}

// the slot store's shape: rows of `stride` floats, one lane reads / writes ONE 12-byte triple at float offset `off` of its row
__global__ __launch_bounds__(256) void cal_read_triple(const float* __restrict__ p, int64_t rows, int stride, int off,
                                                      uint32_t* __restrict__ sink) {
  float acc = 0.0f;
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    const float* q = p + r * stride + off;
    acc += q[0] + q[1] + q[2];
  }
  if (acc == 1.2345e30f) sink[0] = 1;
}
__global__ __launch_bounds__(256) void cal_write_triple(float* __restrict__ p, int64_t rows, int stride, int off) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    float* q = p + r * stride + off;
    q[0] = (float)r;
    q[1] = 1.0f;
    q[2] = 2.0f;
  }
}
// one aligned record of NQ * 16 bytes per lane at a `stride`-byte stride: does a scattered store that covers a whole aligned
// 32- / 64-byte sector avoid the read-modify-write of a partial one?
template <int NQ>
__global__ __launch_bounds__(256) void cal_write_rec(uint4* __restrict__ p, int64_t rows, int stride16) {
  for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < rows; r += (int64_t)gridDim.x * 256) {
    uint4* q = p + r * stride16;
#pragma unroll
    for (int k = 0; k < NQ; ++k) q[k] = make_uint4((uint32_t)r, k, 2u, 3u);
  }
}
// sparse dword stores into a dense array: every `every`-th element (the event rows' refreshed words among idle rows)
__global__ __launch_bounds__(256) void cal_write_sparse4(uint32_t* __restrict__ p, int64_t n, int every) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    if ((i * 2654435761u >> 7) % (unsigned)every == 0u) p[i] = (uint32_t)i;
}
__global__ __launch_bounds__(256) void cal_write_sparse8(uint2* __restrict__ p, int64_t n, int every) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    if ((i * 2654435761u >> 7) % (unsigned)every == 0u) p[i] = make_uint2((uint32_t)i, 0u);
}
__global__ __launch_bounds__(256) void cal_count_sparse(int64_t n, int every, unsigned long long* __restrict__ cnt) {
  unsigned long long c = 0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    if ((i * 2654435761u >> 7) % (unsigned)every == 0u) ++c;
  atomicAdd(cnt, c);
}
// gather like the frame kernels' neighbour reads: lane = environment (coalesced), row picked per WAVE by a hash — 8-byte
// words of a [rows][B] array, each (row, env) read exactly once per pass because the hash is a permutation of the rows
__global__ __launch_bounds__(256) void cal_gather8(const uint2* __restrict__ p, uint32_t rows, uint32_t B, uint32_t mul,
                                                  uint32_t* __restrict__ sink) {
  const uint32_t b = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  if (b < B)
    for (uint32_t r = blockIdx.y; r < rows; r += gridDim.y) {
      const uint32_t rr = (uint32_t)(((uint64_t)r * mul) % rows);   // mul coprime with rows: a permutation
      acc ^= fold<uint2>(p[(size_t)rr * B + b]);
    }
  if (acc == sink[1]) sink[0] = acc;
}

int main() {
  const size_t BYTES = (size_t)1 << 30;
  void* buf;
  uint32_t* sink;
  unsigned long long* cnt;
  CK(hipMalloc(&buf, BYTES));
  CK(hipMalloc(&sink, 64));
  CK(hipMalloc(&cnt, 8));
  CK(hipMemset(buf, 1, BYTES));
  CK(hipMemset(sink, 0xA5, 64));
  CK(hipDeviceSynchronize());
  const dim3 g(256 * 16), t(256);
  const size_t HALF = BYTES / 2;   // reads and writes below use 512 MiB (two Infinity Caches) unless stated
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(cal_read<uint8_t>, g, t, 0, 0, (const uint8_t*)buf, (int64_t)HALF, sink);
    hipLaunchKernelGGL(cal_read<uint32_t>, g, t, 0, 0, (const uint32_t*)buf, (int64_t)(HALF / 4), sink);
    hipLaunchKernelGGL(cal_read<uint2>, g, t, 0, 0, (const uint2*)buf, (int64_t)(HALF / 8), sink);
    hipLaunchKernelGGL(cal_read<uint4>, g, t, 0, 0, (const uint4*)buf, (int64_t)(HALF / 16), sink);
    hipLaunchKernelGGL(cal_read_triple, g, t, 0, 0, (const float*)buf, (int64_t)(BYTES / 192), 48, 12, sink);
    hipLaunchKernelGGL(cal_gather8, dim3(64, 1024), t, 0, 0, (const uint2*)buf, 4093u, 16384u, 1021u, sink);
    hipLaunchKernelGGL(cal_write<uint8_t>, g, t, 0, 0, (uint8_t*)buf, (int64_t)HALF);
    hipLaunchKernelGGL(cal_write<uint32_t>, g, t, 0, 0, (uint32_t*)buf, (int64_t)(HALF / 4));
    hipLaunchKernelGGL(cal_write<uint2>, g, t, 0, 0, (uint2*)buf, (int64_t)(HALF / 8));
    hipLaunchKernelGGL(cal_write<uint4>, g, t, 0, 0, (uint4*)buf, (int64_t)(HALF / 16));
    hipLaunchKernelGGL(cal_write_triple, g, t, 0, 0, (float*)buf, (int64_t)(BYTES / 192), 48, 12);
    hipLaunchKernelGGL(cal_write_rec<1>, g, t, 0, 0, (uint4*)buf, (int64_t)(BYTES / 256), 16);
    hipLaunchKernelGGL(cal_write_rec<2>, g, t, 0, 0, (uint4*)buf, (int64_t)(BYTES / 256), 16);
    hipLaunchKernelGGL(cal_write_rec<4>, g, t, 0, 0, (uint4*)buf, (int64_t)(BYTES / 256), 16);
    hipLaunchKernelGGL(cal_write_rec<8>, g, t, 0, 0, (uint4*)buf, (int64_t)(BYTES / 256), 16);
    hipLaunchKernelGGL(cal_write_sparse4, g, t, 0, 0, (uint32_t*)buf, (int64_t)(HALF / 4), 10);
    hipLaunchKernelGGL(cal_write_sparse8, g, t, 0, 0, (uint2*)buf, (int64_t)(HALF / 8), 10);
  }
  CK(hipDeviceSynchronize());
  unsigned long long c4 = 0, c8 = 0;
  CK(hipMemset(cnt, 0, 8));
  hipLaunchKernelGGL(cal_count_sparse, g, t, 0, 0, (int64_t)(HALF / 4), 10, cnt);
  CK(hipMemcpy(&c4, cnt, 8, hipMemcpyDeviceToHost));
  CK(hipMemset(cnt, 0, 8));
  hipLaunchKernelGGL(cal_count_sparse, g, t, 0, 0, (int64_t)(HALF / 8), 10, cnt);
  CK(hipMemcpy(&c8, cnt, 8, hipMemcpyDeviceToHost));
  // name, useful bytes read, useful bytes written, note
  printf("CAL cal_read<unsigned_char> %zu 0 1B/lane coalesced load\n", HALF);
  printf("CAL cal_read<unsigned_int> %zu 0 4B/lane coalesced load\n", HALF);
  printf("CAL cal_read<uint2> %zu 0 8B/lane coalesced load\n", HALF);
  printf("CAL cal_read<uint4> %zu 0 16B/lane coalesced load\n", HALF);
  printf("CAL cal_read_triple %zu 0 scattered 12B triples at a 192B stride (sectors touched: %zu B at 32B, %zu B at 64B)\n",
         (BYTES / 192) * 12, (BYTES / 192) * 32, (BYTES / 192) * 64);
  printf("CAL cal_gather8 %zu 0 env-minor 8B gather (lane = env, row permuted per wave)\n", (size_t)4093 * 16384 * 8);
  printf("CAL cal_write<unsigned_char> 0 %zu 1B/lane coalesced store\n", HALF);
  printf("CAL cal_write<unsigned_int> 0 %zu 4B/lane coalesced store\n", HALF);
  printf("CAL cal_write<uint2> 0 %zu 8B/lane coalesced store\n", HALF);
  printf("CAL cal_write<uint4> 0 %zu 16B/lane coalesced store\n", HALF);
  printf("CAL cal_write_triple 0 %zu scattered 12B triples at a 192B stride\n", (BYTES / 192) * 12);
  printf("CAL cal_write_rec<1> 0 %zu one aligned 16B record per lane at a 256B stride\n", (BYTES / 256) * 16);
  printf("CAL cal_write_rec<2> 0 %zu one aligned 32B record per lane at a 256B stride\n", (BYTES / 256) * 32);
  printf("CAL cal_write_rec<4> 0 %zu one aligned 64B record per lane at a 256B stride\n", (BYTES / 256) * 64);
  printf("CAL cal_write_rec<8> 0 %zu one aligned 128B record per lane at a 256B stride\n", (BYTES / 256) * 128);
  printf("CAL cal_write_sparse4 0 %llu one dword in ten, hashed (sparse stores into a dense array)\n", c4 * 4ull);
  printf("CAL cal_write_sparse8 0 %llu one 8B word in ten, hashed\n", c8 * 8ull);
  return 0;
}
