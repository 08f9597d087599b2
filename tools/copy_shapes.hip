// copy_shapes -- how fast can kernels shaped like the partition levels move 8-byte items on this GPU?
//   1 float4 grid-stride copy (the guide's yardstick)
//   2 tiles of 12288 items, one 1024-thread workgroup per CU walking them: 12 8-byte loads per thread into
//     registers, then 12 stores, contiguous
//   3 as 2, the loads of tile t+1 issued before the stores of tile t (the partition kernels' software pipeline)
//   4 as 3, but the tile leaves as 64 runs of 192 items, run r to region r (level 2's write pattern)
//   5 as 4 with 128 runs of 96 items (level 1's)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/copy_shapes tools/copy_shapes.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define TILE 12288u
#define PER 12

__global__ __launch_bounds__(256) void k_copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

typedef float qh_f4 __attribute__((ext_vector_type(4)));
// NT bit 0: nontemporal loads, bit 1: nontemporal stores; four independent 16-byte pieces per thread and step
template <int NT>
__global__ __launch_bounds__(256) void k_copy4u(const qh_f4* __restrict__ a, qh_f4* __restrict__ b, size_t n4) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n4; i += 4 * stride) {
    qh_f4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = (NT & 1) ? __builtin_nontemporal_load(&a[i + j * stride]) : a[i + j * stride];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (NT & 2) __builtin_nontemporal_store(v[j], &b[i + j * stride]);
      else b[i + j * stride] = v[j];
    }
  }
  for (; i < n4; i += stride) b[i] = a[i];
}

template <int MODE, int RUNS>
__global__ __launch_bounds__(1024) void k_tiles(const uint64_t* __restrict__ a, uint64_t* __restrict__ b, size_t ntiles,
                                                size_t region) {
  const int tid = threadIdx.x;
  uint64_t cur[PER], nxt[PER];
  size_t t = blockIdx.x;
  if (MODE >= 3 && t < ntiles) {
#pragma unroll
    for (int i = 0; i < PER; ++i) nxt[i] = a[t * TILE + (size_t)i * 1024 + tid];
  }
  for (; t < ntiles; t += gridDim.x) {
    if (MODE >= 3) {
#pragma unroll
      for (int i = 0; i < PER; ++i) cur[i] = nxt[i];
      const size_t tn = t + gridDim.x;
      if (tn < ntiles) {
#pragma unroll
        for (int i = 0; i < PER; ++i) nxt[i] = a[tn * TILE + (size_t)i * 1024 + tid];
      }
    } else {
#pragma unroll
      for (int i = 0; i < PER; ++i) cur[i] = a[t * TILE + (size_t)i * 1024 + tid];
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const uint32_t j = (uint32_t)i * 1024u + tid;  // position inside the tile
      size_t dst;
      if (RUNS == 0) dst = t * TILE + j;
      else {
        const uint32_t runlen = TILE / RUNS, r = j / runlen, o = j - r * runlen;
        dst = (size_t)r * region + t * runlen + o;
      }
      b[dst] = cur[i];
    }
  }
}

int main(int argc, char** argv) {
  const size_t n = (argc > 1 ? strtoull(argv[1], 0, 0) : 152158273ull) / TILE * TILE;
  const size_t ntiles = n / TILE;
  uint64_t *a, *b;
  CK(hipMalloc(&a, n * 8));
  CK(hipMalloc(&b, n * 8 + (1 << 20)));
  CK(hipMemset(a, 1, n * 8));
  CK(hipMemset(b, 0, n * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < 10; ++i) launch();
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 10;
    printf("%-58s %.3f ms  %.2f TB/s read+write\n", name, ms, 2.0 * n * 8 / (ms * 1e-3) / 1e12);
  };
  run("1 float4 grid-stride, 2048 x 256", [&]() { hipLaunchKernelGGL(k_copy4, dim3(2048), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 2); });
  run("1 float4 grid-stride, 8192 x 256", [&]() { hipLaunchKernelGGL(k_copy4, dim3(8192), dim3(256), 0, 0, (const float4*)a, (float4*)b, n / 2); });
  run("1 four float4 in flight, 2048 x 256", [&]() { hipLaunchKernelGGL((k_copy4u<0>), dim3(2048), dim3(256), 0, 0, (const qh_f4*)a, (qh_f4*)b, n / 2); });
  run("1 ... nontemporal loads", [&]() { hipLaunchKernelGGL((k_copy4u<1>), dim3(2048), dim3(256), 0, 0, (const qh_f4*)a, (qh_f4*)b, n / 2); });
  run("1 ... nontemporal stores", [&]() { hipLaunchKernelGGL((k_copy4u<2>), dim3(2048), dim3(256), 0, 0, (const qh_f4*)a, (qh_f4*)b, n / 2); });
  run("1 ... nontemporal loads and stores", [&]() { hipLaunchKernelGGL((k_copy4u<3>), dim3(2048), dim3(256), 0, 0, (const qh_f4*)a, (qh_f4*)b, n / 2); });
  run("1 ... nontemporal both, 1024 x 256", [&]() { hipLaunchKernelGGL((k_copy4u<3>), dim3(1024), dim3(256), 0, 0, (const qh_f4*)a, (qh_f4*)b, n / 2); });
  run("2 tiles, 256 x 1024, load then store", [&]() { hipLaunchKernelGGL((k_tiles<2, 0>), dim3(256), dim3(1024), 0, 0, a, b, ntiles, 0); });
  run("3 tiles, 256 x 1024, next tile's loads first", [&]() { hipLaunchKernelGGL((k_tiles<3, 0>), dim3(256), dim3(1024), 0, 0, a, b, ntiles, 0); });
  run("3 tiles, 512 x 1024 (two workgroups per CU)", [&]() { hipLaunchKernelGGL((k_tiles<3, 0>), dim3(512), dim3(1024), 0, 0, a, b, ntiles, 0); });
  run("4 as 3, 64 runs of 192 items to 64 regions", [&]() { hipLaunchKernelGGL((k_tiles<3, 64>), dim3(256), dim3(1024), 0, 0, a, b, ntiles, n / 64); });
  run("5 as 3, 128 runs of 96 items to 128 regions", [&]() { hipLaunchKernelGGL((k_tiles<3, 128>), dim3(256), dim3(1024), 0, 0, a, b, ntiles, n / 128); });
  run("5 as 3, 512 runs of 24 items to 512 regions", [&]() { hipLaunchKernelGGL((k_tiles<3, 512>), dim3(256), dim3(1024), 0, 0, a, b, ntiles, n / 512); });
  return 0;
}
