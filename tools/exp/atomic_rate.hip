// micro-benchmark (round 6): what a same-address device-scope atomicAdd costs on MI355X when 4 096 one-wave workgroups issue it
// from lane 0, one per "item" -- the hand-out counter and the queue append of the persistent replay kernels.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/atomic_rate tools/exp/atomic_rate.hip && tools/exp/atomic_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k_one(uint32_t* ctr, uint32_t* q, int per_wave, int stride_words) {
  uint32_t* c = ctr + (size_t)(blockIdx.x % (stride_words ? 64 : 1)) * stride_words;
  for (int i = 0; i < per_wave; ++i) {
    if (threadIdx.x == 0) q[atomicAdd(c, 1u) & 0xfffffu] = blockIdx.x;
    __syncthreads();
  }
}
__global__ void k_noret(uint32_t* ctr, int per_wave) {
  for (int i = 0; i < per_wave; ++i) {
    if (threadIdx.x == 0) atomicAdd(ctr, 1u);
    __syncthreads();
  }
}
__global__ void k_batched(uint32_t* ctr, uint32_t* q, int per_wave, int batch) {
  for (int i = 0; i < per_wave; i += batch) {
    if (threadIdx.x == 0) {
      const uint32_t b = atomicAdd(ctr, (uint32_t)batch);
      for (int k = 0; k < batch; ++k) q[(b + k) & 0xfffffu] = blockIdx.x;
    }
    __syncthreads();
  }
}
int main() {
  uint32_t *ctr, *q;
  hipMalloc(&ctr, 1 << 20);
  hipMalloc(&q, 4 << 20);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int waves = 4096, per = 32;
  auto timeit = [&](const char* name, auto launch) {
    hipMemset(ctr, 0, 1 << 20);
    launch();
    hipDeviceSynchronize();
    hipMemset(ctr, 0, 1 << 20);
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %8.3f ms  = %6.1f ns per item (%d items)\n", name, ms, 1e6 * ms / (waves * per), waves * per);
  };
  timeit("one counter, returning, 1 per item", [&] { hipLaunchKernelGGL(k_one, dim3(waves), dim3(64), 0, 0, ctr, q, per, 0); });
  timeit("64 counters 128 B apart, returning", [&] { hipLaunchKernelGGL(k_one, dim3(waves), dim3(64), 0, 0, ctr, q, per, 32); });
  timeit("64 counters 4 KB apart, returning", [&] { hipLaunchKernelGGL(k_one, dim3(waves), dim3(64), 0, 0, ctr, q, per, 1024); });
  timeit("one counter, no return", [&] { hipLaunchKernelGGL(k_noret, dim3(waves), dim3(64), 0, 0, ctr, per); });
  timeit("one counter, returning, 1 per 4 items", [&] { hipLaunchKernelGGL(k_batched, dim3(waves), dim3(64), 0, 0, ctr, q, per, 4); });
  timeit("one counter, returning, 1 per 16 items", [&] { hipLaunchKernelGGL(k_batched, dim3(waves), dim3(64), 0, 0, ctr, q, per, 16); });
  timeit("one counter, returning, 1 per 32 items", [&] { hipLaunchKernelGGL(k_batched, dim3(waves), dim3(64), 0, 0, ctr, q, per, 32); });
  return 0;
}
