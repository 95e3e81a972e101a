// Probe: marginal cost of a transcendental inside a mixed VALU stream on gfx950 (2 waves per SIMD, 256 workgroups):
// per loop iteration 8 x (v_exp_f32 + n x v_fma_f32), all on independent registers.
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/trans_mix tools/probes/trans_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int N, int WITH_T>
__global__ void probe(int iters, float* out, const float* seed) {
  float e[8], f[8];
  for (int i = 0; i < 8; ++i) { e[i] = seed[threadIdx.x & 63] + i; f[i] = e[i] * 0.5f; }
  const float c = 1e-9f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (WITH_T) asm volatile("v_exp_f32 %0, %0" : "+v"(e[k]));
#pragma unroll
      for (int n = 0; n < N; ++n) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[(k + n) & 7]) : "v"(c));
    }
  }
  float s = 0;
  for (int i = 0; i < 8; ++i) s += e[i] + f[i];
  if (s == 12345.f) out[threadIdx.x] = s;
}

template <int N, int WITH_T>
static float run(int threads, float* out, const float* seed) {
  const int iters = 40000;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((probe<N, WITH_T>), dim3(256), dim3(threads), 0, 0, iters, out, seed);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<N, WITH_T>), dim3(256), dim3(threads), 0, 0, iters, out, seed);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5 * 1e6 / (8.0 * iters * (threads / 256));   // ns per (exp + N fma) group per SIMD
}

template <int N>
static void report(float* out, const float* seed) {
  for (int threads : {256, 512, 1024}) {
    const float with_t = run<N, 1>(threads, out, seed), without = N ? run<N, 0>(threads, out, seed) : 0.f;
    printf("1 v_exp + %d v_fma, %d wave(s)/SIMD: %6.2f ns per group per SIMD; the %d v_fma alone %6.2f; marginal v_exp %6.2f ns (= %4.1f cycles at 2.4 GHz)\n",
           N, threads / 256, with_t, N, without, with_t - without, (with_t - without) * 2.4);
  }
}

int main() {
  float *out, *seed;
  hipMalloc(&out, 8192);
  hipMalloc(&seed, 256);
  float hs[64];
  for (int i = 0; i < 64; ++i) hs[i] = 0.37f + 0.013f * i;
  hipMemcpy(seed, hs, 256, hipMemcpyHostToDevice);
  report<0>(out, seed);
  report<1>(out, seed);
  report<2>(out, seed);
  report<3>(out, seed);
  report<4>(out, seed);
  report<6>(out, seed);
  return 0;
}
