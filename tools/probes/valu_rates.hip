// Probe: issue cost of VALU / transcendental instructions on gfx950, one and two waves per SIMD (256 workgroups).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/valu_rates tools/probes/valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>

#define LOOP8(INS)                                                                                       \
  for (int i = 0; i < iters; ++i)                                                                        \
    asm volatile(INS " %0, %0\n" INS " %1, %1\n" INS " %2, %2\n" INS " %3, %3\n" INS " %4, %4\n" INS " %5, %5\n" INS " %6, %6\n" INS " %7, %7\n" \
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
#define LOOP8B(INS)                                                                                      \
  for (int i = 0; i < iters; ++i)                                                                        \
    asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
                 : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(c));

template <int K>
__global__ void probe(int iters, float* out, const float* seed) {
  float v0 = seed[threadIdx.x & 63], v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
  const float c = 1.0000001f;
  if (K == 0) { LOOP8("v_exp_f32") }
  if (K == 1) { LOOP8("v_rcp_f32") }
  if (K == 2) { LOOP8("v_exp_f16") }
  if (K == 3) { LOOP8("v_rcp_f16") }
  if (K == 4) { LOOP8B("v_mul_f32") }
  if (K == 5) { LOOP8B("v_pk_mul_f16") }
  if (K == 6) { LOOP8B("v_pk_add_f16") }
  if (K == 7) { LOOP8B("v_cvt_pk_f16_f32") }
  if (K == 8) { LOOP8("v_mov_b32") }
  if (K == 9) { LOOP8("v_rsq_f32") }
  if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 12345.f) out[threadIdx.x] = v0;
}

template <int K>
static void report(const char* name, float* out, const float* seed) {
  const int iters = 50000;
  for (int threads : {256, 512}) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(probe<K>, dim3(256), dim3(threads), 0, 0, iters, out, seed);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<K>, dim3(256), dim3(threads), 0, 0, iters, out, seed);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    ms /= 5;
    const double per_simd = 8.0 * iters * (threads / 256);   // wave-instructions per SIMD
    printf("%-18s %d wave(s)/SIMD  %7.3f ms  %6.2f ns per wave-instruction per SIMD (= %5.1f cycles at 2.4 GHz)\n", name,
           threads / 256, ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
  }
}

int main() {
  float *out, *seed;
  hipMalloc(&out, 4096);
  hipMalloc(&seed, 256);
  float hs[64];
  for (int i = 0; i < 64; ++i) hs[i] = 0.37f + 0.013f * i;
  hipMemcpy(seed, hs, 256, hipMemcpyHostToDevice);
  report<0>("v_exp_f32", out, seed);
  report<1>("v_rcp_f32", out, seed);
  report<9>("v_rsq_f32", out, seed);
  report<2>("v_exp_f16", out, seed);
  report<3>("v_rcp_f16", out, seed);
  report<4>("v_mul_f32", out, seed);
  report<5>("v_pk_mul_f16", out, seed);
  report<6>("v_pk_add_f16", out, seed);
  report<7>("v_cvt_pk_f16_f32", out, seed);
  report<8>("v_mov_b32", out, seed);
  return 0;
}
