// Probe: does the VALU / transcendental stream of one wave overlap with the 16-bit MFMA stream (v_mfma_f32_16x16x32_f16,
// v_mfma_f32_32x32x16_f16) of ANOTHER wave on the same SIMD?  (The fp32-MFMA question: mfma_valu_overlap.hip.)
// A workgroup of 8 waves per CU: waves 0-3 run an MFMA loop, waves 4-7 (their SIMD partners) a VALU loop.  Also: one wave
// interleaving both streams itself (1 MFMA : n VALU).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/bin/mfma16_valu_overlap tools/probes/mfma16_valu_overlap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef float f32x16v __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));

// MK: 0 = 16x16x32 f16, 1 = 32x32x16 f16.  VK: 0 = v_fma_f32, 1 = v_exp_f32, 2 = v_pk_mul_f32, 3 = v_cvt_pk_f16_f32 (as fma mix)
template <int MK, int VK>
__global__ __launch_bounds__(512) void probe(int do_mfma, int do_valu, int im, int iv, float* out, const float* seed) {
  const int wave = threadIdx.x >> 6;
  const float x = seed[threadIdx.x & 255];
  if (wave < 4) {
    if (!do_mfma) return;
    f16x8v h;
    for (int j = 0; j < 8; ++j) h[j] = (_Float16)(x * (j + 1));
    if (MK == 0) {
      f32x4v a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
      for (int i = 0; i < im; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a3, 0, 0, 0);
      }
      if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f) out[threadIdx.x] = a0[0];
    } else {
      f32x16v a0 = {0}, a1 = a0;
      for (int i = 0; i < im; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, a1, 0, 0, 0);
      }
      if (a0[0] + a1[1] == 12345.f) out[threadIdx.x] = a0[0];
    }
  } else {
    if (!do_valu) return;
    float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
    if (VK == 0) {
      for (int i = 0; i < iv; ++i)
        asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                     "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(1e-9f));
    } else if (VK == 1) {
      for (int i = 0; i < iv; ++i)
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
    }
    if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 12345.f) out[threadIdx.x] = v0;
  }
}

// one wave per SIMD issuing both streams itself: per MFMA, NV VALU instructions (VK as above)
template <int MK, int VK, int NV>
__global__ __launch_bounds__(256) void mixed(int iters, float* out, const float* seed) {
  const float x = seed[threadIdx.x & 255];
  f16x8v h;
  for (int j = 0; j < 8; ++j) h[j] = (_Float16)(x * (j + 1));
  float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3;
  f32x4v a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  f32x16v b0 = {0}, b1 = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (MK == 0) {
        if (k == 0) a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a0, 0, 0, 0);
        if (k == 1) a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a1, 0, 0, 0);
        if (k == 2) a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a2, 0, 0, 0);
        if (k == 3) a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, a3, 0, 0, 0);
      } else {
        if (k & 1) b1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, b1, 0, 0, 0);
        else b0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, b0, 0, 0, 0);
      }
#pragma unroll
      for (int n = 0; n < NV; ++n) {
        if (VK == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(n & 1 ? v1 : v0) : "v"(1e-9f));
        else asm volatile("v_exp_f32 %0, %0" : "+v"(n & 1 ? v3 : v2));
      }
    }
  }
  if (v0 + v1 + v2 + v3 + a0[0] + a1[1] + a2[2] + a3[3] + b0[0] + b1[1] == 12345.f) out[threadIdx.x] = v0;
}

template <typename F>
static float timeit(F launch) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

template <int MK, int VK>
static void report(const char* name, int im, int iv, float* out, const float* seed) {
  auto run = [&](int m, int v) {
    return timeit([&] { hipLaunchKernelGGL((probe<MK, VK>), dim3(256), dim3(512), 0, 0, m, v, im, iv, out, seed); });
  };
  const float tm = run(1, 0), tv = run(0, 1), tb = run(1, 1);
  printf("%-44s mfma alone %7.3f ms   partner alone %7.3f ms   both %7.3f ms   (sum %7.3f, max %7.3f)\n", name, tm, tv, tb,
         tm + tv, tm > tv ? tm : tv);
}

template <int MK, int VK, int NV>
static void report_mixed(const char* name, int iters, float* out, const float* seed) {
  const float t = timeit([&] { hipLaunchKernelGGL((mixed<MK, VK, NV>), dim3(256), dim3(256), 0, 0, iters, out, seed); });
  printf("%-44s %7.3f ms  = %6.1f ns per (MFMA + %d VALU)\n", name, t, t * 1e6 / (4.0 * iters), NV);
}

int main() {
  float *out, *seed;
  hipMalloc(&out, 4096);
  hipMalloc(&seed, 1024);
  float hs[256];
  for (int i = 0; i < 256; ++i) hs[i] = 0.37f + 0.013f * ((i * 7919) % 101);
  hipMemcpy(seed, hs, 1024, hipMemcpyHostToDevice);
  printf("waves 0-3: MFMA stream; waves 4-7 (SIMD partners): VALU stream; 256 workgroups\n");
  report<0, 0>("16x16x32 f16 (x4 per iter) + v_fma_f32 x8", 160000, 100000, out, seed);
  report<0, 1>("16x16x32 f16 (x4 per iter) + v_exp_f32 x8", 160000, 60000, out, seed);
  report<1, 0>("32x32x16 f16 (x2 per iter) + v_fma_f32 x8", 160000, 100000, out, seed);
  report<1, 1>("32x32x16 f16 (x2 per iter) + v_exp_f32 x8", 160000, 60000, out, seed);
  printf("one wave per SIMD issuing both streams (per MFMA: n VALU)\n");
  report_mixed<0, 0, 0>("16x16x32 + 0", 100000, out, seed);
  report_mixed<0, 0, 1>("16x16x32 + 1 v_fma", 100000, out, seed);
  report_mixed<0, 0, 2>("16x16x32 + 2 v_fma", 100000, out, seed);
  report_mixed<0, 0, 4>("16x16x32 + 4 v_fma", 100000, out, seed);
  report_mixed<0, 1, 1>("16x16x32 + 1 v_exp", 100000, out, seed);
  report_mixed<0, 1, 2>("16x16x32 + 2 v_exp", 100000, out, seed);
  report_mixed<1, 0, 0>("32x32x16 + 0", 100000, out, seed);
  report_mixed<1, 0, 2>("32x32x16 + 2 v_fma", 100000, out, seed);
  report_mixed<1, 0, 4>("32x32x16 + 4 v_fma", 100000, out, seed);
  report_mixed<1, 0, 6>("32x32x16 + 6 v_fma", 100000, out, seed);
  report_mixed<1, 1, 2>("32x32x16 + 2 v_exp", 100000, out, seed);
  report_mixed<1, 1, 3>("32x32x16 + 3 v_exp", 100000, out, seed);
  report_mixed<1, 1, 4>("32x32x16 + 4 v_exp", 100000, out, seed);
  return 0;
}
