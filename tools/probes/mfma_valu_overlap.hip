// Probe: do fp32 VALU / transcendental instructions of one wave overlap with the v_mfma_f32_16x16x4_f32 stream of ANOTHER wave
// on the same SIMD?  A workgroup of 8 waves per CU: waves 0-3 (one per SIMD) run an MFMA loop, waves 4-7 (their SIMD
// partners) run a VALU loop.  Timed: MFMA waves alone, VALU waves alone, both.  If t(both) ~ max(...) the pipes overlap; if
// t(both) ~ t(mfma) + t(valu) they share an issue/execution resource.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_valu_overlap tools/probes/mfma_valu_overlap.hip && gpurun_out/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4v __attribute__((ext_vector_type(4)));

template <int KIND>   // VALU partner: 0 = v_fma_f32, 1 = v_exp_f32 (transcendental), 2 = v_pk_fma_f32, 3 = f16 MFMA partner
__global__ __launch_bounds__(512) void probe(int do_mfma, int do_valu, int iters_mfma, int iters_valu, float* out) {
  const int wave = threadIdx.x >> 6;
  const float x = 1.0f + threadIdx.x * 1e-6f;
  if (wave < 4 && do_mfma != 2) {
    if (!do_mfma) return;
    f32x4v a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    for (int i = 0; i < iters_mfma; ++i) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a3, 0, 0, 0);
    }
    if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f) out[threadIdx.x] = a0[0];
  } else {
    if (!do_valu) return;
    float v0 = x, v1 = x + 1, v2 = x + 2, v3 = x + 3, v4 = x + 4, v5 = x + 5, v6 = x + 6, v7 = x + 7;
    if (KIND == 0) {
      for (int i = 0; i < iters_valu; ++i) {
        asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                     "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7) : "v"(1e-9f));
      }
    } else if (KIND == 1) {
      for (int i = 0; i < iters_valu; ++i) {
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                     "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                     : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
      }
    } else if (KIND == 2) {
      typedef float f32x2v __attribute__((ext_vector_type(2)));
      f32x2v p0 = {v0, v1}, p1 = {v2, v3}, p2 = {v4, v5}, p3 = {v6, v7}, c = {1e-9f, 1e-9f};
      for (int i = 0; i < iters_valu; ++i) {
        asm volatile("v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                     "v_pk_fma_f32 %0, %0, %4, %0\n v_pk_fma_f32 %1, %1, %4, %1\n v_pk_fma_f32 %2, %2, %4, %2\n v_pk_fma_f32 %3, %3, %4, %3\n"
                     : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(c));
      }
      v0 = p0[0] + p1[0] + p2[0] + p3[0];
    } else {
      typedef _Float16 f16x8v __attribute__((ext_vector_type(8)));
      f16x8v h;
      for (int j = 0; j < 8; ++j) h[j] = (_Float16)x;
      f32x4v b0 = {0, 0, 0, 0}, b1 = b0, b2 = b0, b3 = b0;
      for (int i = 0; i < iters_valu; ++i) {
        b0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, b0, 0, 0, 0);
        b1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, b1, 0, 0, 0);
        b2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, b2, 0, 0, 0);
        b3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, b3, 0, 0, 0);
      }
      v0 = b0[0] + b1[1] + b2[2] + b3[3];
    }
    if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 12345.f) out[threadIdx.x] = v0;
  }
}

template <int KIND>
static float run(int do_mfma, int do_valu, int im, int iv, float* out) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, do_mfma, do_valu, im, iv, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(probe<KIND>, dim3(256), dim3(512), 0, 0, do_mfma, do_valu, im, iv, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}

template <int KIND>
static void report(const char* name, int im, int iv, float* out) {
  const float tm = run<KIND>(1, 0, im, iv, out), tv = run<KIND>(0, 1, im, iv, out), tb = run<KIND>(1, 1, im, iv, out);
  printf("%-34s mfma alone %7.3f ms   partner alone %7.3f ms   both %7.3f ms   (sum %7.3f, max %7.3f)\n", name, tm, tv, tb,
         tm + tv, tm > tv ? tm : tv);
}

int main() {
  float* out;
  hipMalloc(&out, 4096);
  const int im = 40000;   // 160k MFMAs per wave
  printf("one MFMA wave + one partner wave per SIMD, 256 workgroups of 8 waves (1 per CU)\n");
  const float tm = run<0>(1, 0, im, 0, out);
  printf("fp32 16x16x4 MFMA alone: %.3f ms -> %.1f TFLOP/s (4 waves/CU)\n", tm, 256.0 * 4 * 4 * im * 2048.0 / tm / 1e9);
  report<0>("partner = v_fma_f32 x8", im, 200000, out);
  report<2>("partner = v_pk_fma_f32 x8", im, 200000, out);
  report<1>("partner = v_exp_f32 x8", im, 60000, out);
  report<3>("partner = f16 16x16x32 MFMA x4", im, 80000, out);
  // the same VALU loop on ONE wave per SIMD vs on TWO waves per SIMD (do_mfma = 2: waves 0-3 run the partner loop too)
  printf("v_fma_f32 x8:    1 wave/SIMD %7.3f ms   2 waves/SIMD %7.3f ms (twice the work)\n", run<0>(0, 1, 0, 200000, out), run<0>(2, 1, 0, 200000, out));
  printf("v_pk_fma_f32 x8: 1 wave/SIMD %7.3f ms   2 waves/SIMD %7.3f ms\n", run<2>(0, 1, 0, 200000, out), run<2>(2, 1, 0, 200000, out));
  printf("v_exp_f32 x8:    1 wave/SIMD %7.3f ms   2 waves/SIMD %7.3f ms\n", run<1>(0, 1, 0, 60000, out), run<1>(2, 1, 0, 60000, out));
  printf("f16 MFMA x4:     1 wave/SIMD %7.3f ms   2 waves/SIMD %7.3f ms\n", run<3>(0, 1, 0, 80000, out), run<3>(2, 1, 0, 80000, out));
  return 0;
}
