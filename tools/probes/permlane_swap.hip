// What v_permlane32_swap / v_permlane16_swap (gfx950) do to two registers, lane by lane — the building blocks of a 4 x 4
// transpose among the four 16-lane rows of a wave (used to turn a lane's 64-byte run into whole-line stores).
//   hipcc --offload-arch=gfx950 -O2 -o permlane_swap tools/probes/permlane_swap.hip && ./permlane_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned lane = threadIdx.x;
  unsigned x = 1000 + lane, y = 2000 + lane;          // x: register A, y: register B
  auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  o[lane] = r[0]; o[64 + lane] = r[1];
  auto s = __builtin_amdgcn_permlane16_swap(x, y, false, false);
  o[128 + lane] = s[0]; o[192 + lane] = s[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane32_swap -> A", "permlane32_swap -> B", "permlane16_swap -> A", "permlane16_swap -> B"};
  for (int b = 0; b < 4; ++b) {
    printf("%s (lanes 0,16,32,48 and 1,17,33,49):", names[b]);
    for (int l : {0, 16, 32, 48, 1, 17, 33, 49}) printf(" %u", h[64 * b + l]);
    printf("\n");
  }
  return 0;
}
