// swap_probe.hip — what v_permlane32_swap / v_permlane16_swap do on gfx950 (hmm_kernels.hip: swap_halves, swap_rows rely on it).
// Prints, for both results, which (register, lane) every lane ends up with.   hipcc --offload-arch=gfx950 -o tools/swap_probe tools/swap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out){
  const unsigned l = threadIdx.x;
  const unsigned a = l, b = 100 + l;
  const auto h = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  const auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[l] = h[0]; out[64 + l] = h[1]; out[128 + l] = r[0]; out[192 + l] = r[1];
}
int main(){
  unsigned* d; unsigned h[256];
  if (hipMalloc(&d, sizeof h) != hipSuccess){ printf("no device\n"); return 1; }
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  const char* names[4] = { "permlane32_swap a'", "permlane32_swap b'", "permlane16_swap a'", "permlane16_swap b'" };
  for (int i = 0; i < 4; i++){
    printf("%s:", names[i]);
    for (int l = 0; l < 64; l += 16) printf("  lanes %2d..%2d <- %s[%u..%u]", l, l + 15, h[64*i + l] >= 100 ? "b" : "a", h[64*i + l] % 100, h[64*i + l + 15] % 100);
    printf("\n");
  }
  return 0;
}
