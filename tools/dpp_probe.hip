// Probe: does v_mov_b32_dpp wave_shr:1 shift across all 64 lanes on this GPU? (used by the systolic DP sweep)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int* out){
  int lane = threadIdx.x;
  int v = 100 + lane;
  int s = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   // wave_shr:1, lane0 keeps old (-1)
  int r = __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(out[64]));
  out[lane] = s; out[65+lane] = r;
}
int main(){
  int* d; hipMalloc(&d, 256*sizeof(int));
  int h[256] = {0}; h[64] = 37;
  hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1,64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 64; i++){ int want = i == 0 ? -1 : 100+i-1; if (h[i] != want){ ok = 0; printf("lane %d got %d want %d\n", i, h[i], want); } }
  printf("wave_shr:1 %s; readlane(37) = %d\n", ok ? "OK" : "BROKEN", h[65]);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  printf("%s CUs=%d clock=%d MHz lds/block=%zu arch=%s\n", p.name, p.multiProcessorCount, p.clockRate/1000, p.sharedMemPerBlock, p.gcnArchName);
  return ok ? 0 : 1;
}
