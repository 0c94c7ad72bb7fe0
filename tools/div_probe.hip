// div_probe.hip — is a 6-instruction float division (v_rcp_f32 + Newton step + residual correction) the IEEE quotient on the two operand
// families of the reference's bit-trick exp / log (fastonebigheader.h:188-198: 27.7280233f / (4.84252568f - z), z in [0, 1];
// :320-338: 1.72587999f / (0.3520887068f + mx), mx in [0.5, 1))?  Exhaustive over every float denominator of both ranges (+ a margin).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o build/div_probe tools/div_probe.hip && build/div_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
__device__ __forceinline__ float fast_div(float n, float d){
  float r = __builtin_amdgcn_rcpf(d);
  r = fmaf(fmaf(-d, r, 1.0f), r, r);
  float q = n * r;
  const float e = fmaf(-d, q, n);
  return fmaf(e, r, q);
}
__global__ void probe(float n, uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first){
  const uint64_t i = (uint64_t)blockIdx.x*blockDim.x + threadIdx.x;
  const uint64_t b = (uint64_t)lo + i;
  if (b > hi) return;
  const float d = __uint_as_float((uint32_t)b);
  const float want = __fdiv_rn(n, d), got = fast_div(n, d);
  if (__float_as_uint(want) != __float_as_uint(got)){ if (atomicAdd(bad, 1ull) == 0) *first = (uint32_t)b; }
}
static uint32_t bits(float f){ uint32_t u; memcpy(&u, &f, 4); return u; }
int main(){
  unsigned long long* bad; uint32_t* first;
  hipMalloc(&bad, 8); hipMalloc(&first, 4);
  struct { float n, lo, hi; const char* what; } fam[2] = { {27.7280233f, 3.80f, 4.90f, "pow2: 27.7280233 / [3.80, 4.90]"}, {1.72587999f, 0.84f, 1.36f, "log: 1.72587999 / [0.84, 1.36]"} };
  int rc = 0;
  for (auto& f : fam){
    hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    const uint32_t lo = bits(f.lo), hi = bits(f.hi); const uint64_t cnt = (uint64_t)hi - lo + 1;
    probe<<<(unsigned)((cnt + 255)/256), 256>>>(f.n, lo, hi, bad, first);
    unsigned long long nb = 0; uint32_t fb = 0; hipMemcpy(&nb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&fb, first, 4, hipMemcpyDeviceToHost);
    printf("%s: %llu denominators, %llu differ from the IEEE quotient%s\n", f.what, (unsigned long long)cnt, nb, nb ? " (first bits below)" : "");
    if (nb){ printf("  first: 0x%08x\n", fb); rc = 1; }
  }
  return rc;
}
