// How fast does the CPU read / write host memory the device can reach, by allocation kind?  (hipcc -O2 tools/pinned_probe.hip -o /tmp/pinned_probe)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
static volatile unsigned long g_sink = 0;
static void use(const char* p, size_t n){ unsigned long x = 0; for (size_t i = 0; i < n; i += 4096) x += (unsigned char)p[i]; g_sink += x; }
static double now(){ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(){
  const size_t N = (size_t)128 << 20;
  char* plain = (char*)malloc(N); memset(plain, 1, N);
  char* dst = (char*)malloc(N); memset(dst, 2, N);
  struct Kind { const char* name; unsigned flags; int reg; } kinds[] = {
    {"hipHostMalloc default", hipHostMallocDefault, 0}, {"hipHostMalloc NonCoherent", hipHostMallocNonCoherent, 0},
    {"hipHostMalloc Coherent", hipHostMallocCoherent, 0}, {"hipHostMalloc Portable|Mapped", hipHostMallocPortable | hipHostMallocMapped, 0},
    {"malloc + hipHostRegister", 0, 1} };
  void* d = NULL; hipMalloc(&d, N);
  for (const Kind& k : kinds){
    char* p = NULL;
    if (k.reg){ p = (char*)aligned_alloc(4096, N); memset(p, 3, N); if (hipHostRegister(p, N, hipHostRegisterDefault) != hipSuccess){ printf("%s: register failed\n", k.name); continue; } }
    else if (hipHostMalloc((void**)&p, N, k.flags) != hipSuccess){ printf("%s: alloc failed\n", k.name); continue; }
    memset(p, 4, N);
    double t0 = now(); memcpy(p, plain, N); double tw = now() - t0;          // CPU write into it
    t0 = now(); memcpy(dst, p, N); double tr = now() - t0; use(dst, N);                    // CPU read from it
    t0 = now(); hipMemcpy(d, p, N, hipMemcpyHostToDevice); double th = now() - t0;
    t0 = now(); hipMemcpy(p, d, N, hipMemcpyDeviceToHost); double td = now() - t0;
    t0 = now(); memcpy(dst, p, N); double tr2 = now() - t0; use(dst, N);                   // CPU read after the device wrote it
    printf("%-34s cpu write %6.2f GB/s  cpu read %6.2f GB/s  H2D %6.2f GB/s  D2H %6.2f GB/s  cpu read after D2H %6.2f GB/s\n", k.name, N/tw/1e9, N/tr/1e9, N/th/1e9, N/td/1e9, N/tr2/1e9);
  }
  double t0 = now(); memcpy(dst, plain, N); double tc = now() - t0; use(dst, N); printf("%-34s cpu copy  %6.2f GB/s\n", "malloc -> malloc", N/tc/1e9);
  return 0;
}
