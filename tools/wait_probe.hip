// What does a host thread burn while it waits for the device?  hipEventSynchronize on a plain / blocking-sync event, hipStreamSynchronize, and a query + usleep loop.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <time.h>
#include <unistd.h>
static double wall(){ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double cpu(){ timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return ts.tv_sec + 1e-9*ts.tv_nsec; }
__global__ void spin_kernel(long long cycles, int* out){ const long long t0 = clock64(); while (clock64() - t0 < cycles){} if (out) *out = 1; }
int main(){
  hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  hipEvent_t plain, block; hipEventCreateWithFlags(&plain, hipEventDisableTiming); hipEventCreateWithFlags(&block, hipEventDisableTiming | hipEventBlockingSync);
  const long long cyc = 100000000LL;      // ~50 ms at 2 GHz
  for (int mode = 0; mode < 4; mode++){
    hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, st, cyc, (int*)NULL);
    hipEvent_t ev = mode == 1 ? block : plain;
    hipEventRecord(ev, st);
    const double w0 = wall(), c0 = cpu();
    if (mode == 0 || mode == 1) hipEventSynchronize(ev);
    else if (mode == 2) hipStreamSynchronize(st);
    else while (hipEventQuery(ev) == hipErrorNotReady) usleep(100);
    printf("%-34s wall %6.2f ms  thread cpu %6.2f ms\n", mode == 0 ? "hipEventSynchronize (plain event)" : mode == 1 ? "hipEventSynchronize (blocking)" : mode == 2 ? "hipStreamSynchronize" : "hipEventQuery + usleep(100)",
           1e3*(wall() - w0), 1e3*(cpu() - c0));
  }
  return 0;
}
