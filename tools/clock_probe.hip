// clock_probe.hip — what shader clock do SHORT launches run at?  A one-wavefront kernel times a fixed chain of dependent FP64 additions with
// the shader-clock counter (clock64: s_memtime) and the constant 100 MHz counter (wall_clock64: s_memrealtime); their ratio is the shader
// clock during the kernel.  Launched (a) after the device idled for 50 ms, (b) back to back, (c) right behind a kernel that kept all CUs busy
// for ~20 ms, (d) while such a kernel is still running on another stream.
//   hipcc --offload-arch=gfx950 -O2 tools/clock_probe.hip -o tools/clock_probe && tools/clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
__global__ void chain(double* out, long long* t, int n){
  double a = out[0];
  const long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < n; i++) a = a + 1.0000001;
  const long long c1 = clock64(), w1 = wall_clock64();
  out[0] = a; t[0] = c1 - c0; t[1] = w1 - w0;
}
__global__ void burn(double* out, int n){
  double a = out[threadIdx.x & 1] + threadIdx.x;
  for (int i = 0; i < n; i++) a = a*1.0000001 + 0.5;
  if (a == 12345.678) out[2] = a;
}
int main(){
  double* d; long long* t; hipMalloc(&d, 64); hipMalloc(&t, 16); hipMemset(d, 0, 64);
  hipStream_t s1, s2; hipStreamCreate(&s1); hipStreamCreate(&s2);
  auto run = [&](const char* what){
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, t, 20000);
    hipStreamSynchronize(s1);
    long long h[2]; hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("%-44s %8lld shader cycles in %7.1f us = %6.0f MHz, %5.2f cycles per dependent v_add_f64\n", what, h[0], h[1]/100.0, 100.0*h[0]/h[1], (double)h[0]/20000);
  };
  run("first launch of the process");
  usleep(50000); run("after 50 ms idle");
  run("back to back"); run("back to back"); run("back to back");
  for (int k = 0; k < 3; k++){
    hipLaunchKernelGGL(burn, dim3(2048), dim3(256), 0, s1, d, 400000); hipStreamSynchronize(s1);
    run("right behind a ~20 ms all-CU kernel");
  }
  hipLaunchKernelGGL(burn, dim3(2048), dim3(256), 0, s2, d, 2000000);
  usleep(3000);
  run("while an all-CU kernel runs on another stream"); run("while an all-CU kernel runs on another stream");
  hipStreamSynchronize(s2);
  usleep(2000); run("2 ms after it ended"); usleep(20000); run("20 ms later"); usleep(200000); run("200 ms later");
  // many short launches in a row: does the clock come up by itself?
  for (int k = 0; k < 2000; k++) hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, s1, d, t, 20000);
  hipStreamSynchronize(s1); run("after 2000 short launches in a row");
  return 0;
}
