// step_probe.hip — the fixed cost of one step of the cooperative flank sweep (hs_lead/hs_trail_kernel_coop): W wavefronts of a workgroup in a
// systolic chain, each step = boundary (2 doubles per lane) from the neighbour's LDS ring slot, R rows of the max-plus recurrence on registers
// (M/I bottom-up: independent across rows; D top-down: 2 dependent FP64 operations per row), boundary to the own ring slot, s_waitcnt + s_barrier.
// Prints shader cycles per step for the full step and with parts left out, for a few (W, R) and workgroups per launch.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/step_probe.hip -o tools/step_probe && tools/step_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int R, int MODE>     // MODE bit 0: barrier, bit 1: LDS ring, bit 2: arithmetic
__global__ void __launch_bounds__(512) sweep(double* out, long long* cyc, int nsteps, int W){
  __shared__ double2 ring[8][2*64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double Mp[R], Ip[R], Dp[R];
  for (int r = 0; r < R; r++){ Mp[r] = -1.0 - r - lane; Ip[r] = -2.0 - r; Dp[r] = -3.0 - lane; }
  double diagM = -1.0, diagD = -2.0;
  const double k1 = -0.001, k2 = -4.5, k3 = -2.25, k4 = -0.7, e = -0.01;
  ring[w][lane] = make_double2(-1.0, -2.0); ring[w][64 + lane] = make_double2(-1.5, -2.5);
  __syncthreads();
  const long long c0 = clock64();
  for (int t = 0; t < nsteps; t++){
    double upM = -1.0 - t, upD = -2.0;
    if (MODE & 2){ const double2 b = ring[w > 0 ? w - 1 : W - 1][(t & 1)*64 + lane]; upM = b.x; upD = b.y; }
    const double topM = upM, topD = upD;
    if (MODE & 4){
#pragma unroll
      for (int r = R - 1; r >= 0; r--){
        const double dM = (r == 0) ? diagM : Mp[r > 0 ? r-1 : 0], dD = (r == 0) ? diagD : Dp[r > 0 ? r-1 : 0];
        const double nM = e + fmax(dM + k1, fmax(Ip[r], dD) + k2);
        const double nI = e + fmax(dM + k3, Ip[r] + k4);
        Mp[r] = nM; Ip[r] = nI;
      }
#pragma unroll
      for (int r = 0; r < R; r++){
        const double nD = fmax(upM + k2, upD + k4);
        Dp[r] = nD; upM = Mp[r]; upD = nD;
      }
    }
    if (MODE & 2) ring[w][(t & 1)*64 + lane] = make_double2(upM, upD);
    diagM = topM; diagD = topD;
    if (MODE & 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long c1 = clock64();
  double s = diagM + diagD;
  for (int r = 0; r < R; r++) s += Mp[r] + Ip[r] + Dp[r];
  out[blockIdx.x*512 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = c1 - c0;
}
template <int R, int MODE> void run(const char* what, int W, int blocks, double* d, long long* c){
  const int nsteps = 2000;
  hipLaunchKernelGGL((sweep<R, MODE>), dim3(blocks), dim3(64*W), 0, 0, d, c, nsteps, W);
  hipLaunchKernelGGL((sweep<R, MODE>), dim3(blocks), dim3(64*W), 0, 0, d, c, nsteps, W);
  hipDeviceSynchronize();
  long long h; hipMemcpy(&h, c, 8, hipMemcpyDeviceToHost);
  printf("R %2d W %d blocks %5d  %-34s %7.0f cycles per step = %.3f us\n", R, W, blocks, what, (double)h/nsteps, (double)h/nsteps/2400.0);
}
int main(){
  double* d; long long* c; hipMalloc(&d, 8*512*4096); hipMalloc(&c, 8*4096);
  for (int W : {4, 8}) for (int blocks : {40, 1024}){
    run<8, 7>("full step", W, blocks, d, c);
    run<8, 6>("no barrier", W, blocks, d, c);
    run<8, 5>("no LDS ring", W, blocks, d, c);
    run<8, 3>("no arithmetic", W, blocks, d, c);
    run<8, 4>("arithmetic only", W, blocks, d, c);
    run<8, 1>("barrier only", W, blocks, d, c);
    run<15, 7>("full step", W, blocks, d, c);
    run<15, 4>("arithmetic only", W, blocks, d, c);
    run<4, 7>("full step", W, blocks, d, c);
    run<4, 4>("arithmetic only", W, blocks, d, c);
  }
  return 0;
}
