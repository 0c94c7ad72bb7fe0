// post_kernels.hip — per-sample diplotype posteriors on gfx950.
//
// Genotyper::calc_log_sample_posteriors (genotyper.cpp:44-80) + get_optimal_haplotypes
// (genotyper.cpp:82-97).  One workgroup per (locus, sample); each thread owns diplotypes
// (a1,a2) and walks the sample's reads in read order, so the scatter-add of the reference
// becomes a private sequential accumulation with the reference's exact operation order
// (bit-identical, including the float pair log-sum-exp of mathops.cpp:86-95).  The final
// exact log-sum-exp over the A^2 diplotypes (mathops.cpp:44-50) is a wavefront + LDS tree
// reduction; it differs from the reference's sequential libm sum only by rounding
// (|diff| <~ 1e-13, tolerance stated in tests/test_posteriors_gpu.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "post_layout.h"

namespace {

__device__ __forceinline__ float f_fastpow2(float p){             // fastonebigheader.h:188-198
  const float offset = (p < 0.0f) ? 1.0f : 0.0f;
  const float clipp = (p < -126.0f) ? -126.0f : p;
  const int w = (int)clipp;
  const float z = __fadd_rn(__fsub_rn(clipp, (float)w), offset);
  const float t = __fsub_rn(__fadd_rn(__fadd_rn(clipp, 121.2740575f), __fdiv_rn(27.7280233f, __fsub_rn(4.84252568f, z))), __fmul_rn(1.49012907f, z));
  return __uint_as_float((uint32_t)__fmul_rn(8388608.0f, t));
}
__device__ __forceinline__ float f_fastexp(float p){ return f_fastpow2(__fmul_rn(1.442695040f, p)); }
__device__ __forceinline__ float f_fastlog(float x){              // fastonebigheader.h:320-338
  const uint32_t vi = __float_as_uint(x);
  const float mx = __uint_as_float((vi & 0x007FFFFFu) | 0x3f000000u);
  float y = (float)vi;
  y = __fmul_rn(y, 1.1920928955078125e-7f);
  const float l2 = __fsub_rn(__fsub_rn(__fsub_rn(y, 124.22551499f), __fmul_rn(1.498030302f, mx)),
                             __fdiv_rn(1.72587999f, __fadd_rn(0.3520887068f, mx)));
  return __fmul_rn(0.69314718f, l2);
}
__device__ __forceinline__ double fast_lse2(double a, double b, double thr){    // mathops.cpp:86-95
  const double hi = a > b ? a : b, lo = a > b ? b : a;
  const double diff = lo - hi;
  return diff < thr ? hi : hi + (double)f_fastlog(__fadd_rn(1.0f, f_fastexp((float)diff)));
}

}  // namespace

extern "C" __global__ void __launch_bounds__(256)
hs_posterior_kernel(const hs_post_dev_t* __restrict__ dp){
  const hs_post_dev_t& d = *dp;
  const hs_post_unit_t u = d.units[blockIdx.x];
  const int A = u.n_alleles, nd = A*A, tid = threadIdx.x;
  double* post = d.log_post + u.post_off;
  const double* LL0 = d.log_aln_probs + u.ll_off;      // row of the sample's first read

  __shared__ double red_v[256];
  __shared__ int    red_i[256];

  // ---- accumulate (genotyper.cpp:47-61)
  double lmax = -1.0e300;
  for (int idx = tid; idx < nd; idx += 256){
    const int a1 = idx / A, a2 = idx - a1*A;
    double v = d.log_prior ? d.log_prior[u.post_off + idx] : ((a1 == a2) ? u.log_hom_prior : u.log_het_prior);
    for (int r = 0; r < u.n_reads; r++){
      const int g = u.read_begin + r;
      const double* LL = LL0 + (int64_t)r*A;
      const double x = fast_lse2((d.log_half + d.log_p1[g]) + LL[a1], (d.log_half + d.log_p2[g]) + LL[a2], d.log_thresh);
      v += (double)d.read_weight[g] * x;
    }
    post[idx] = v;
    lmax = fmax(lmax, v);
  }
  // ---- exact log-sum-exp over diplotypes (genotyper.cpp:63-72)
  red_v[tid] = lmax; __syncthreads();
  for (int s = 128; s > 0; s >>= 1){ if (tid < s) red_v[tid] = fmax(red_v[tid], red_v[tid+s]); __syncthreads(); }
  const double mx = red_v[0]; __syncthreads();
  double lsum = 0.0;
  for (int idx = tid; idx < nd; idx += 256) lsum += exp(post[idx] - mx);
  red_v[tid] = lsum; __syncthreads();
  for (int s = 128; s > 0; s >>= 1){ if (tid < s) red_v[tid] += red_v[tid+s]; __syncthreads(); }
  const double total = mx + log(red_v[0]); __syncthreads();
  // ---- normalise + MAP diplotype: first maximum in a1-major order (genotyper.cpp:88-95)
  double bv = -1.7976931348623157e308; int bi = 0x7fffffff;
  for (int idx = tid; idx < nd; idx += 256){
    const double v = post[idx] - total;
    post[idx] = v;
    if (v > bv){ bv = v; bi = idx; }
  }
  red_v[tid] = bv; red_i[tid] = bi; __syncthreads();
  for (int s = 128; s > 0; s >>= 1){
    if (tid < s){
      const double ov = red_v[tid+s]; const int oi = red_i[tid+s];
      if (ov > red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])){ red_v[tid] = ov; red_i[tid] = oi; }
    }
    __syncthreads();
  }
  if (tid == 0){
    d.sample_total[u.samp_index] = total;
    d.map_gt[2*u.samp_index]   = red_i[0] / A;
    d.map_gt[2*u.samp_index+1] = red_i[0] % A;
  }
}
