// post_kernels.hip — per-sample diplotype posteriors on gfx950.
//
// Genotyper::calc_log_sample_posteriors (genotyper.cpp:44-80) + get_optimal_haplotypes
// (genotyper.cpp:82-97).  One workgroup per (locus, sample); each thread owns diplotypes
// (a1,a2) and walks the sample's reads in read order, so the scatter-add of the reference
// becomes a private sequential accumulation with the reference's exact operation order
// (bit-identical, including the float pair log-sum-exp of mathops.cpp:86-95).  The final
// exact log-sum-exp over the A^2 diplotypes (mathops.cpp:44-50) — round 5: the exponentials are CORRECTLY ROUNDED (cr_math.h), computed
// by all threads, and summed by one thread in the reference's index order, so the total and the normalised posteriors have the host's
// bits wherever the host's libm is itself correctly rounded (glibc: all but ~8 in 10^4 exp arguments, ~1 in 10^5 log arguments;
// tests/test_cr_math.py).  What the genotype stage's FLOAT pair log-sum-exps are fed is therefore what the reference feeds them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "post_layout.h"
#include "cr_math.h"

#define HS_POST_ECHUNK 2048
#define HS_POST_REGS 8            // diplotypes per thread a unit may have to stay in registers (256 x 8 = HS_POST_ECHUNK: its exponentials fit the LDS chunk)

namespace {

// The two float divisions of the reference's bit-trick exp2 / log (fastonebigheader.h:188-198: 27.7280233f / (4.84252568f - z), z in [0, 1];
// :320-338: 1.72587999f / (0.3520887068f + mx), mx in [0.5, 1)) as v_rcp_f32 + one Newton step + a residual correction: six instructions
// instead of the compiler's IEEE sequence (scale, reciprocal, three refinements, fmas, fixup: twice that), and the IEEE quotient for EVERY
// float denominator of both ranges — tools/div_probe.hip checks all 8.4 M of them on the device.  Operands outside those ranges: never here.
__device__ __forceinline__ float f_div_tab(float n, float d){
  float r = __builtin_amdgcn_rcpf(d);
  r = __fmaf_rn(__fmaf_rn(-d, r, 1.0f), r, r);
  const float q = __fmul_rn(n, r);
  return __fmaf_rn(__fmaf_rn(-d, q, n), r, q);
}
__device__ __forceinline__ float f_fastpow2(float p){             // fastonebigheader.h:188-198
  const float offset = (p < 0.0f) ? 1.0f : 0.0f;
  const float clipp = (p < -126.0f) ? -126.0f : p;
  const int w = (int)clipp;
  const float z = __fadd_rn(__fsub_rn(clipp, (float)w), offset);
  const float t = __fsub_rn(__fadd_rn(__fadd_rn(clipp, 121.2740575f), f_div_tab(27.7280233f, __fsub_rn(4.84252568f, z))), __fmul_rn(1.49012907f, z));
  return __uint_as_float((uint32_t)__fmul_rn(8388608.0f, t));
}
__device__ __forceinline__ float f_fastexp(float p){ return f_fastpow2(__fmul_rn(1.442695040f, p)); }
__device__ __forceinline__ float f_fastlog(float x){              // fastonebigheader.h:320-338
  const uint32_t vi = __float_as_uint(x);
  const float mx = __uint_as_float((vi & 0x007FFFFFu) | 0x3f000000u);
  float y = (float)vi;
  y = __fmul_rn(y, 1.1920928955078125e-7f);
  const float l2 = __fsub_rn(__fsub_rn(__fsub_rn(y, 124.22551499f), __fmul_rn(1.498030302f, mx)),
                             f_div_tab(1.72587999f, __fadd_rn(0.3520887068f, mx)));
  return __fmul_rn(0.69314718f, l2);
}
__device__ __forceinline__ double fast_lse2(double a, double b, double thr){    // mathops.cpp:86-95
  const double hi = a > b ? a : b, lo = a > b ? b : a;
  const double diff = lo - hi;
  return diff < thr ? hi : hi + (double)f_fastlog(__fadd_rn(1.0f, f_fastexp((float)diff)));
}

}  // namespace

// phase 0: the whole of it, one workgroup per (locus, sample).  With few units and many diplotypes (configs[4]: 256 loci x ONE sample x
// 128^2 diplotypes x 200 reads: 256 workgroups of 64 sequential diplotypes per thread on 256 CUs) the accumulation is split over
// gridDim.y workgroups per unit (phase 1: diplotypes idx = 256 (slice + k gridDim.y) + tid) and the rest — maximum, exact log-sum-exp,
// normalisation, MAP scan — runs as before in a second launch (phase 2) on the stored sums: a diplotype's accumulation does not depend on
// which thread owns it and phase 2 is phase 0's second half on the same values, so the results are bit-identical however a batch is split.
template <int PHASE>
__device__ __forceinline__ void posterior_body(const hs_post_dev_t& d){
  int ub = blockIdx.x;
  if (d.unit_list){ if (ub >= *d.n_list) return; ub = d.unit_list[ub]; }
  if (d.unit_active && !d.unit_active[ub]) return;
  const hs_post_unit_t u = d.units[ub];
  const int A = u.n_alleles, nd = A*A, tid = threadIdx.x;
  double* post = d.log_post + u.post_off;
  const double* LL0 = d.log_aln_probs + u.ll_off;      // row of the sample's first read

  __shared__ double red_v[256];
  __shared__ int    red_i[256];
  __shared__ double ebuf[HS_POST_ECHUNK];              // a chunk of exponentials, summed in index order by thread 0

  // ---- round 5: units of up to 2048 diplotypes (A <= 45: nearly all) keep their values in REGISTERS from the accumulation to the normalised
  // posteriors — the general path below writes them, reads them back for the maximum, for the first-maximum search, for the exponentials and
  // for the normalisation (40 KB of traffic per 1024-diplotype unit, a million units per EM round) and pays ~30 workgroup barriers in
  // its three tree reductions; here the reductions are wavefront shuffles plus one LDS exchange.  Same operations on the same values in the
  // same order (the sum of the exponentials is still thread 0's, in index order): bit-identical.
  if (PHASE == 0 && nd <= 256*HS_POST_REGS && !d.raw){
#ifdef HS_POST_TIME
    unsigned long long tk[6]; tk[0] = __builtin_amdgcn_s_memtime();
#define HS_PT(i) tk[i] = __builtin_amdgcn_s_memtime()
#else
#define HS_PT(i) do {} while (0)
#endif
    double v[HS_POST_REGS];
    double lmax = -1.0e300;
    // Without phasing information (log_p1 == log_p2 for every read of the sample: the stutter EM, unphased samples) and with a prior
    // that does not tell the two alleles apart, diplotypes (a1, a2) and (a2, a1) accumulate the same numbers in the same order — the
    // float pair log-sum-exp takes the larger argument first whichever comes first (mathops.cpp:86-95), the default priors depend on
    // a1 == a2 only — so only the pairs a1 <= a2 are computed (A (A + 1) / 2 of A^2) and mirrored through LDS: the same bits.
    bool sym = (d.log_prior == NULL || d.sym_prior != 0) && A >= 4;
    for (int r = 0; r < u.n_reads && sym; r++) sym = (d.log_p1[u.read_begin + r] == d.log_p2[u.read_begin + r]);
    // Round 6: what a (diplotype, read) step needs — (log 1/2 + log_p1[read]) + LL[read][a1], (log 1/2 + log_p2[read]) + LL[read][a2] and the
    // read's weight — depends on (read, allele) only, not on the diplotype, and used to be fetched and added per diplotype straight from
    // memory: a chain of five loads per step, 13 steps per thread in the stutter EM's units — the accumulation was waiting for L2 most of
    // its time (55 % of a unit's 55 us).  The reads are now taken a tile at a time: all threads form the two addends per (read, allele) once,
    // side by side in LDS (the chunk buffer of the exponentials, unused until then), then every thread adds the tile's reads to its
    // diplotypes in read order.  Same additions on the same values in the same order: bit-identical.
    const int npairs = A*(A + 1)/2;
    const int n_mine = sym ? npairs : nd;                 // items dealt to the threads: item q = tid + 256 k
    int ij[HS_POST_REGS];                                 // a1 | a2 << 16 of this thread's items
#pragma unroll
    for (int k = 0; k < HS_POST_REGS; k++){
      const int q = tid + 256*k;
      int i = 0, j = 0;
      v[k] = -1.0e300;
      if (q < n_mine){
        if (sym){
          // pair q of the upper triangle, row by row: row i starts at i A - i (i - 1) / 2
          i = (int)(((float)(2*A + 1) - sqrtf((float)((2*A + 1)*(2*A + 1) - 8*q))) * 0.5f);
          i = max(0, min(i, A - 1));
          while (i > 0 && i*A - i*(i - 1)/2 > q) i--;
          while ((i + 1)*A - (i + 1)*i/2 <= q) i++;
          j = i + (q - (i*A - i*(i - 1)/2));
        } else { i = q / A; j = q - i*A; }
        v[k] = d.log_prior ? d.log_prior[u.prior_off + i*A + j] : ((i == j) ? u.log_hom_prior : u.log_het_prior);
      }
      ij[k] = i | (j << 16);
    }
    {
      const int RT = max(1, min(u.n_reads, HS_POST_ECHUNK / (2*A + 1)));      // reads per tile: two addends per (read, allele) + a weight per read
      double* const sA1 = ebuf; double* const sA2 = ebuf + RT*A; double* const sW = ebuf + 2*RT*A;
      for (int r0 = 0; r0 < u.n_reads; r0 += RT){
        const int nr = min(RT, u.n_reads - r0);
        for (int e = tid; e < nr*A; e += 256){
          const int r = e / A, g = u.read_begin + r0 + r;
          const double ll = LL0[(int64_t)(r0 + r)*A + (e - r*A)];
          sA1[e] = (d.log_half + d.log_p1[g]) + ll;
          sA2[e] = (d.log_half + d.log_p2[g]) + ll;
        }
        if (tid < nr) sW[tid] = (double)d.read_weight[u.read_begin + r0 + tid];
        for (int r = tid + 256; r < nr; r += 256) sW[r] = (double)d.read_weight[u.read_begin + r0 + r];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < HS_POST_REGS; k++){
          if (tid + 256*k < n_mine){                      // (the same for every thread of a wavefront but the last item's)
            const int i = ij[k] & 0xffff, j = ij[k] >> 16;
            double x = v[k];
            for (int r = 0; r < nr; r++) x += sW[r] * fast_lse2(sA1[r*A + i], sA2[r*A + j], d.log_thresh);
            v[k] = x;
          }
        }
        __syncthreads();                                  // (the tile buffer is written again: next tile, or the mirror / the exponentials below)
      }
    }
    if (sym){
#pragma unroll
      for (int k = 0; k < HS_POST_REGS; k++) if (tid + 256*k < npairs){ const int i = ij[k] & 0xffff, j = ij[k] >> 16; ebuf[i*A + j] = v[k]; ebuf[j*A + i] = v[k]; }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < HS_POST_REGS; k++){
        const int idx = tid + 256*k;
        v[k] = -1.0e300;
        if (idx < nd){ v[k] = ebuf[idx]; lmax = fmax(lmax, v[k]); }
      }
      __syncthreads();                                    // (ebuf is written again below)
    } else {
#pragma unroll
      for (int k = 0; k < HS_POST_REGS; k++) if (tid + 256*k < nd) lmax = fmax(lmax, v[k]);
    }
    HS_PT(1);
    const int lane = tid & 63, w = tid >> 6;
    for (int o = 32; o >= 1; o >>= 1) lmax = fmax(lmax, __shfl_xor(lmax, o));
    if (lane == 0) red_v[w] = lmax;
    __syncthreads();
    const double mx = fmax(fmax(red_v[0], red_v[1]), fmax(red_v[2], red_v[3]));
    int fi = 0x7fffffff;
#pragma unroll
    for (int k = HS_POST_REGS - 1; k >= 0; k--) if (tid + 256*k < nd && v[k] == mx) fi = tid + 256*k;
    for (int o = 32; o >= 1; o >>= 1) fi = min(fi, __shfl_xor(fi, o));
    if (lane == 0) red_i[w] = fi;
    __syncthreads();
    const int first_max = min(min(red_i[0], red_i[1]), min(red_i[2], red_i[3]));
    HS_PT(2);
#pragma unroll
    for (int k = 0; k < HS_POST_REGS; k++){
      const int idx = tid + 256*k;
      if (idx < nd){ const double x = v[k] - mx; ebuf[idx] = (idx > first_max && x < -37.43) ? 0.0 : cr_exp(x); }
    }
    __syncthreads();
    HS_PT(3);
    if (tid == 0){
      __builtin_amdgcn_s_setprio(3);                      // 255 threads wait for this chain of additions: it goes first whenever it is ready
      double lsum = 0.0;
      int k = 0;
      for (; k + 8 <= nd; k += 8){
        const double e0 = ebuf[k], e1 = ebuf[k+1], e2 = ebuf[k+2], e3 = ebuf[k+3], e4 = ebuf[k+4], e5 = ebuf[k+5], e6 = ebuf[k+6], e7 = ebuf[k+7];
        lsum += e0; lsum += e1; lsum += e2; lsum += e3; lsum += e4; lsum += e5; lsum += e6; lsum += e7;
      }
      for (; k < nd; k++) lsum += ebuf[k];
      red_v[8] = mx + cr_log(lsum);
      __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
    HS_PT(4);
    const double total = red_v[8];
    double bv = -1.7976931348623157e308; int bi = 0x7fffffff;
#pragma unroll
    for (int k = 0; k < HS_POST_REGS; k++){
      const int idx = tid + 256*k;
      if (idx < nd){
        const double x = v[k] - total;
        post[idx] = x;
        if (x > bv){ bv = x; bi = idx; }
      }
    }
    for (int o = 32; o >= 1; o >>= 1){
      const double ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)){ bv = ov; bi = oi; }
    }
    if (lane == 0){ red_v[16 + w] = bv; red_i[16 + w] = bi; }
    __syncthreads();
    if (tid == 0){
      for (int q = 1; q < 4; q++){ const double ov = red_v[16 + q]; const int oi = red_i[16 + q]; if (ov > bv || (ov == bv && oi < bi)){ bv = ov; bi = oi; } }
      d.sample_total[u.samp_index] = total;
      const bool none = bi == 0x7fffffff;
      d.map_gt[2*u.samp_index]   = none ? -1 : bi / A;
      d.map_gt[2*u.samp_index+1] = none ? -1 : bi % A;
    }
#ifdef HS_POST_TIME
    HS_PT(5);
    if (tid == 0 && (blockIdx.x % 100000) == 777)
      printf("post unit %d nd %d reads %d: accumulate %llu  max+first %llu  exps %llu  sum+log %llu  normalise+map %llu (x10 ns)\n", (int)blockIdx.x, nd, u.n_reads,
             tk[1]-tk[0], tk[2]-tk[1], tk[3]-tk[2], tk[4]-tk[3], tk[5]-tk[4]);
#endif
    return;
  }
  // ---- accumulate (genotyper.cpp:47-61)
  double lmax = -1.0e300;
  if (PHASE != 2){
    const int first = PHASE == 1 ? 256*(int)blockIdx.y + tid : tid, step = PHASE == 1 ? 256*(int)gridDim.y : 256;
    for (int idx = first; idx < nd; idx += step){
      const int a1 = idx / A, a2 = idx - a1*A;
      double v = d.log_prior ? d.log_prior[u.prior_off + idx] : ((a1 == a2) ? u.log_hom_prior : u.log_het_prior);
      for (int r = 0; r < u.n_reads; r++){
        const int g = u.read_begin + r;
        const double* LL = LL0 + (int64_t)r*A;
        const double x = fast_lse2((d.log_half + d.log_p1[g]) + LL[a1], (d.log_half + d.log_p2[g]) + LL[a2], d.log_thresh);
        v += (double)d.read_weight[g] * x;
      }
      post[idx] = v;
      lmax = fmax(lmax, v);
    }
    if (PHASE == 1) return;
  } else {
    for (int idx = tid; idx < nd; idx += 256) lmax = fmax(lmax, post[idx]);
  }
  if (d.raw) return;          // (debug) the host normalises
  // ---- exact log-sum-exp over diplotypes (genotyper.cpp:63-72, mathops.cpp:44-50): max, then total += exp(v - max) in index order
  red_v[tid] = lmax; __syncthreads();
  for (int s = 128; s > 0; s >>= 1){ if (tid < s) red_v[tid] = fmax(red_v[tid], red_v[tid+s]); __syncthreads(); }
  const double mx = red_v[0]; __syncthreads();
  // first index that holds the maximum: from there on the running total is >= 1, and a term below 2^-54 (v - max < -37.43) cannot
  // change it — RN(total + t) = total — so its exponential need not be formed (it is added as +0.0: the same bits).  In front of it every
  // exponential counts (the running total may still be tiny).
  int fi = 0x7fffffff;
  for (int idx = tid; idx < nd; idx += 256) if (post[idx] == mx){ fi = idx; break; }
  red_i[tid] = fi; __syncthreads();
  for (int s = 128; s > 0; s >>= 1){ if (tid < s) red_i[tid] = min(red_i[tid], red_i[tid+s]); __syncthreads(); }
  const int first_max = red_i[0]; __syncthreads();
  double lsum = 0.0;                                      // (thread 0's)
  for (int base = 0; base < nd; base += HS_POST_ECHUNK){
    const int cnt = min(HS_POST_ECHUNK, nd - base);
    for (int k = tid; k < cnt; k += 256){
      const double x = post[base + k] - mx;
      ebuf[k] = (base + k > first_max && x < -37.43) ? 0.0 : cr_exp(x);
    }
    __syncthreads();
    if (tid == 0){
      int k = 0;
      for (; k + 8 <= cnt; k += 8){                       // the loads of a group ahead of its (dependent) additions
        const double e0 = ebuf[k], e1 = ebuf[k+1], e2 = ebuf[k+2], e3 = ebuf[k+3], e4 = ebuf[k+4], e5 = ebuf[k+5], e6 = ebuf[k+6], e7 = ebuf[k+7];
        lsum += e0; lsum += e1; lsum += e2; lsum += e3; lsum += e4; lsum += e5; lsum += e6; lsum += e7;
      }
      for (; k < cnt; k++) lsum += ebuf[k];
    }
    __syncthreads();
  }
  if (tid == 0) red_v[0] = mx + cr_log(lsum);
  __syncthreads();
  const double total = red_v[0]; __syncthreads();
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
    // no diplotype above -DBL_MAX (all -inf / NaN: bad custom priors or NaN likelihoods): the reference's scan leaves its
    // initial pair (-1,-1) (genotyper.cpp:84)
    const bool none = red_i[0] == 0x7fffffff;
    d.map_gt[2*u.samp_index]   = none ? -1 : red_i[0] / A;
    d.map_gt[2*u.samp_index+1] = none ? -1 : red_i[0] % A;
  }
}

extern "C" __global__ void __launch_bounds__(256) hs_posterior_kernel(const hs_post_dev_t* __restrict__ dp){ posterior_body<0>(*dp); }
extern "C" __global__ void __launch_bounds__(256) hs_posterior_accumulate_kernel(const hs_post_dev_t* __restrict__ dp){ posterior_body<1>(*dp); }
extern "C" __global__ void __launch_bounds__(256) hs_posterior_finish_kernel(const hs_post_dev_t* __restrict__ dp){ posterior_body<2>(*dp); }


// Genotyper::extract_genotypes_and_likelihoods (genotyper.cpp:129-251), calc_PLs (99-104), calc_gl_diff (106-127).
// One workgroup per (locus, sample).  A thread owns genotypes (v1, v2) and streams over the haplotype pairs that map to
// them in the reference's scan order (index_1 ascending, then index_2), so update_streaming_log_sum_exp (mathops.cpp:72-80)
// sees the same sequence; exp / log are the correctly rounded ones of cr_math.h: the host libm's bits wherever that is correctly rounded.
extern "C" __global__ void __launch_bounds__(256)
hs_genotype_kernel(const hs_gt_dev_t* __restrict__ dp){
  const hs_gt_dev_t& d = *dp;
  const hs_gt_unit_t u = d.units[blockIdx.x];
  const int A = u.n_alleles, V = u.n_variants, tid = threadIdx.x;
  const double* post = d.log_post + u.post_off;
  double* T = d.tot + u.tot_off;
  const int32_t* h2a = d.h2a + u.map_off;
  const int32_t* gmem = d.gmem + u.map_off;
  const int32_t* goff = d.goff + u.goff_off;
  const double LOG_E_BASE_10 = 0.4342944819;           // mathops.cpp:11
  const double TOLERANCE = 1e-10;                      // mathops.cpp:10
  __shared__ double red_v[256];

  if (!d.tot_given)
  for (int gt = tid; gt < V*V; gt += 256){
    const int v1 = gt / V, v2 = gt - v1*V;
    double mx = -1.7976931348623157e308/2, tot = 0.0;   // -DBL_MAX/2 (genotyper.cpp:152)
    for (int x = goff[v1]; x < goff[v1+1]; x++){
      const double* row = post + (int64_t)gmem[x]*A;
      for (int y = goff[v2]; y < goff[v2+1]; y++){
        const double lv = row[gmem[y]];
        // (once a maximum has been set the total is >= 1: a term below 2^-54 leaves it as it is, bit for bit — its exponential is not formed)
        if (lv <= mx){ const double x = lv - mx; if (!(tot >= 1.0 && x < -37.43)) tot += cr_exp(x); }
        else { tot *= cr_exp(mx - lv); tot += 1.0; mx = lv; }
      }
    }
    T[gt] = mx + cr_log(tot);
  }
  __syncthreads();

  const int s = u.samp_index;
  const int ha = d.map_gt[2*s], hb = d.map_gt[2*s+1];
  if (ha < 0 || hb < 0){      // sample without a MAP diplotype (see hs_posterior_kernel): the reference would index gts (-1,-1); report it
    if (tid == 0){
      d.best_gt[2*s] = -1; d.best_gt[2*s+1] = -1;
      const double nan = __longlong_as_double(0x7ff8000000000000ll);
      d.hap_log_phased[s] = nan; d.hap_log_unphased[s] = nan; d.log_phased[s] = nan; d.log_unphased[s] = nan;
      if (d.calc_any) d.gl_diff[s] = nan;
    }
    return;
  }
  const int ga = h2a[ha], gb = h2a[hb];
  if (tid == 0){
    d.best_gt[2*s] = ga; d.best_gt[2*s+1] = gb;
    const double pab = post[(int64_t)ha*A + hb], pba = post[(int64_t)hb*A + ha];
    d.hap_log_phased[s] = pab;
    d.hap_log_unphased[s] = (ha != hb) ? fast_lse2(pab, pba, d.log_thresh) : pab;
    const double lp = T[V*ga + gb];
    d.log_phased[s] = lp;
    if (ga == gb) d.log_unphased[s] = lp;
    else {
      const double alt = T[V*gb + ga];                 // exact pair log_sum_exp (mathops.cpp:52-57)
      d.log_unphased[s] = (lp > alt) ? lp + cr_log(1 + cr_exp(alt - lp)) : alt + cr_log(1 + cr_exp(lp - alt));
    }
  }
  if (!d.calc_any) return;

  // GLs in VCF order, PHASEDGLs (genotyper.cpp:211-229); kept in the output (or scratch-free registers) for GLDIFF / PL
  const double total_ll = d.sample_total[s];
  const int ngl = u.haploid ? V : V*(V+1)/2;
  double* gls = d.gls + u.gl_off;                     // always allocated when any flag is set (GLDIFF needs them)
  for (int gt = tid; gt < V*V; gt += 256){
    const int i1 = gt / V, i2 = gt - i1*V;
    const double corr = (i1 == i2) ? u.hom_corr : u.het_corr;
    if ((i2 <= i1) && (!u.haploid || i1 == i2)){
      const double gl_e = (total_ll - (corr + u.gl_ncfg)) + fast_lse2(T[gt], T[i2*V + i1], d.log_thresh);
      gls[u.haploid ? i1 : i1*(i1+1)/2 + i2] = gl_e*LOG_E_BASE_10;
    }
    if (d.calc_pgls && (!u.haploid || i1 == i2))
      d.pgls[u.pgl_off + (u.haploid ? i1 : gt)] = ((total_ll - (corr + u.pgl_ncfg)) + T[gt])*LOG_E_BASE_10;
  }
  __syncthreads();
  // max and runner-up GL (calc_gl_diff)
  double m = -1.7976931348623157e308;
  for (int g = tid; g < ngl; g += 256) m = fmax(m, gls[g]);
  red_v[tid] = m; __syncthreads();
  for (int st = 128; st > 0; st >>= 1){ if (tid < st) red_v[tid] = fmax(red_v[tid], red_v[tid+st]); __syncthreads(); }
  const double max_gl = red_v[0]; __syncthreads();
  double m2 = -1.7976931348623157e308;
  for (int g = tid; g < ngl; g += 256) if (gls[g] < max_gl) m2 = fmax(m2, gls[g]);
  red_v[tid] = m2; __syncthreads();
  for (int st = 128; st > 0; st >>= 1){ if (tid < st) red_v[tid] = fmax(red_v[tid], red_v[tid+st]); __syncthreads(); }
  double second_gl = red_v[0];
  if (second_gl == -1.7976931348623157e308) second_gl = max_gl;
  if (tid == 0){
    double diff = -1000;
    if (A != 1){
      const int lo = min(ga, gb), hi = max(ga, gb);
      const int gi = u.haploid ? ga : hi*(hi+1)/2 + lo;
      diff = (fabs(max_gl - gls[gi]) < TOLERANCE) ? (max_gl - second_gl) : gls[gi] - max_gl;
    }
    d.gl_diff[s] = diff;
  }
  if (d.calc_pls)
    for (int g = tid; g < ngl; g += 256) d.pls[u.gl_off + g] = min(999, (int)(-10*(gls[g] - max_gl)));
}


// tests: cr_math.h on the device, element by element (tests/test_cr_math_gpu.py compares with the same header compiled for the host)
extern "C" __global__ void __launch_bounds__(256) hs_cr_math_kernel(int which, const double* __restrict__ x, double* __restrict__ y, int64_t n){
  const int64_t i = (int64_t)blockIdx.x*256 + threadIdx.x;
  if (i < n) y[i] = which ? cr_log(x[i]) : cr_exp(x[i]);
}
