// em.hip — de novo stutter model training on gfx950: EMStutterGenotyper::train (em_stutter_genotyper.cpp:146-226) for a batch
// of loci.
//
// A read is its observed STR size; the "alignment" likelihood of a read given an allele is the stutter pmf, so every O(R·A)
// and O(R·A²) array of the reference is a function of small tables and is recomputed on the fly instead of stored:
//   hs_em_fill        per locus: log_aln_probs[r][a] = log_stutter_pmf(bps[a], bps[obs_r]) (calc_hap_aln_probs, :146-150) and the
//                     allele-frequency diplotype priors, ONE A×A block per locus shared by its samples (:129-144)
//   hs_posterior_kernel  (post_kernels.hip, unchanged arithmetic) = Genotyper::calc_log_sample_posteriors
//   hs_em_mstep       per locus: total LL; recalc_log_gt_priors (:22-57: streaming log-sum-exps, a thread per allele, samples
//                     in order); the seven fast_log_sum_exp reductions of recalc_stutter_model (:64-127) over
//                     (read, allele_1, allele_2, phase), with the read-phase posteriors of :152-169 evaluated in place — the
//                     R×A²×2 array log_read_phase_posteriors_ never exists.  fast_log_sum_exp(vector) is a max and a sum of
//                     float terms, both order-independent, so the reductions are parallel.
// The host keeps the loop: per iteration it reads back 8 doubles per locus (LL + seven totals), forms the new parameters and
// the convergence tests with the host libm exactly as the reference does, and masks converged loci out.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>

#include "../../include/hipstr_hmm.h"
#include "post_layout.h"
#include "prep.h"
#include "api_internal.h"

extern "C" __global__ void hs_posterior_kernel(const hs_post_dev_t* dp);

struct hs_em_locus_t {
  int32_t A, S, R, period, haploid;
  int32_t read_begin;      // into the per-read arrays
  int32_t samp_begin;      // into sample_total
  int32_t bps_off;         // into bps / log_gt_priors (A entries)
  int64_t post_off;        // S*A*A posteriors
  int64_t ll_off;          // R*A log_aln_probs
  int64_t prior_off;       // A*A priors
};

struct hs_em_dev_t {
  const hs_em_locus_t* loci;
  const int32_t* active;       // [n_loci]
  const double*  logp;         // [9*n_loci] in_step, in_nostep, in_up, in_down, out_step, out_nostep, out_up, out_down, log_equal
  const int32_t* bps;          // allele sizes
  const int32_t* obs;          // [n_reads] allele index of the read's size
  const int32_t* sample_label; // [n_reads]
  const double*  log_p1;
  const double*  log_p2;
  double*  gtp;                // log_gt_priors_, same offsets as bps
  double*  ll;                 // log_aln_probs
  double*  prior;
  const double*  post;
  const double*  sample_total;
  const double*  int_log;
  double*  new_ll;             // [n_loci]
  double*  sums;               // [7*n_loci] in_up, in_down, in_eq, in_diffs, out_up, out_down, out_diffs
  double*  row_lse;            // scratch: log_sum_exp of every posterior row (s, allele_1); a locus uses [post_off, post_off + S*A)
  int32_t* cat;                // [R*A per locus, at ll_off] what a read of this size says about the stutter model if it came from this allele:
                               //   category (0 in_up, 1 in_down, 2 in_eq, 4 out_up, 5 out_down) | diffs vector (3 in, 6 out, 255 none) << 8
  double*  leff;               // same layout: ln |effective difference| (the addend of the diffs vectors), 0 where there is none
  double*  part;               // [n_loci][HS_EM_PARTS][7] partial maxima, then partial sums, of the seven M-step vectors
  double   log_thresh, log_half, log_1p1;
};
#define HS_EM_PARTS 8          // workgroups per locus in the two big M-step reductions

namespace {

__device__ __forceinline__ float e_fasterexp(float p){           // fastonebigheader.h:206-218
  const float y = __fmul_rn(1.442695040f, p);
  const float c = (y < -126.0f) ? -126.0f : y;
  return __uint_as_float((uint32_t)__fmul_rn(8388608.0f, __fadd_rn(c, 126.94269504f)));
}
__device__ __forceinline__ float e_fasterlog(float x){           // fastonebigheader.h:348-358
  float y = (float)__float_as_uint(x);
  y = __fmul_rn(y, 8.2629582881927490e-8f);
  return __fsub_rn(y, 87.989971088f);
}
__device__ __forceinline__ float e_fastpow2(float p){            // fastonebigheader.h:188-198
  const float offset = (p < 0.0f) ? 1.0f : 0.0f;
  const float clipp = (p < -126.0f) ? -126.0f : p;
  const int w = (int)clipp;
  const float z = __fadd_rn(__fsub_rn(clipp, (float)w), offset);
  const float t = __fsub_rn(__fadd_rn(__fadd_rn(clipp, 121.2740575f), __fdiv_rn(27.7280233f, __fsub_rn(4.84252568f, z))), __fmul_rn(1.49012907f, z));
  return __uint_as_float((uint32_t)__fmul_rn(8388608.0f, t));
}
__device__ __forceinline__ float e_fastlog(float x){             // fastonebigheader.h:320-338
  const uint32_t vi = __float_as_uint(x);
  const float mx = __uint_as_float((vi & 0x007FFFFFu) | 0x3f000000u);
  float y = (float)vi;
  y = __fmul_rn(y, 1.1920928955078125e-7f);
  const float l2 = __fsub_rn(__fsub_rn(__fsub_rn(y, 124.22551499f), __fmul_rn(1.498030302f, mx)),
                             __fdiv_rn(1.72587999f, __fadd_rn(0.3520887068f, mx)));
  return __fmul_rn(0.69314718f, l2);
}
__device__ __forceinline__ double e_fast_lse2(double a, double b, double thr){    // mathops.cpp:86-95
  const double hi = a > b ? a : b, lo = a > b ? b : a;
  const double diff = lo - hi;
  return diff < thr ? hi : hi + (double)e_fastlog(__fadd_rn(1.0f, e_fastpow2(__fmul_rn(1.442695040f, (float)diff))));
}

// StutterModel::log_stutter_pmf (stutter_model.cpp:29-53) from the nine logs the constructor keeps (stutter_model.h:44-58)
__device__ __forceinline__ double em_pmf(const double* lp, int period, int sample_bps, int read_bps){
  const int diff = read_bps - sample_bps;
  if (diff % period != 0){
    const int eff = diff - diff/period;
    return eff < 0 ? (lp[7] + lp[5]) + lp[4]*(double)(-eff-1) : (lp[6] + lp[5]) + lp[4]*(double)(eff-1);
  }
  const int rep = diff/period;
  if (rep == 0) return lp[8];
  return rep < 0 ? (lp[3] + lp[1]) + lp[0]*(double)(-rep-1) : (lp[2] + lp[1]) + lp[0]*(double)(rep-1);
}

__global__ void __launch_bounds__(256) hs_em_fill(const hs_em_dev_t* __restrict__ dp){
  const hs_em_dev_t& d = *dp;
  const int l = blockIdx.x;
  if (!d.active[l]) return;
  const hs_em_locus_t L = d.loci[l];
  const double* lp = d.logp + 9*l;
  const int32_t* bps = d.bps + L.bps_off;
  const double* gtp = d.gtp + L.bps_off;
  const int A = L.A;
  for (int x = threadIdx.x; x < L.R*A; x += 256){
    const int r = x / A, a = x - r*A;
    const int ob = bps[d.obs[L.read_begin + r]];
    d.ll[L.ll_off + x] = em_pmf(lp, L.period, bps[a], ob);
    // recalc_stutter_model's bookkeeping for (read, source allele) (:76-104): it does not depend on the iteration, but it is cheap
    // next to the pmf and keeps the M-step free of integer divisions
    const int bd = ob - bps[a];
    int cat = 2, dcat = 255; double le = 0.0;
    if (bd != 0){
      if (bd % L.period != 0){ const int eff = bd - bd/L.period; cat = bd > 0 ? 4 : 5; dcat = 6; le = d.int_log[abs(eff)]; }
      else { const int eff = bd/L.period; cat = bd > 0 ? 0 : 1; dcat = 3; le = d.int_log[abs(eff)]; }
    }
    d.cat[L.ll_off + x] = cat | (dcat << 8);
    d.leff[L.ll_off + x] = le;
  }
  for (int x = threadIdx.x; x < A*A; x += 256){          // EMStutterGenotyper::init_log_sample_priors (:129-144)
    const int i1 = x / A, i2 = x - i1*A;
    d.prior[L.prior_off + x] = !L.haploid ? gtp[i1] + gtp[i2] : (i1 == i2 ? gtp[i1] : -DBL_MAX/2);
  }
}

// block reductions over 256 threads
__device__ __forceinline__ double block_max(double v, double* red){
  const int tid = threadIdx.x;
  red[tid] = v; __syncthreads();
  for (int s = 128; s > 0; s >>= 1){ if (tid < s) red[tid] = fmax(red[tid], red[tid+s]); __syncthreads(); }
  const double out = red[0]; __syncthreads();
  return out;
}
__device__ __forceinline__ double block_sum(double v, double* red){
  const int tid = threadIdx.x;
  red[tid] = v; __syncthreads();
  for (int s = 128; s > 0; s >>= 1){ if (tid < s) red[tid] += red[tid+s]; __syncthreads(); }
  const double out = red[0]; __syncthreads();
  return out;
}

// recalc_log_gt_priors (:22-57) of one locus by one workgroup: thread a owns allele a; the two scans in the reference's order.
// Independent of the stutter reductions, so it rides in their first launch as one more slice (blockIdx.y == HS_EM_PARTS).
__device__ void em_gt_priors(const hs_em_dev_t& d, const hs_em_locus_t& L, int tid){
  const int A = L.A, S = L.S;
  const double* post = d.post + L.post_off;
  double* gtp = d.gtp + L.bps_off;
  // log_sum_exp of every row (sample, allele_1) first, all rows in parallel (each row summed in allele_2 order as the reference does);
  // the streaming scans below are sequential per allele by definition
  double* row_lse = d.row_lse + L.post_off;     // S*A values in this locus' own S*A*A region: disjoint between loci whatever their A
  for (int x = tid; x < S*A; x += 256){
    const double* row = post + (int64_t)x*A;
    double rm = row[0];
    for (int j = 1; j < A; j++) rm = fmax(rm, row[j]);
    double rs = 0.0;
    for (int j = 0; j < A; j++) rs += exp(row[j] - rm);
    row_lse[x] = rm + log(rs);
  }
  __syncthreads();
  // (the scans are one lane per allele; a wavefront without alleles walks through to the barrier below — every thread of the workgroup
  // reaches every barrier)
  // The scans are one dependent chain per allele; the values they eat are fetched eight at a time ahead of the chain, or every step
  // would wait out a trip to L2.
#define EM_SCAN_STEP(lv) do { if ((lv) <= m) t += exp((lv) - m); else { t *= exp(m - (lv)); t += 1.0; m = (lv); } } while (0)
  for (int a = tid; a < A; a += 256){
    double m = -DBL_MAX/2, t = 0.0;
    int s = 0;
    for (; s + 8 <= S; s += 8){                          // first allele of the diplotype: log_sum_exp of row (s, a)
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = row_lse[(int64_t)(s + q)*A + a];
#pragma unroll
      for (int q = 0; q < 8; q++) EM_SCAN_STEP(v[q]);
    }
    for (; s < S; s++){ const double lv = row_lse[(int64_t)s*A + a]; EM_SCAN_STEP(lv); }
    const int64_t n2 = (int64_t)S*A;                     // second allele: (s, i1) in order is one run of stride A from post[a]
    int64_t y = 0;
    for (; y + 8 <= n2; y += 8){
      double v[8];
#pragma unroll
      for (int q = 0; q < 8; q++) v[q] = post[(y + q)*A + a];
#pragma unroll
      for (int q = 0; q < 8; q++) EM_SCAN_STEP(v[q]);
    }
    for (; y < n2; y++){ const double lv = post[y*A + a]; EM_SCAN_STEP(lv); }
    gtp[a] = m + log(t);
  }
#undef EM_SCAN_STEP
  __syncthreads();
  if (tid == 0){                                          // normalise: exact log_sum_exp in allele order
    double m = gtp[0];
    for (int a = 1; a < A; a++) m = fmax(m, gtp[a]);
    double t = 0.0;
    for (int a = 0; a < A; a++) t += exp(gtp[a] - m);
    const double lt = m + log(t);
    for (int a = 0; a < A; a++) gtp[a] -= lt;
  }
}

// recalc_stutter_model (:64-127): seven log-sum-exps over factor = log P(diplotype | sample) + log P(phase | read, diplotype), one term per
// (read, diplotype, phase) — R x A^2 x 2 of them, the only large loop of the EM.  fast_log_sum_exp needs the maximum first, so the loop
// runs twice, as two kernels; each locus is cut into HS_EM_PARTS slices (one workgroup each) whose partial maxima / partial sums are
// combined afterwards: a maximum is order-independent, and the sums are sums of float terms taken in double, exact in any order.
//   PASS 0: part[l][k][0..6] = maxima of slice k;  hs_em_mstep_keepmax: keep[l][0..6] = their maximum (with the pseudocount entries, :69-71);
//   PASS 1: part[l][k][0..6] = sums of slice k relative to keep[l]
template <int PASS>
__global__ void __launch_bounds__(256) hs_em_mstep_part(const hs_em_dev_t* __restrict__ dp, const double* __restrict__ keep){
  const hs_em_dev_t& d = *dp;
  const int l = blockIdx.x, k_part = blockIdx.y, tid = threadIdx.x;
  if (!d.active[l]) return;
  const hs_em_locus_t L = d.loci[l];
  const int A = L.A, nd = A*A;
  const double* post = d.post + L.post_off;
  const double* ll = d.ll + L.ll_off;
  const int32_t* catv = d.cat + L.ll_off;
  const double* leff = d.leff + L.ll_off;
  double* part = d.part + ((size_t)l*HS_EM_PARTS)*7;
  __shared__ double red[256];
  double mx[7];
  if (PASS == 1) for (int k = 0; k < 7; k++) mx[k] = keep[7*l + k];      // the maxima over all slices and the pseudocount entries (hs_em_mstep_keepmax)
  double acc[7];
  for (int k = 0; k < 7; k++) acc[k] = (PASS == 0) ? ((k == 3 || k == 6) ? d.log_1p1 : 0.0) : 0.0;
  // A thread owns (read r, source allele a) and walks the other allele j: both phases' terms with source a — phase 0 of diplotype
  // (a, j) and phase 1 of (j, a) — feed the same two vectors (the category and diffs vector of (r, a), hs_em_fill), so the
  // seven-way choice is made once per (r, a) and the walk itself is branch-free.
  const int total = L.R*A;
  const int x0 = (int)((int64_t)total*k_part/HS_EM_PARTS), x1 = (int)((int64_t)total*(k_part + 1)/HS_EM_PARTS);
  for (int x = x0 + tid; x < x1; x += 256){
    const int r = x / A, a = x - r*A;
    const int g = L.read_begin + r;
    // recalc_log_read_phase_posteriors (:152-169); the pmf values are the E-step's log_aln_probs
    const double b1 = d.log_half + d.log_p1[g], b2 = d.log_half + d.log_p2[g];
    const double one_a = b1 + ll[x], two_a = b2 + ll[x];
    const double* gp = post + (int64_t)d.sample_label[g]*nd;
    const double* llr = ll + r*A;
    const int c2 = catv[x];
    const int cat = c2 & 0xff, dcat = c2 >> 8;
    const double le = leff[x];
    const bool same = (b1 == b2);                         // no phasing information: the two mixtures of a pair are the same number
    if (PASS == 0){
      double m = -DBL_MAX;
      for (int j = 0; j < A; j++){
        const double lj = llr[j];
        const double both0 = e_fast_lse2(one_a, b2 + lj, d.log_thresh);                       // diplotype (a, j)
        const double both1 = same ? both0 : e_fast_lse2(b1 + lj, two_a, d.log_thresh);        // diplotype (j, a)
        m = fmax(m, fmax(gp[a*A + j] + (one_a - both0), gp[j*A + a] + (two_a - both1)));
      }
      const double md = m + le;                           // adding one number keeps the order: max(f) + le == max(f + le)
#pragma unroll
      for (int k = 0; k < 7; k++){
        if (k == cat) acc[k] = fmax(acc[k], m);
        if (k == dcat) acc[k] = fmax(acc[k], md);
      }
    } else {
      double mc = 0.0, md = 0.0;
#pragma unroll
      for (int k = 0; k < 7; k++){ if (k == cat) mc = mx[k]; if (k == dcat) md = mx[k]; }
      const bool has_d = dcat < 7;
      double sc = 0.0, sd = 0.0;
      for (int j = 0; j < A; j++){
        const double lj = llr[j];
        const double both0 = e_fast_lse2(one_a, b2 + lj, d.log_thresh);
        const double both1 = same ? both0 : e_fast_lse2(b1 + lj, two_a, d.log_thresh);
        const double f0 = gp[a*A + j] + (one_a - both0), f1 = gp[j*A + a] + (two_a - both1);
        { const double df = f0 - mc; if (df > d.log_thresh) sc += (double)e_fasterexp((float)df); }
        { const double df = f1 - mc; if (df > d.log_thresh) sc += (double)e_fasterexp((float)df); }
        if (has_d){
          { const double df = (f0 + le) - md; if (df > d.log_thresh) sd += (double)e_fasterexp((float)df); }
          { const double df = (f1 + le) - md; if (df > d.log_thresh) sd += (double)e_fasterexp((float)df); }
        }
      }
#pragma unroll
      for (int k = 0; k < 7; k++){
        if (k == cat) acc[k] += sc;
        if (k == dcat) acc[k] += sd;
      }
    }
  }
  double out[7];
  for (int k = 0; k < 7; k++) out[k] = (PASS == 0) ? block_max(acc[k], red) : block_sum(acc[k], red);
  if (tid == 0) for (int k = 0; k < 7; k++) part[k_part*7 + k] = out[k];
}

// recalc_log_gt_priors of every active locus: long dependent scans on a few lanes.  Nothing in the stutter reductions needs their result
// (the next iteration's hs_em_fill does), so the kernel runs beside them on a second stream.
__global__ void __launch_bounds__(256) hs_em_gt_priors(const hs_em_dev_t* __restrict__ dp){
  const hs_em_dev_t& d = *dp;
  const int l = blockIdx.x;
  if (!d.active[l]) return;
  const hs_em_locus_t L = d.loci[l];
  em_gt_priors(d, L, threadIdx.x);
}

__global__ void __launch_bounds__(64) hs_em_mstep_keepmax(const hs_em_dev_t* __restrict__ dp, double* __restrict__ keep){
  const hs_em_dev_t& d = *dp;
  const int l = blockIdx.x;
  if (!d.active[l] || threadIdx.x >= 7) return;
  const double* part = d.part + ((size_t)l*HS_EM_PARTS)*7;
  double m = (threadIdx.x == 3 || threadIdx.x == 6) ? d.log_1p1 : 0.0;
  for (int q = 0; q < HS_EM_PARTS; q++) m = fmax(m, part[q*7 + threadIdx.x]);
  keep[7*l + threadIdx.x] = m;
}

__global__ void __launch_bounds__(256) hs_em_mstep(const hs_em_dev_t* __restrict__ dp, const double* __restrict__ keep){
  const hs_em_dev_t& d = *dp;
  const int l = blockIdx.x, tid = threadIdx.x;
  if (!d.active[l]) return;
  const hs_em_locus_t L = d.loci[l];
  const int S = L.S;

  // total log-likelihood of the E-step: sum of the sample totals in sample order (genotyper.cpp:75)
  if (tid == 0){
    double t = 0.0;
    for (int s = 0; s < S; s++) t += d.sample_total[L.samp_begin + s];
    d.new_ll[l] = t;
  }
  if (tid == 0){
    const double* part = d.part + ((size_t)l*HS_EM_PARTS)*7;
    for (int k = 0; k < 7; k++){
      const double mxk = keep[7*l + k];
      double t = 0.0;
      for (int q = 0; q < HS_EM_PARTS; q++) t += part[q*7 + k];
      // the pseudocount entries: 0.0 in every vector, ln 1.1 in the two diffs vectors
      { const double df = 0.0 - mxk; if (df > d.log_thresh) t += (double)e_fasterexp((float)df); }
      if (k == 3 || k == 6){ const double df = d.log_1p1 - mxk; if (df > d.log_thresh) t += (double)e_fasterexp((float)df); }
      d.sums[7*l + k] = mxk + (double)e_fasterlog((float)t);
    }
  }
}

// ---- host side -------------------------------------------------------------------------------------------------
#define EM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess){ \
  hipstr::api_fail(std::string(#call) + ": " + hipGetErrorString(e_)); return 1; } } while (0)

struct EmBufs {
  std::vector<void*> p;
  hipstr::Ctx* ctx = NULL;          // blocks come from (and return to) the context's cache: no hipMalloc / hipFree per call
  // the blocks go back to a cache other host threads draw from: nothing launched by this call may still be running (error paths leave early)
  hipStream_t stream = NULL;        // the stream the call's kernels run on
  ~EmBufs(){ if (ctx){ if (stream) hipStreamSynchronize(stream); for (void* x : p) hipstr::dev_free(ctx, x); } }
  template <typename T> int alloc(T** out, size_t count){
    *out = NULL;
    if (!ctx) ctx = hipstr::api_current_ctx();
    if (!ctx) return 1;
    *out = (T*)hipstr::dev_alloc(ctx, (count ? count : 1)*sizeof(T));
    if (!*out) return 1;
    p.push_back(*out);
    return 0;
  }
  template <typename T> int put(T** out, const T* src, size_t count){
    if (alloc(out, count)) return 1;
    if (count) EM_HIP(hipMemcpy(*out, src, count*sizeof(T), hipMemcpyHostToDevice));
    return 0;
  }
};

// host copies of the reference's float approximations (mathops.cpp:86-95, fastonebigheader.h:188-204, 320-338)
inline float h_bits(uint32_t u){ float f; memcpy(&f, &u, 4); return f; }
inline uint32_t h_ubits(float f){ uint32_t u; memcpy(&u, &f, 4); return u; }
float h_fastpow2(float p){
  const float offset = (p < 0) ? 1.0f : 0.0f;
  const float clipp = (p < -126) ? -126.0f : p;
  const int w = (int)clipp;
  const float z = clipp - w + offset;
  return h_bits((uint32_t)((1 << 23) * (clipp + 121.2740575f + 27.7280233f / (4.84252568f - z) - 1.49012907f * z)));
}
float h_fastlog(float x){
  const uint32_t vi = h_ubits(x);
  const float mx = h_bits((vi & 0x007FFFFF) | 0x3f000000);
  float y = (float)vi;
  y *= 1.1920928955078125e-7f;
  return 0.69314718f * (y - 124.22551499f - 1.498030302f * mx - 1.72587999f / (0.3520887068f + mx));
}
double h_fast_lse2(double a, double b, double thr){
  const double hi = a > b ? a : b, lo = a > b ? b : a;
  const double diff = lo - hi;
  return diff < thr ? hi : hi + h_fastlog(1 + h_fastpow2(1.442695040f * (float)diff));
}
double h_lse2(double a, double b){ return a > b ? a + log(1 + exp(b - a)) : b + log(1 + exp(a - b)); }       // mathops.cpp:52-57

}  // namespace

extern "C" int hipstr_em_train(const hipstr_em_batch_t* eb, uint8_t* trained, double* stutter, int32_t* n_iter, double* final_ll){
  hipstr::ApiTimer prof_t(hipstr::PB_EM_TRAIN);
  using hipstr::api_fail;
  if (!eb || !trained || !stutter || !n_iter || !final_ll) return api_fail("null argument");
  const int nl = eb->n_loci;
  if (nl < 0) return api_fail("negative locus count");
  if (nl == 0) return 0;
  hipstr::ApiTables T;
  if (hipstr::api_device_tables(&T)) return 1;
  const hipstr::HostTables& HT = hipstr::host_tables();
  const int n_reads = eb->read_off[nl];
  const bool timing = getenv("HIPSTR_TIMING") != NULL;
  auto t_prev = std::chrono::steady_clock::now();
  double t_gpu = 0.0, t_host = 0.0;
  auto lap = [&](const char* what, double* into){
    const auto now = std::chrono::steady_clock::now();
    const double dt = std::chrono::duration<double>(now - t_prev).count();
    t_prev = now;
    if (into) *into += dt; else if (timing) fprintf(stderr, "hipstr_em_train: %s %.3f ms\n", what, 1e3*dt);
  };

  // ---- alleles, read -> allele index, initial allele frequencies (em_stutter_genotyper.h:55-100, .cpp:10-20)
  std::vector<hs_em_locus_t> loci(nl);
  std::vector<hs_post_unit_t> units;
  std::vector<int32_t> bps, obs(n_reads), unit_locus;
  std::vector<double> gtp;
  // per locus, independent of the others (host threads): allele sizes, the reads' allele indices, the initial allele frequencies
  struct LocusPrep { std::vector<int> sizes; std::vector<double> gtp; std::vector<int32_t> reads_of_sample; const char* err = NULL; };
  std::vector<LocusPrep> lp(nl);
  hipstr::parallel_for(nl, nl >= 64 ? hipstr::host_threads() : 1, [&](int l){
    LocusPrep& Q = lp[l];
    const int S = eb->n_samples[l], r0 = eb->read_off[l], r1 = eb->read_off[l+1];
    if (eb->period[l] < 1 || eb->period[l] > 9){ Q.err = "STR period must be in [1,9] (stutter_model.h:38)"; return; }
    if (S < 1){ Q.err = "locus without samples"; return; }
    std::vector<int>& sizes = Q.sizes;
    for (int r = r0; r < r1; r++) if (eb->num_bps[r] != eb->ref_allele) sizes.push_back(eb->num_bps[r]);
    std::sort(sizes.begin(), sizes.end());
    sizes.erase(std::unique(sizes.begin(), sizes.end()), sizes.end());
    sizes.insert(sizes.begin(), eb->ref_allele);
    const int A = (int)sizes.size();
    if (A + 1 >= 10000){ Q.err = "too many distinct allele sizes"; return; }
    Q.reads_of_sample.assign(S, 0);
    int prev = 0;
    for (int r = r0; r < r1; r++){
      const int s = eb->sample_label[r];
      if (s < prev || s >= S){ Q.err = "reads of a locus must be grouped by ascending sample label (genotyper.h:112-119)"; return; }
      prev = s; Q.reads_of_sample[s]++;
      obs[r] = (int32_t)(std::lower_bound(sizes.begin() + 1, sizes.end(), eb->num_bps[r]) - sizes.begin());
      if (eb->num_bps[r] == eb->ref_allele) obs[r] = 0;
    }
    std::vector<double> g(A, 1.0);                                     // init_log_gt_priors (:10-20)
    for (int r = r0; r < r1; r++) g[obs[r]] += 1.0/Q.reads_of_sample[eb->sample_label[r]];
    double tot = 0.0; for (int a = 0; a < A; a++) tot += g[a];
    const double lt = log(tot);
    Q.gtp.resize(A);
    for (int a = 0; a < A; a++) Q.gtp[a] = log(g[a]) - lt;
  });
  int64_t post_off = 0, ll_off = 0, prior_off = 0; int samp_off = 0;
  for (int l = 0; l < nl; l++){
    const LocusPrep& Q = lp[l];
    if (Q.err) return api_fail(Q.err);
    const int S = eb->n_samples[l], r0 = eb->read_off[l], r1 = eb->read_off[l+1], R = r1 - r0;
    const int A = (int)Q.sizes.size();
    hs_em_locus_t& L = loci[l];
    memset(&L, 0, sizeof L);
    L.A = A; L.S = S; L.R = R; L.period = eb->period[l]; L.haploid = (eb->haploid && eb->haploid[l]) ? 1 : 0;
    L.read_begin = r0; L.samp_begin = samp_off; L.bps_off = (int32_t)bps.size();
    L.post_off = post_off; L.ll_off = ll_off; L.prior_off = prior_off;
    gtp.insert(gtp.end(), Q.gtp.begin(), Q.gtp.end());
    bps.insert(bps.end(), Q.sizes.begin(), Q.sizes.end());
    int r = r0;
    for (int s = 0; s < S; s++){                                       // posterior-kernel units: (locus, sample)
      hs_post_unit_t u; memset(&u, 0, sizeof u);
      u.post_off = post_off + (int64_t)s*A*A; u.prior_off = prior_off; u.n_alleles = A; u.samp_index = samp_off + s;
      u.read_begin = r; u.ll_off = ll_off + (int64_t)(r - r0)*A;
      r += Q.reads_of_sample[s];
      u.n_reads = r - u.read_begin;
      units.push_back(u); unit_locus.push_back(l);
    }
    post_off += (int64_t)S*A*A; ll_off += (int64_t)R*A; prior_off += (int64_t)A*A; samp_off += S;
  }
  std::vector<LocusPrep>().swap(lp);
  lap("alleles and units", NULL);
  // ---- device state
  EmBufs dev;
  dev.stream = T.stream;
  hs_em_dev_t h; memset(&h, 0, sizeof h);
  hs_post_dev_t ph; memset(&ph, 0, sizeof ph);
  hs_em_locus_t* d_loci; hs_post_unit_t* d_units; int32_t *d_active, *d_unit_active, *d_bps, *d_obs, *d_lab, *d_w, *d_mapgt;
  double *d_logp, *d_p1, *d_p2, *d_gtp, *d_ll, *d_prior, *d_post, *d_tot, *d_newll, *d_sums, *d_rowlse, *d_leff, *d_part, *d_keep; int32_t* d_cat;
  std::vector<int32_t> ones(n_reads, 1);
  if (dev.put(&d_loci, loci.data(), loci.size()) || dev.put(&d_units, units.data(), units.size()) || dev.alloc(&d_active, nl) ||
      dev.alloc(&d_unit_active, units.size()) || dev.put(&d_bps, bps.data(), bps.size()) || dev.put(&d_obs, obs.data(), obs.size()) ||
      dev.put(&d_lab, eb->sample_label, n_reads) || dev.put(&d_w, ones.data(), ones.size()) || dev.alloc(&d_mapgt, 2*(size_t)samp_off) ||
      dev.alloc(&d_logp, 9*(size_t)nl) || dev.put(&d_p1, eb->log_p1, n_reads) || dev.put(&d_p2, eb->log_p2, n_reads) ||
      dev.put(&d_gtp, gtp.data(), gtp.size()) || dev.alloc(&d_ll, ll_off) || dev.alloc(&d_prior, prior_off) || dev.alloc(&d_post, post_off) ||
      dev.alloc(&d_tot, samp_off) || dev.alloc(&d_newll, nl) || dev.alloc(&d_sums, 7*(size_t)nl) || dev.alloc(&d_rowlse, post_off) ||
      dev.alloc(&d_cat, ll_off) || dev.alloc(&d_leff, ll_off) || dev.alloc(&d_part, 7*(size_t)HS_EM_PARTS*nl) || dev.alloc(&d_keep, 7*(size_t)nl)) return 1;
  h.loci = d_loci; h.active = d_active; h.logp = d_logp; h.bps = d_bps; h.obs = d_obs; h.sample_label = d_lab; h.log_p1 = d_p1; h.log_p2 = d_p2;
  h.gtp = d_gtp; h.ll = d_ll; h.prior = d_prior; h.post = d_post; h.sample_total = d_tot; h.int_log = T.int_log; h.new_ll = d_newll; h.sums = d_sums; h.row_lse = d_rowlse; h.cat = d_cat; h.leff = d_leff; h.part = d_part;
  h.log_thresh = HT.log_thresh; h.log_half = HT.log_half; h.log_1p1 = log(1.1);
  ph.units = d_units; ph.log_aln_probs = d_ll; ph.log_p1 = d_p1; ph.log_p2 = d_p2; ph.read_weight = d_w; ph.log_prior = d_prior;
  ph.unit_active = d_unit_active; ph.log_post = d_post; ph.sample_total = d_tot; ph.map_gt = d_mapgt;
  ph.log_thresh = HT.log_thresh; ph.log_half = HT.log_half;
  hs_em_dev_t* d_h; hs_post_dev_t* d_ph;
  if (dev.put(&d_h, &h, 1) || dev.put(&d_ph, &ph, 1)) return 1;

  lap("device state", NULL);
  struct SideStream {                 // a second stream for the allele-frequency scans of an iteration
    hipStream_t stream = NULL; hipEvent_t ev_fork = NULL, ev_join = NULL;
    ~SideStream(){ if (stream) hipStreamSynchronize(stream); if (ev_fork) hipEventDestroy(ev_fork); if (ev_join) hipEventDestroy(ev_join); if (stream) hipStreamDestroy(stream); }
  } side;
  EM_HIP(hipStreamCreateWithFlags(&side.stream, hipStreamNonBlocking));
  EM_HIP(hipEventCreateWithFlags(&side.ev_fork, hipEventDisableTiming));
  EM_HIP(hipEventCreateWithFlags(&side.ev_join, hipEventDisableTiming));
  // ---- the EM loop of train() (:171-226), all loci in lock step, converged loci masked out
  struct State { double sp[6]; double LL; int it; bool done, ok; };
  std::vector<State> st(nl);
  for (int l = 0; l < nl; l++){
    const double init[6] = { 0.9, 0.1, 0.1, 0.8, 0.01, 0.01 };        // init_stutter_model (:59-62)
    memcpy(st[l].sp, init, sizeof init);
    st[l].LL = -DBL_MAX; st[l].it = 1; st[l].done = false; st[l].ok = false;
    n_iter[l] = 0; final_ll[l] = 0; trained[l] = 0;
  }
  std::vector<int32_t> active(nl), unit_active(units.size());
  std::vector<double> logp(9*(size_t)nl), newll(nl), sums(7*(size_t)nl);
  for (;;){
    int n_active = 0;
    for (int l = 0; l < nl; l++){
      State& s = st[l];
      if (!s.done && s.it > eb->max_iter){ s.done = true; s.ok = false; }      // ran out of iterations: train() returns false
      active[l] = s.done ? 0 : 1; n_active += active[l];
      const double* sp = s.sp;                                                 // StutterModel constructor (stutter_model.h:44-58)
      double* q = &logp[9*(size_t)l];
      q[0] = log(1-sp[0]); q[1] = log(sp[0]); q[2] = log(sp[1]); q[3] = log(sp[2]);
      q[4] = log(1-sp[3]); q[5] = log(sp[3]); q[6] = log(sp[4]); q[7] = log(sp[5]);
      q[8] = log(1-sp[1]-sp[2]-sp[4]-sp[5]);
    }
    if (n_active == 0) break;
    for (size_t u = 0; u < units.size(); u++) unit_active[u] = active[unit_locus[u]];
    lap("", &t_host);
    EM_HIP(hipMemcpy(d_active, active.data(), nl*sizeof(int32_t), hipMemcpyHostToDevice));
    EM_HIP(hipMemcpy(d_unit_active, unit_active.data(), units.size()*sizeof(int32_t), hipMemcpyHostToDevice));
    EM_HIP(hipMemcpy(d_logp, logp.data(), logp.size()*sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(hs_em_fill, dim3(nl), dim3(256), 0, T.stream, d_h);
    hipLaunchKernelGGL(hs_posterior_kernel, dim3((unsigned)units.size()), dim3(256), 0, T.stream, (const hs_post_dev_t*)d_ph);
    EM_HIP(hipEventRecord(side.ev_fork, T.stream));                      // posteriors are in place: the allele-frequency scans branch off
    EM_HIP(hipStreamWaitEvent(side.stream, side.ev_fork, 0));
    hipLaunchKernelGGL(hs_em_gt_priors, dim3(nl), dim3(256), 0, side.stream, d_h);
    EM_HIP(hipEventRecord(side.ev_join, side.stream));
    hipLaunchKernelGGL(hs_em_mstep_part<0>, dim3(nl, HS_EM_PARTS), dim3(256), 0, T.stream, d_h, (const double*)NULL);
    hipLaunchKernelGGL(hs_em_mstep_keepmax, dim3(nl), dim3(64), 0, T.stream, d_h, d_keep);
    hipLaunchKernelGGL(hs_em_mstep_part<1>, dim3(nl, HS_EM_PARTS), dim3(256), 0, T.stream, d_h, (const double*)d_keep);
    hipLaunchKernelGGL(hs_em_mstep, dim3(nl), dim3(256), 0, T.stream, d_h, (const double*)d_keep);
    EM_HIP(hipStreamWaitEvent(T.stream, side.ev_join, 0));                // ... and join before the host reads this iteration's results
    EM_HIP(hipGetLastError());
    EM_HIP(hipstr::wait_stream(T.stream));
    EM_HIP(hipMemcpy(newll.data(), d_newll, nl*sizeof(double), hipMemcpyDeviceToHost));
    EM_HIP(hipMemcpy(sums.data(), d_sums, sums.size()*sizeof(double), hipMemcpyDeviceToHost));
    lap("", &t_gpu);
    for (int l = 0; l < nl; l++){
      State& s = st[l];
      if (!active[l]) continue;
      const double new_LL = newll[l];
      n_iter[l] = s.it; final_ll[l] = new_LL;
      if (new_LL < s.LL + 1e-10){ s.done = true; s.ok = true; continue; }      // :190-194 (TOLERANCE = 1e-10)
      const double* t = &sums[7*(size_t)l];                                    // in_up, in_down, in_eq, in_diffs, out_up, out_down, out_diffs
      const double out_total = h_fast_lse2(t[4], t[5], HT.log_thresh);
      const double in_pgeom  = std::min(0.999, exp(h_lse2(t[0], t[1]) - t[3]));
      const double out_pgeom = std::min(0.999, exp(out_total - t[6]));
      const double m3 = std::max(std::max(t[0], t[1]), t[2]);
      const double lse3 = m3 + log(exp(t[0]-m3) + exp(t[1]-m3) + exp(t[2]-m3));          // mathops.cpp:59-62
      const double log_total = h_lse2(lse3, out_total);
      const double nw[6] = { in_pgeom, exp(t[0] - log_total), exp(t[1] - log_total), out_pgeom, exp(t[4] - log_total), exp(t[5] - log_total) };
      const double abs_change = new_LL - s.LL, frac_change = -(new_LL - s.LL)/s.LL;
      bool conv = false;
      if (abs_change < eb->min_ll_abs_change && frac_change < eb->min_ll_frac_change) conv = true;
      else {
        conv = true;
        for (int k = 0; k < 6; k++) if (!(fabs(s.sp[k] - nw[k]) < 0.0001)) conv = false;      // parameters_within_threshold (stutter_model.h:62-65)
      }
      memcpy(s.sp, nw, sizeof nw);
      if (conv){ s.done = true; s.ok = true; continue; }
      s.LL = new_LL; s.it++;
    }
  }
  if (timing) fprintf(stderr, "hipstr_em_train: iterations: copies + kernels %.3f ms, host updates %.3f ms\n", 1e3*t_gpu, 1e3*t_host);
  for (int l = 0; l < nl; l++){
    trained[l] = st[l].ok ? 1 : 0;
    memcpy(stutter + 6*(size_t)l, st[l].sp, 6*sizeof(double));
  }
  return 0;
}
