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
// Round 5: the iteration loop is device-resident.  hs_em_finish forms the new parameters and train()'s convergence tests on the device
// (the host did, with three synchronous copies in and two out per lock-step round), with the CORRECTLY ROUNDED exp / log of cr_math.h in
// the reference's operation order — the host libm's bits wherever that is itself correctly rounded, so iteration counts and parameters
// are the CPU run's; hs_em_compact / hs_em_units rebuild the list of loci (and of their (locus, sample) posterior units) that are still
// training, so a late round launches and touches only those; the host enqueues round i + 1 while it waits for the 8-byte count of round
// i - 1 (a bound for the grids, and the stop signal).  HIPSTR_EM_HOST_LOOP=1 selects the round-4 loop (host libm, all loci per round).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>
#include <chrono>
#include <sched.h>
#include <unistd.h>

#include "../../include/hipstr_hmm.h"
#include "post_layout.h"
#include "prep.h"
#include "api_internal.h"
#include "cr_math.h"

extern "C" __global__ void hs_posterior_kernel(const hs_post_dev_t* dp);

struct hs_em_locus_t {
  int32_t A, S, R, period, haploid;
  int32_t read_begin;      // into the per-read arrays
  int32_t samp_begin;      // into sample_total
  int32_t bps_off;         // into bps / log_gt_priors (A entries)
  int64_t post_off;        // S*A*A posteriors
  int64_t ll_off;          // R*A log_aln_probs
  int64_t prior_off;       // A*A priors
  int64_t sa_off;          // S*A per-(sample, allele) values (gmax)
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
  unsigned long long* dbg;     // HIPSTR_TIMING: [0..1] rows listed / rows scanned in the maxima pass, [2..3] in the sums pass
  double*  gmax;               // [S*A per locus, at sa_off] largest log posterior of any diplotype of the sample that holds the allele (hs_em_gmax)
  double   log_thresh, log_half, log_1p1;
  // ---- device-resident loop (NULL / unused with the host loop): workgroup b of a per-locus kernel takes locus list[b] if b < counts[0]
  const int32_t* list;         // loci still training, ascending
  const int32_t* counts;       // [0] loci in `list`, [1] their (locus, sample) units
  int32_t* next_list;          // the lists of the next round (hs_em_compact / hs_em_units write them; the two argument blocks swap roles)
  int32_t* next_counts;
  int32_t* next_units;         // (locus, sample) units of the loci in next_list, in order
  int32_t* next_unit_begin;    // [n_loci] scratch: first slot of a listed locus' units in next_units
  const int32_t* unit_first;   // [n_loci] index of the locus' first unit
  double*  logp_rw;            // = logp, writable (hs_em_finish writes the next round's nine logs)
  double*  sp;                 // [6*n_loci] current stutter parameters
  double*  cur_ll;             // [n_loci] LL of the previous round (-DBL_MAX before the first)
  int32_t* iter;               // [n_loci] 1-based number of the round a locus is in
  int32_t* state;              // [n_loci] 0 training, 1 converged (train() returned true), 2 out of iterations (false)
  int32_t* n_iter;             // [n_loci] outputs: rounds run, LL of the last one
  double*  final_ll;
  int32_t  max_iter, n_loci;
  double   min_abs, min_frac;
};
// the locus of this workgroup: through the active list (device loop) or the mask (host loop); -1 = nothing to do
__device__ __forceinline__ int em_locus(const hs_em_dev_t& d){
  if (d.list) return ((int)blockIdx.x < d.counts[0]) ? d.list[blockIdx.x] : -1;
  return d.active[blockIdx.x] ? (int)blockIdx.x : -1;
}
#define HS_EM_PARTS 8          // workgroups per locus in the two big M-step reductions
#define HS_EM_TILE 2048         // rows of a slice whose live rows are listed at a time (hs_em_mstep_part)
#define HS_EM_CHUNK 32          // positions of the allele-frequency scans per round of prepared exponentials (em_gt_priors)
#define HS_EM_MAXA_LDS 64      // alleles whose chains run side by side (lanes of the first wavefront); more alleles: several sweeps

namespace {

// The two float divisions of the reference's bit-trick exp2 / log (fastonebigheader.h:188-198: 27.7280233f / (4.84252568f - z), z in [0, 1];
// :320-338: 1.72587999f / (0.3520887068f + mx), mx in [0.5, 1)) as v_rcp_f32 + one Newton step + a residual correction: six instructions
// instead of the compiler's IEEE sequence (scale, reciprocal, three refinements, fmas, fixup: twice that), and the IEEE quotient for EVERY
// float denominator of both ranges — tools/div_probe.hip checks all 8.4 M of them on the device.  Operands outside those ranges: never here.
__device__ __forceinline__ float e_div_tab(float n, float d){
  float r = __builtin_amdgcn_rcpf(d);
  r = __fmaf_rn(__fmaf_rn(-d, r, 1.0f), r, r);
  const float q = __fmul_rn(n, r);
  return __fmaf_rn(__fmaf_rn(-d, q, n), r, q);
}
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
  const float t = __fsub_rn(__fadd_rn(__fadd_rn(clipp, 121.2740575f), e_div_tab(27.7280233f, __fsub_rn(4.84252568f, z))), __fmul_rn(1.49012907f, z));
  return __uint_as_float((uint32_t)__fmul_rn(8388608.0f, t));
}
__device__ __forceinline__ float e_fastlog(float x){             // fastonebigheader.h:320-338
  const uint32_t vi = __float_as_uint(x);
  const float mx = __uint_as_float((vi & 0x007FFFFFu) | 0x3f000000u);
  float y = (float)vi;
  y = __fmul_rn(y, 1.1920928955078125e-7f);
  const float l2 = __fsub_rn(__fsub_rn(__fsub_rn(y, 124.22551499f), __fmul_rn(1.498030302f, mx)),
                             e_div_tab(1.72587999f, __fadd_rn(0.3520887068f, mx)));
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
  const int l = em_locus(d);
  if (l < 0) return;
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
#ifdef HS_EM_TIME
  unsigned long long tk0 = __builtin_amdgcn_s_memtime(), tk1 = 0, tk2 = 0, tkA = 0, tkB = 0;
#endif
  const int A = L.A, S = L.S;
  const double* post = d.post + L.post_off;
  double* gtp = d.gtp + L.bps_off;
  // log_sum_exp of every row (sample, allele_1) first, all rows in parallel (each row summed in allele_2 order as the reference does);
  // the streaming scans below are sequential per allele by definition
  double* row_lse = d.row_lse + L.post_off;     // S*A values in this locus' own S*A*A region: disjoint between loci whatever their A
  // The rows are fetched a tile at a time into LDS, contiguously (a thread walking its own row straight from memory touches 64 cache lines
  // per load: the kernel was bound by exactly that), then: a thread per row takes the maximum, ALL threads form the exponentials in place,
  // a thread per row adds them in allele order and takes the logarithm.
  extern __shared__ double hs_em_dyn[];
  {
    const int Ap = A | 1;                                // odd row stride: the per-row walks of neighbouring threads fall into different banks
    const int tile_rows = min(256, (int)((2*HS_EM_CHUNK*HS_EM_MAXA_LDS) / Ap));
    __shared__ double s_rm[256];
    __shared__ int s_fm[256];
    if (tile_rows >= 1){
      for (int x0 = 0; x0 < S*A; x0 += tile_rows){
        const int nr = min(tile_rows, S*A - x0);
        for (int e = tid; e < nr*A; e += 256){ const int rr = e / A, j = e - rr*A; hs_em_dyn[rr*Ap + j] = post[(int64_t)x0*A + e]; }
        __syncthreads();
        if (tid < nr){
          const double* row = hs_em_dyn + tid*Ap;
          double rm = row[0]; int fm = 0;
          for (int j = 1; j < A; j++) if (row[j] > rm){ rm = row[j]; fm = j; }
          s_rm[tid] = rm; s_fm[tid] = fm;
        }
        __syncthreads();
        // (behind the row's first maximum the running total is >= 1: a term below 2^-54 leaves it as it is and is not formed — +0.0 instead)
        for (int e = tid; e < nr*A; e += 256){
          const int rr = e / A, j = e - rr*A;
          const double x_ = hs_em_dyn[rr*Ap + j] - s_rm[rr];
          hs_em_dyn[rr*Ap + j] = (j > s_fm[rr] && x_ < -37.43) ? 0.0 : cr_exp(x_);
        }
        __syncthreads();
        if (tid < nr){
          __builtin_amdgcn_s_setprio(3);
          const double* row = hs_em_dyn + tid*Ap;
          double rs = 0.0;
          for (int j = 0; j < A; j++) rs += row[j];
          row_lse[x0 + tid] = s_rm[tid] + cr_log(rs);
          __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();
      }
    } else {                                             // a row does not fit the tile (thousands of alleles): straight from memory
      for (int x = tid; x < S*A; x += 256){
        const double* row = post + (int64_t)x*A;
        double rm = row[0];
        for (int j = 1; j < A; j++) rm = fmax(rm, row[j]);
        double rs = 0.0;
        for (int j = 0; j < A; j++) rs += cr_exp(row[j] - rm);
        row_lse[x] = rm + cr_log(rs);
      }
    }
  }
  __syncthreads();
#ifdef HS_EM_TIME
  tk1 = __builtin_amdgcn_s_memtime();
#endif
  // The two scans of allele a are ONE dependent chain over S + S A values (update_streaming_log_sum_exp, mathops.cpp:72-80): lane a owns it.
  // What is expensive in a step is the exponential — of (value - running maximum) where the value does not exceed the maximum, of
  // (old maximum - value) where it becomes the new one — and the running maximum in front of every step is a prefix maximum of the values
  // alone: it does not depend on the sums.  So the values are taken HS_EM_CHUNK at a time in four phases: (1) all threads put the chunk's
  // values into LDS, (2) lane a walks its chunk once for the maximum in front of every step (a compare per step), (3) ALL threads form
  // the step's exponential with that maximum (correctly rounded, cr_math.h; a term that cannot change a total >= 1 is marked instead),
  // (4) lane a walks its chunk again: an addition per step, a multiplication and an addition where the maximum moves.  Same values,
  // same operations in the same order as the scalar chain, and no exponential inside the dependent chain (until round 5 a lane whose
  // maximum had moved inside a chunk evaluated the rest of the chunk the long way — with 33 chains side by side in one wavefront nearly
  // every step of the walk paid for somebody's exponential: ~540 cycles per step).  (A > 64: alleles beyond the first wavefront's lanes loop.)
  constexpr int CH = HS_EM_CHUNK;
  constexpr int NE = (CH*HS_EM_MAXA_LDS + 255)/256;      // a thread's share of a chunk's (position, allele) pairs
  double* const Ebuf = hs_em_dyn;                         // [CH][na]: the maximum in front of the step, then the step's exponential
  double* const Vbuf = hs_em_dyn + CH*HS_EM_MAXA_LDS;    // [CH][na]: the values (the walks are dependent chains and must not wait for L2 at every step)
  const int64_t n1 = S, n2 = (int64_t)S*A, ntot = n1 + n2;
  for (int a0 = 0; a0 < A; a0 += HS_EM_MAXA_LDS){
    const int na = min(HS_EM_MAXA_LDS, A - a0);
    double m = -DBL_MAX/2, t = 0.0;                      // lane a - a0 < na of the first wavefront: the chain's state
    // a thread's pairs are the same in every chunk: position i (of the chunk) and allele al of pair e = tid + 256 q
    int pi[NE], pidx[NE];
#pragma unroll
    for (int q = 0; q < NE; q++){ const int e = tid + 256*q; const int i = e / na; pi[q] = i; pidx[q] = i*na + (e - i*na); }
    double nxt[NE];
    auto fetch = [&](int64_t c0){                        // this thread's share of the chunk at c0: requested here, used after the next barrier but one
      const int cn = (int)min((int64_t)CH, ntot - c0);
#pragma unroll
      for (int q = 0; q < NE; q++) if (pi[q] < cn){
        const int64_t ci = c0 + pi[q]; const int a = a0 + (pidx[q] - pi[q]*na);
        nxt[q] = ci < n1 ? row_lse[ci*A + a] : post[(ci - n1)*A + a];      // the chain of allele a: row_lse(s, a) for s < S, then post[(s, i1), a]
      }
    };
    fetch(0);
    for (int64_t c0 = 0; c0 < ntot; c0 += CH){
      const int cn = (int)min((int64_t)CH, ntot - c0);
#pragma unroll
      for (int q = 0; q < NE; q++) if (pi[q] < cn) Vbuf[pidx[q]] = nxt[q];
      __syncthreads();
#ifdef HS_EM_TIME
      const unsigned long long ta = __builtin_amdgcn_s_memtime();
#endif
      if (tid < na){                                     // (2) the maximum in front of every step
        __builtin_amdgcn_s_setprio(3);                   // the workgroup waits for this one wavefront's chain: it goes first whenever it is ready
        double mr = m;
        for (int i0 = 0; i0 < cn; i0 += 8){
          double lvv[8];
#pragma unroll
          for (int q = 0; q < 8; q++) lvv[q] = Vbuf[min(i0 + q, cn - 1)*na + tid];
#pragma unroll
          for (int q = 0; q < 8; q++){
            if (i0 + q >= cn) break;
            Ebuf[(i0 + q)*na + tid] = mr;
            mr = (lvv[q] > mr) ? lvv[q] : mr;            // (the walk below moves its maximum by the same test)
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
#ifdef HS_EM_TIME
      tkB += __builtin_amdgcn_s_memtime() - ta;
#endif
      __syncthreads();
      if (c0 + CH < ntot) fetch(c0 + CH);                // the next chunk's values travel while this chunk's exponentials are formed
#pragma unroll
      for (int q = 0; q < NE; q++) if (pi[q] < cn){      // (3)
        const double lv = Vbuf[pidx[q]], mb = Ebuf[pidx[q]];
        double ex;
        if (lv <= mb){ const double x_ = lv - mb; ex = (x_ < -37.43) ? -1.0 : cr_exp(x_); }       // -1: "below 2^-54" (see the walk)
        else ex = cr_exp(mb - lv);                        // the factor of the total gathered under the old maximum
        Ebuf[pidx[q]] = ex;
      }
      __syncthreads();
#ifdef HS_EM_TIME
      const unsigned long long tb = __builtin_amdgcn_s_memtime();
#endif
      if (tid < na){                                     // (4)
        __builtin_amdgcn_s_setprio(3);
        // eight steps' operands are fetched together, ahead of the (dependent) chain: a step that waited for its own two LDS reads took ~400 cycles
        for (int i0 = 0; i0 < cn; i0 += 8){
          double lvv[8], exv[8];
#pragma unroll
          for (int q = 0; q < 8; q++){
            const int i = min(i0 + q, cn - 1);
            lvv[q] = Vbuf[i*na + tid]; exv[q] = Ebuf[i*na + tid];
          }
#pragma unroll
          for (int q = 0; q < 8; q++){
            if (i0 + q >= cn) break;
            const double lv = lvv[q];
            if (lv <= m){
              // (once a maximum is set the total is >= 1: a term below 2^-54 leaves it as it is, bit for bit; before that — t < 1 only while
              //  m is still the initial -DBL_MAX/2, where no value is below it — it cannot occur)
              if (exv[q] >= 0.0) t += exv[q];
            } else { t *= exv[q]; t += 1.0; m = lv; }
          }
        }
        __builtin_amdgcn_s_setprio(0);
      }
#ifdef HS_EM_TIME
      tkB += __builtin_amdgcn_s_memtime() - tb;
#endif
      __syncthreads();
    }
    if (tid < na) gtp[a0 + tid] = m + cr_log(t);
    __syncthreads();
  }
#ifdef HS_EM_TIME
  tk2 = __builtin_amdgcn_s_memtime();
  if (tid == 0 && (blockIdx.x % 2000) == 7) printf("gt_priors locus %d A %d S %d: rows %llu  scans %llu (walks %llu)\n", (int)blockIdx.x, A, S, tk1 - tk0, tk2 - tk1, tkB);
#endif
  if (tid == 0){                                          // normalise: exact log_sum_exp in allele order
    double m = gtp[0];
    for (int a = 1; a < A; a++) m = fmax(m, gtp[a]);
    double t = 0.0;
    for (int a = 0; a < A; a++) t += cr_exp(gtp[a] - m);
    const double lt = m + cr_log(t);
    for (int a = 0; a < A; a++) gtp[a] -= lt;
  }
}

// gmax[s][a] = the largest log posterior among the diplotypes (a, j) and (j, a) of sample s: the bound hs_em_mstep_part prunes by.
// A wavefront per sample, lanes = the second allele: every row of the sample's A x A block is read once, contiguously (a thread per
// (sample, allele) walking its row and its column touched 64 cache lines per load: 15 ms per round of 10 000 loci, now ~1).
__global__ void __launch_bounds__(256) hs_em_gmax(const hs_em_dev_t* __restrict__ dp){
  const hs_em_dev_t& d = *dp;
  const int l = em_locus(d);
  if (l < 0) return;
  const hs_em_locus_t L = d.loci[l];
  const int A = L.A, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const double* post = d.post + L.post_off;
  double* g = d.gmax + L.sa_off;
  if (A <= 64){
    for (int s_ = w; s_ < L.S; s_ += 4){
      const double* gp = post + (int64_t)s_*A*A;
      double colmax = -DBL_MAX, mine = -DBL_MAX;              // lane j: max over a of gp[a][j];  lane a: max over j of gp[a][j]
      for (int a = 0; a < A; a++){
        const double v = (lane < A) ? gp[a*A + lane] : -DBL_MAX;
        colmax = fmax(colmax, v);
        double rm = v;
        for (int o = 32; o >= 1; o >>= 1) rm = fmax(rm, __shfl_xor(rm, o));
        if (lane == a) mine = rm;
      }
      if (lane < A) g[s_*A + lane] = fmax(colmax, mine);
    }
  } else {
    for (int x = threadIdx.x; x < L.S*A; x += 256){
      const int s_ = x / A, a = x - s_*A;
      const double* gp = post + (int64_t)s_*A*A;
      double m = gp[a*A];
      for (int j = 0; j < A; j++) m = fmax(m, fmax(gp[a*A + j], gp[j*A + a]));
      g[x] = m;
    }
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
  const int l = em_locus(d), k_part = blockIdx.y, tid = threadIdx.x;
  if (l < 0) return;
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
  // Round 5: most (read, source allele) rows cannot matter and are recognised without walking them.  Every term of a row is
  //   f = log P(diplotype | sample) + log P(phase | read, diplotype)  <=  G + c,   G = gmax[sample][a] (hs_em_gmax: the sample's best diplotype
  // that holds allele a), c = 4e-6 (the phase term is <= 0 up to the float log-sum-exp's approximation error: its bit-trick log dips to
  // -1.65e-6 at 1.0, fastonebigheader.h:320-338), and every operation between f and its use is monotone in f.  So
  //   PASS 0: a row whose bound cannot exceed the pseudocount entries every vector starts from (0, ln 1.1) cannot raise a maximum;
  //   PASS 1: a row whose bound is not above max + LOG_THRESH only has terms the threshold drops (mathops.cpp:102-103)
  // — and neither is walked.  The rows that are left (a few per cent: the alleles of a read's sample's plausible genotypes) are gathered
  // into a list per tile of HS_EM_TILE rows, so that the walk keeps every lane busy.  Same maxima, same sums (of float terms, in double:
  // exact in any order).
  const double* gmax = d.gmax + L.sa_off;
  constexpr double PH_C = 4.0e-6;
  __shared__ int s_rows[HS_EM_TILE];
  __shared__ int s_cnt;
  for (int t0 = x0; t0 < x1; t0 += HS_EM_TILE){
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const int t1 = min(x1, t0 + HS_EM_TILE);
    for (int x = t0 + tid; x < t1; x += 256){
      const int r = x / A, a = x - r*A;
      const double Gb = gmax[(int64_t)d.sample_label[L.read_begin + r]*A + a] + PH_C;
      const int c2 = catv[x];
      const int cat = c2 & 0xff, dcat = c2 >> 8;
      const double le = leff[x];
      bool need;
      if (PASS == 0){
        need = Gb > 0.0;                                                   // (vectors 0, 1, 2, 4, 5 start from the pseudocount entry 0.0)
        if (dcat < 7) need = need || (Gb + le > d.log_1p1);                // (the diffs vectors from ln 1.1)
      } else {
        double mc = 0.0, md = 0.0;
#pragma unroll
        for (int k = 0; k < 7; k++){ if (k == cat) mc = mx[k]; if (k == dcat) md = mx[k]; }
        need = (Gb - mc > d.log_thresh);
        if (dcat < 7) need = need || ((Gb + le) - md > d.log_thresh);
      }
      if (need) s_rows[atomicAdd(&s_cnt, 1)] = x;
    }
    __syncthreads();
    const int n_live = s_cnt;
    if (d.dbg && tid == 0){ atomicAdd(d.dbg + 2*PASS, (unsigned long long)n_live); atomicAdd(d.dbg + 2*PASS + 1, (unsigned long long)(t1 - t0)); }
    for (int li = tid; li < n_live; li += 256){
      const int x = s_rows[li];
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
    __syncthreads();
  }
  double out[7];
  for (int k = 0; k < 7; k++) out[k] = (PASS == 0) ? block_max(acc[k], red) : block_sum(acc[k], red);
  if (tid == 0) for (int k = 0; k < 7; k++) part[k_part*7 + k] = out[k];
}

// recalc_log_gt_priors of every active locus: long dependent scans on a few lanes.  Nothing in the stutter reductions needs their result
// (the next iteration's hs_em_fill does), so the kernel runs beside them on a second stream.
__global__ void __launch_bounds__(256) hs_em_gt_priors(const hs_em_dev_t* __restrict__ dp){
  const hs_em_dev_t& d = *dp;
  const int l = em_locus(d);
  if (l < 0) return;
  const hs_em_locus_t L = d.loci[l];
  em_gt_priors(d, L, threadIdx.x);
}

__global__ void __launch_bounds__(64) hs_em_mstep_keepmax(const hs_em_dev_t* __restrict__ dp, double* __restrict__ keep){
  const hs_em_dev_t& d = *dp;
  const int l = em_locus(d);
  if (l < 0 || threadIdx.x >= 7) return;
  const double* part = d.part + ((size_t)l*HS_EM_PARTS)*7;
  double m = (threadIdx.x == 3 || threadIdx.x == 6) ? d.log_1p1 : 0.0;
  for (int q = 0; q < HS_EM_PARTS; q++) m = fmax(m, part[q*7 + threadIdx.x]);
  keep[7*l + threadIdx.x] = m;
}

__global__ void __launch_bounds__(256) hs_em_mstep(const hs_em_dev_t* __restrict__ dp, const double* __restrict__ keep){
  const hs_em_dev_t& d = *dp;
  const int l = em_locus(d), tid = threadIdx.x;
  if (l < 0) return;
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

// ---- the device-resident loop (round 5) ---------------------------------------------------------------------------------
// StutterModel's nine logs (stutter_model.h:44-58) from the six parameters, correctly rounded
__device__ __forceinline__ void em_logs(const double* sp, double* q){
  q[0] = cr_log(1-sp[0]); q[1] = cr_log(sp[0]); q[2] = cr_log(sp[1]); q[3] = cr_log(sp[2]);
  q[4] = cr_log(1-sp[3]); q[5] = cr_log(sp[3]); q[6] = cr_log(sp[4]); q[7] = cr_log(sp[5]);
  q[8] = cr_log(1-sp[1]-sp[2]-sp[4]-sp[5]);
}
__device__ __forceinline__ double em_lse2_exact(double a, double b){      // mathops.cpp:52-57
  return a > b ? a + cr_log(1 + cr_exp(b - a)) : b + cr_log(1 + cr_exp(a - b));
}

// init_stutter_model (:59-62) and the loop's starting state, a thread per locus
__global__ void __launch_bounds__(256) hs_em_init(const hs_em_dev_t* __restrict__ dp){
  const hs_em_dev_t& d = *dp;
  const int l = blockIdx.x*256 + threadIdx.x;
  if (l >= d.n_loci) return;
  const double init[6] = { 0.9, 0.1, 0.1, 0.8, 0.01, 0.01 };
  for (int k = 0; k < 6; k++) d.sp[6*l + k] = init[k];
  em_logs(init, d.logp_rw + 9*(size_t)l);
  d.cur_ll[l] = -DBL_MAX; d.iter[l] = 1; d.n_iter[l] = 0; d.final_ll[l] = 0.0;
  d.state[l] = (1 > d.max_iter) ? 2 : 0;                  // no iteration allowed: train() returns false without running one
}

// The end of a round for one locus (a thread): the E-step's total LL and the seven M-step totals (what hs_em_mstep hands the host loop),
// then train()'s tests and update (:186-224) in the reference's order, exp / log correctly rounded, and the next round's nine logs.
__global__ void __launch_bounds__(64) hs_em_finish(const hs_em_dev_t* __restrict__ dp, const double* __restrict__ keep){
  const hs_em_dev_t& d = *dp;
  const int l = em_locus(d);
  if (l < 0 || threadIdx.x != 0) return;
  const hs_em_locus_t L = d.loci[l];
  double new_LL = 0.0;
  for (int s = 0; s < L.S; s++) new_LL += d.sample_total[L.samp_begin + s];     // genotyper.cpp:75
  double t[7];
  const double* part = d.part + ((size_t)l*HS_EM_PARTS)*7;
  for (int k = 0; k < 7; k++){
    const double mxk = keep[7*l + k];
    double tt = 0.0;
    for (int q = 0; q < HS_EM_PARTS; q++) tt += part[q*7 + k];
    { const double df = 0.0 - mxk; if (df > d.log_thresh) tt += (double)e_fasterexp((float)df); }
    if (k == 3 || k == 6){ const double df = d.log_1p1 - mxk; if (df > d.log_thresh) tt += (double)e_fasterexp((float)df); }
    t[k] = mxk + (double)e_fasterlog((float)tt);
  }
  const int it = d.iter[l];
  d.n_iter[l] = it; d.final_ll[l] = new_LL;
  const double LL = d.cur_ll[l];
  if (new_LL < LL + 1e-10){ d.state[l] = 1; return; }                           // :190-194 (TOLERANCE = 1e-10)
  const double out_total = e_fast_lse2(t[4], t[5], d.log_thresh);
  const double in_pgeom  = fmin(0.999, cr_exp(em_lse2_exact(t[0], t[1]) - t[3]));
  const double out_pgeom = fmin(0.999, cr_exp(out_total - t[6]));
  const double m3 = fmax(fmax(t[0], t[1]), t[2]);
  const double lse3 = m3 + cr_log(cr_exp(t[0]-m3) + cr_exp(t[1]-m3) + cr_exp(t[2]-m3));      // mathops.cpp:59-62
  const double log_total = em_lse2_exact(lse3, out_total);
  const double nw[6] = { in_pgeom, cr_exp(t[0] - log_total), cr_exp(t[1] - log_total), out_pgeom, cr_exp(t[4] - log_total), cr_exp(t[5] - log_total) };
  const double abs_change = new_LL - LL, frac_change = -(new_LL - LL)/LL;
  bool conv = false;
  if (abs_change < d.min_abs && frac_change < d.min_frac) conv = true;
  else {
    conv = true;
    for (int k = 0; k < 6; k++) if (!(fabs(d.sp[6*l + k] - nw[k]) < 0.0001)) conv = false;     // parameters_within_threshold (stutter_model.h:62-65)
  }
  for (int k = 0; k < 6; k++) d.sp[6*l + k] = nw[k];
  if (conv){ d.state[l] = 1; return; }
  d.cur_ll[l] = new_LL; d.iter[l] = it + 1;
  if (it + 1 > d.max_iter){ d.state[l] = 2; return; }                           // ran out of iterations: train() returns false
  em_logs(nw, d.logp_rw + 9*(size_t)l);
}

// The loci still training, ascending, and where their (locus, sample) units go: one workgroup, an exclusive scan over the loci in chunks.
__global__ void __launch_bounds__(1024) hs_em_compact(const hs_em_dev_t* __restrict__ dp){
  const hs_em_dev_t& d = *dp;
  const int tid = threadIdx.x;
  __shared__ int sc_n[1024], sc_u[1024];
  __shared__ int base_n, base_u;
  if (tid == 0){ base_n = 0; base_u = 0; }
  __syncthreads();
  for (int l0 = 0; l0 < d.n_loci; l0 += 1024){
    const int l = l0 + tid;
    const int on = (l < d.n_loci && d.state[l] == 0) ? 1 : 0;
    const int nu = on ? d.loci[l].S : 0;
    sc_n[tid] = on; sc_u[tid] = nu;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1){               // inclusive scan (Hillis-Steele)
      const int a = (tid >= off) ? sc_n[tid - off] : 0, b = (tid >= off) ? sc_u[tid - off] : 0;
      __syncthreads();
      sc_n[tid] += a; sc_u[tid] += b;
      __syncthreads();
    }
    if (on){
      d.next_list[base_n + sc_n[tid] - 1] = l;
      d.next_unit_begin[l] = base_u + sc_u[tid] - nu;
    }
    __syncthreads();
    if (tid == 1023){ base_n += sc_n[1023]; base_u += sc_u[1023]; }
    __syncthreads();
  }
  if (tid == 0){ d.next_counts[0] = base_n; d.next_counts[1] = base_u; }
}
__global__ void __launch_bounds__(256) hs_em_units(const hs_em_dev_t* __restrict__ dp){
  const hs_em_dev_t& d = *dp;
  if ((int)blockIdx.x >= d.next_counts[0]) return;
  const int l = d.next_list[blockIdx.x];
  const int S = d.loci[l].S, b = d.next_unit_begin[l], u0 = d.unit_first[l];
  for (int s = threadIdx.x; s < S; s += 256) d.next_units[b + s] = u0 + s;
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
  const bool host_loop = getenv("HIPSTR_EM_HOST_LOOP") && atoi(getenv("HIPSTR_EM_HOST_LOOP")) != 0;       // the round-4 loop: host libm, every locus in every round
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
    // (cr_math.h's correctly rounded log on the host as well: every exp / log of the EM is the same function on either side — the host libm's
    //  bits wherever that is correctly rounded; HIPSTR_EM_HOST_LOOP=1 takes the libm itself, as in round 4)
    const double lt = host_loop ? log(tot) : cr_log(tot);
    Q.gtp.resize(A);
    for (int a = 0; a < A; a++) Q.gtp[a] = (host_loop ? log(g[a]) : cr_log(g[a])) - lt;
  });
  int64_t post_off = 0, ll_off = 0, prior_off = 0, sa_off = 0; int samp_off = 0;
  for (int l = 0; l < nl; l++){
    const LocusPrep& Q = lp[l];
    if (Q.err) return api_fail(Q.err);
    const int S = eb->n_samples[l], r0 = eb->read_off[l], r1 = eb->read_off[l+1], R = r1 - r0;
    const int A = (int)Q.sizes.size();
    hs_em_locus_t& L = loci[l];
    memset(&L, 0, sizeof L);
    L.A = A; L.S = S; L.R = R; L.period = eb->period[l]; L.haploid = (eb->haploid && eb->haploid[l]) ? 1 : 0;
    L.read_begin = r0; L.samp_begin = samp_off; L.bps_off = (int32_t)bps.size();
    L.post_off = post_off; L.ll_off = ll_off; L.prior_off = prior_off; L.sa_off = sa_off;
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
    post_off += (int64_t)S*A*A; ll_off += (int64_t)R*A; prior_off += (int64_t)A*A; samp_off += S; sa_off += (int64_t)S*A;
  }
  std::vector<LocusPrep>().swap(lp);
  lap("alleles and units", NULL);
  // ---- device state
  EmBufs dev;
  dev.stream = T.stream;
  hs_em_dev_t h; memset(&h, 0, sizeof h);
  hs_post_dev_t ph; memset(&ph, 0, sizeof ph);
  hs_em_locus_t* d_loci; hs_post_unit_t* d_units; int32_t *d_active, *d_unit_active, *d_bps, *d_obs, *d_lab, *d_w, *d_mapgt;
  double *d_logp, *d_p1, *d_p2, *d_gtp, *d_ll, *d_prior, *d_post, *d_tot, *d_newll, *d_sums, *d_rowlse, *d_leff, *d_part, *d_keep, *d_gmax; int32_t* d_cat;
  std::vector<int32_t> ones(n_reads, 1);
  if (dev.put(&d_loci, loci.data(), loci.size()) || dev.put(&d_units, units.data(), units.size()) || dev.alloc(&d_active, nl) ||
      dev.alloc(&d_unit_active, units.size()) || dev.put(&d_bps, bps.data(), bps.size()) || dev.put(&d_obs, obs.data(), obs.size()) ||
      dev.put(&d_lab, eb->sample_label, n_reads) || dev.put(&d_w, ones.data(), ones.size()) || dev.alloc(&d_mapgt, 2*(size_t)samp_off) ||
      dev.alloc(&d_logp, 9*(size_t)nl) || dev.put(&d_p1, eb->log_p1, n_reads) || dev.put(&d_p2, eb->log_p2, n_reads) ||
      dev.put(&d_gtp, gtp.data(), gtp.size()) || dev.alloc(&d_ll, ll_off) || dev.alloc(&d_prior, prior_off) || dev.alloc(&d_post, post_off) ||
      dev.alloc(&d_tot, samp_off) || dev.alloc(&d_newll, nl) || dev.alloc(&d_sums, 7*(size_t)nl) || dev.alloc(&d_rowlse, post_off) ||
      dev.alloc(&d_cat, ll_off) || dev.alloc(&d_leff, ll_off) || dev.alloc(&d_part, 7*(size_t)HS_EM_PARTS*nl) || dev.alloc(&d_keep, 7*(size_t)nl) || dev.alloc(&d_gmax, sa_off)) return 1;
  h.loci = d_loci; h.active = d_active; h.logp = d_logp; h.bps = d_bps; h.obs = d_obs; h.sample_label = d_lab; h.log_p1 = d_p1; h.log_p2 = d_p2;
  h.gtp = d_gtp; h.ll = d_ll; h.prior = d_prior; h.post = d_post; h.sample_total = d_tot; h.int_log = T.int_log; h.new_ll = d_newll; h.sums = d_sums; h.row_lse = d_rowlse; h.cat = d_cat; h.leff = d_leff; h.part = d_part; h.gmax = d_gmax;
  unsigned long long* d_dbg = NULL;
  if (timing){ if (dev.alloc(&d_dbg, 4)) return 1; EM_HIP(hipMemset(d_dbg, 0, 4*sizeof(unsigned long long))); }
  h.dbg = d_dbg;
  h.log_thresh = HT.log_thresh; h.log_half = HT.log_half; h.log_1p1 = host_loop ? log(1.1) : cr_log(1.1);
  ph.units = d_units; ph.log_aln_probs = d_ll; ph.log_p1 = d_p1; ph.log_p2 = d_p2; ph.read_weight = d_w; ph.log_prior = d_prior;
  ph.unit_active = d_unit_active; ph.log_post = d_post; ph.sample_total = d_tot; ph.map_gt = d_mapgt;
  ph.log_thresh = HT.log_thresh; ph.log_half = HT.log_half; ph.sym_prior = 1;          // hs_em_fill: log f(a1) + log f(a2), or the haploid diagonal
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
  if (!host_loop){
    // ---- the device-resident loop: state, the two sets of lists (a round reads one and writes the other), two argument blocks each for
    // the EM kernels and the posterior kernel that differ only in which set is which
    const size_t n_units = units.size();
    std::vector<int32_t> unit_first(nl);
    { int32_t u = 0; for (int l = 0; l < nl; l++){ unit_first[l] = u; u += loci[l].S; } }
    int32_t *d_list[2], *d_counts[2], *d_ulist[2], *d_ubegin, *d_ufirst, *d_iter, *d_state, *d_niter; double *d_sp, *d_curll, *d_fll;
    if (dev.alloc(&d_list[0], nl) || dev.alloc(&d_list[1], nl) || dev.alloc(&d_counts[0], 2) || dev.alloc(&d_counts[1], 2) || dev.alloc(&d_ulist[0], n_units) ||
        dev.alloc(&d_ulist[1], n_units) || dev.alloc(&d_ubegin, nl) || dev.put(&d_ufirst, unit_first.data(), unit_first.size()) || dev.alloc(&d_iter, nl) ||
        dev.alloc(&d_state, nl) || dev.alloc(&d_niter, nl) || dev.alloc(&d_sp, 6*(size_t)nl) || dev.alloc(&d_curll, nl) || dev.alloc(&d_fll, nl)) return 1;
    hs_em_dev_t hb[2]; hs_post_dev_t pb2[2];
    for (int k = 0; k < 2; k++){
      hb[k] = h; hb[k].active = NULL;
      hb[k].list = d_list[k]; hb[k].counts = d_counts[k]; hb[k].next_list = d_list[k ^ 1]; hb[k].next_counts = d_counts[k ^ 1]; hb[k].next_units = d_ulist[k ^ 1];
      hb[k].next_unit_begin = d_ubegin; hb[k].unit_first = d_ufirst; hb[k].logp_rw = d_logp; hb[k].sp = d_sp; hb[k].cur_ll = d_curll; hb[k].iter = d_iter;
      hb[k].state = d_state; hb[k].n_iter = d_niter; hb[k].final_ll = d_fll; hb[k].max_iter = eb->max_iter; hb[k].n_loci = nl;
      hb[k].min_abs = eb->min_ll_abs_change; hb[k].min_frac = eb->min_ll_frac_change;
      pb2[k] = ph; pb2[k].unit_active = NULL; pb2[k].unit_list = d_ulist[k]; pb2[k].n_list = d_counts[k] + 1;
    }
    hs_em_dev_t* d_hb; hs_post_dev_t* d_pb;
    if (dev.put(&d_hb, hb, 2) || dev.put(&d_pb, pb2, 2)) return 1;
    struct Pinned { hipstr::Ctx* ctx; int32_t* p; ~Pinned(){ if (p) hipstr::pin_free(ctx, p); } } pin{T.ctx, NULL};
    constexpr int NBUF = 4;
    pin.p = (int32_t*)hipstr::pin_alloc(T.ctx, NBUF*2*sizeof(int32_t));
    if (!pin.p) return 1;
    hipEvent_t ev_cnt[NBUF] = {NULL, NULL, NULL, NULL};
    struct EvGuard { hipEvent_t* e; int n; ~EvGuard(){ for (int i = 0; i < n; i++) if (e[i]) hipEventDestroy(e[i]); } } evg{ev_cnt, NBUF};
    for (int i = 0; i < NBUF; i++) EM_HIP(hipEventCreateWithFlags(&ev_cnt[i], hipEventDisableTiming));
    auto wait_event = [&](hipEvent_t e) -> hipError_t {           // without burning a core (hipEventSynchronize spins: tools/wait_probe.hip)
      for (unsigned n = 0;; n++){
        const hipError_t q = hipEventQuery(e);
        if (q != hipErrorNotReady) return q;
        if (n < 2000) sched_yield(); else usleep(20);
      }
    };
    lap("device-loop state", NULL);
    const unsigned g_loci = (unsigned)((nl + 255)/256);
    // the starting state, and the lists of round 0 (written through block 1, whose "next" set is block 0's current one)
    hipLaunchKernelGGL(hs_em_init, dim3(g_loci), dim3(256), 0, T.stream, (const hs_em_dev_t*)(d_hb + 1));
    hipLaunchKernelGGL(hs_em_compact, dim3(1), dim3(1024), 0, T.stream, (const hs_em_dev_t*)(d_hb + 1));
    hipLaunchKernelGGL(hs_em_units, dim3(nl), dim3(256), 0, T.stream, (const hs_em_dev_t*)(d_hb + 1));
    unsigned bound_l = (unsigned)nl, bound_u = (unsigned)n_units;
    int rounds = 0;
    constexpr bool em_serial = false;      // (the allele-frequency scans run beside the M-step on the side stream: 0.51 against 0.57 s per 10 000 loci, profiles/r05_notes.md)
    for (int r = 0; r <= eb->max_iter + 1; r++){
      const hs_em_dev_t* H = d_hb + (r & 1); const hs_post_dev_t* PH = d_pb + (r & 1);
      if (bound_l > 0){
        hipLaunchKernelGGL(hs_em_fill, dim3(bound_l), dim3(256), 0, T.stream, H);
        hipLaunchKernelGGL(hs_posterior_kernel, dim3(std::max(1u, bound_u)), dim3(256), 0, T.stream, PH);
        EM_HIP(hipEventRecord(side.ev_fork, T.stream));                  // posteriors are in place: the allele-frequency scans branch off
        EM_HIP(hipStreamWaitEvent(side.stream, side.ev_fork, 0));
        hipLaunchKernelGGL(hs_em_gt_priors, dim3(bound_l), dim3(256), 2*HS_EM_CHUNK*HS_EM_MAXA_LDS*sizeof(double), em_serial ? T.stream : side.stream, H);
        EM_HIP(hipEventRecord(side.ev_join, side.stream));
        hipLaunchKernelGGL(hs_em_gmax, dim3(bound_l), dim3(256), 0, T.stream, H);
        hipLaunchKernelGGL(hs_em_mstep_part<0>, dim3(bound_l, HS_EM_PARTS), dim3(256), 0, T.stream, H, (const double*)NULL);
        hipLaunchKernelGGL(hs_em_mstep_keepmax, dim3(bound_l), dim3(64), 0, T.stream, H, d_keep);
        hipLaunchKernelGGL(hs_em_mstep_part<1>, dim3(bound_l, HS_EM_PARTS), dim3(256), 0, T.stream, H, (const double*)d_keep);
        hipLaunchKernelGGL(hs_em_finish, dim3(bound_l), dim3(64), 0, T.stream, H, (const double*)d_keep);
        EM_HIP(hipStreamWaitEvent(T.stream, side.ev_join, 0));            // the next round's hs_em_fill reads the new allele frequencies
        hipLaunchKernelGGL(hs_em_compact, dim3(1), dim3(1024), 0, T.stream, H);
        hipLaunchKernelGGL(hs_em_units, dim3(bound_l), dim3(256), 0, T.stream, H);
        rounds++;
      }
      EM_HIP(hipMemcpyAsync(pin.p + 2*(r % NBUF), d_counts[(r & 1) ^ 1], 2*sizeof(int32_t), hipMemcpyDeviceToHost, T.stream));
      EM_HIP(hipEventRecord(ev_cnt[r % NBUF], T.stream));
      EM_HIP(hipGetLastError());
      if (r >= 1){
        // what round r - 1 left: the loci (and units) of round r — a bound for the grids of round r + 1, and the stop signal — while
        // round r is already queued
        EM_HIP(wait_event(ev_cnt[(r - 1) % NBUF]));
        bound_l = (unsigned)pin.p[2*((r - 1) % NBUF)]; bound_u = (unsigned)pin.p[2*((r - 1) % NBUF) + 1];
        if (bound_l == 0) break;
      }
    }
    EM_HIP(hipstr::wait_stream(T.stream));
    lap("device loop", &t_gpu);
    std::vector<int32_t> state(nl), niter(nl);
    EM_HIP(hipMemcpy(state.data(), d_state, nl*sizeof(int32_t), hipMemcpyDeviceToHost));
    EM_HIP(hipMemcpy(niter.data(), d_niter, nl*sizeof(int32_t), hipMemcpyDeviceToHost));
    EM_HIP(hipMemcpy(final_ll, d_fll, nl*sizeof(double), hipMemcpyDeviceToHost));
    EM_HIP(hipMemcpy(stutter, d_sp, 6*(size_t)nl*sizeof(double), hipMemcpyDeviceToHost));
    for (int l = 0; l < nl; l++){ trained[l] = state[l] == 1 ? 1 : 0; n_iter[l] = niter[l]; }
    if (timing){
      unsigned long long c[4] = {0, 0, 0, 0};
      if (d_dbg) hipMemcpy(c, d_dbg, sizeof c, hipMemcpyDeviceToHost);
      fprintf(stderr, "hipstr_em_train: device-resident loop, %d rounds queued: %.3f ms; M-step rows walked: maxima pass %llu of %llu, sums pass %llu of %llu\n", rounds, 1e3*t_gpu, c[0], c[1], c[2], c[3]);
    }
    return 0;
  }
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
    hipLaunchKernelGGL(hs_em_gt_priors, dim3(nl), dim3(256), 2*HS_EM_CHUNK*HS_EM_MAXA_LDS*sizeof(double), side.stream, d_h);
    EM_HIP(hipEventRecord(side.ev_join, side.stream));
    hipLaunchKernelGGL(hs_em_gmax, dim3(nl), dim3(256), 0, T.stream, d_h);
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
