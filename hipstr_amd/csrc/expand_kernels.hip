// expand_kernels.hip — tables the device builds for itself (round 4).
//
// The constants and the closed-form table of an STR option (prep.cpp emit_stropt: stutter pmf, position priors, one {A, G, Bnd} entry per
// list and bound class) and the 256-byte per-allele records of hs_str_group_kernel_p were 80 % of the bytes the host wrote, packed and sent
// per locus — 16 KB of a 30x locus' 22 KB of tables — although they are functions of (block length, period) and 13 numbers per locus.
// The host now sends those inputs and reserves the space; these two kernels, queued on the upload's stream behind the copy, fill it in
// before any other kernel runs.  Same arithmetic as the host code they replace, operation for operation (prep.cpp simple_table_entry_compute;
// tests/test_expand_gpu.py compares the two byte for byte): double adds and compares, the reference's float bit tricks, int_log from the
// table the host computed with its libm.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "layout.h"
#include "device_common.h"

namespace {

// prep.cpp simple_table_entry_compute
__device__ void table_entry(const hs_dev_t& d, int lim, int U0, int tail, double* ent){
  const bool skip = (U0 > 0) && (lim > 0);
  const int nplain = lim - U0 > 0 ? lim - U0 : 0;
  const int stop = (lim <= 0) ? 0 : ((U0 > 0 && lim <= U0) ? U0 : lim);
  const bool has_tail = stop < tail;
  const double a[3] = {0.0, d.int_log[U0], d.int_log[tail - stop > 0 ? tail - stop : 0]};
  const bool on[3] = {true, skip, has_tail};
  const double w[3] = {(double)(1 + nplain), 1.0, 1.0};
  double amax = 0.0;
  for (int i = 1; i < 3; i++) if (on[i] && a[i] > amax) amax = a[i];
  double tot = 0.0, delta = 1e300;
  for (int i = 0; i < 3; i++){
    if (!on[i]) continue;
    if (a[i] == amax){ tot += w[i] * (double)f_fasterexp(0.0f); continue; }
    const double x = a[i] - amax;
    const float f = (float)x;
    const double m_lo = 0.5*((double)f + (double)nextafterf(f, -INFINITY)), m_hi = 0.5*((double)f + (double)nextafterf(f, INFINITY));
    const double d1 = x - m_lo, d2 = m_hi - x;
    const double dm = d2 < d1 ? d2 : d1;
    if (dm < delta) delta = dm;
    const double dt = fabs(x - d.log_thresh);
    if (dt < delta) delta = dt;
    if (x > d.log_thresh) tot += w[i] * (double)f_fasterexp(f);
  }
  ent[0] = amax;
  ent[1] = (double)f_fasterlog((float)tot);
  const double b = delta * 1125899906842624.0 /* 2^50 */ - amax - 1.0;
  ent[2] = (delta >= 1e300) ? 1e300 : (b > 0.0 ? b : 0.0);
}

}  // namespace

// One WAVEFRONT per STR option the host marked `gen` (layout.h hs_stropt_t): lanes 0-19 write the 20 constants, every lane one entry of the
// table (at most HS_TAB_CAP = 48 entries: list and bound class from the lists' entry counts), the smallest Bnd by a wave-wide minimum
// (exact in any order).  One thread per option did the ~20 entries one after the other: 20 us for the 64 options of a one-locus call,
// a sixteenth of its device time; the entries are independent.
extern "C" __global__ void __launch_bounds__(256) hs_expand_stropts_kernel(const hs_dev_t* dp){
  const hs_dev_t& d = *dp;
  const int s = blockIdx.x*4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (s >= d.n_stropts) return;
  hs_stropt_t* so = (hs_stropt_t*)d.stropts + s;
  if (!so->gen) return;                                   // (the same for every lane of the wavefront; nobody has cleared it yet: see the end)
  const int B = so->B, period = so->period;
  const int f64_off = so->f64_off, tab_off = so->tab_off, tab_len = so->tab_len;
  double* pool = (double*)d.f64pool + d.f64_gen_base;
  double* c = pool + f64_off;
  const double* pmf = d.pmf13 + so->pmf_off;
  if (lane < HS_NART) c[lane] = (B + (lane - HS_MAXREP)*period < 0) ? -10e6 /* LARGE_NEGATIVE, RepeatStutterInfo.h:12 */ : pmf[lane];
  else if (lane == HS_NART) c[HS_NART] = -d.int_log[B+1];                    // StutterAlignerClass.cpp:64
  else if (lane <= HS_NART + HS_MAXREP){
    const int q = lane - HS_NART - 1, D = -(q+1)*period;
    c[HS_NART + 1 + q] = B+D >= 0 ? -d.int_log[B+D+1] : 0.0;                 // StutterAlignerClass.cpp:112
  }
  if (tab_len > 0){
    double* ent0 = pool + tab_off;
    double bnd = 1e300;
    // entry `lane` of the table: the lists' entries back to back in list order (the order the host wrote them in)
    int base = 0;
    for (int k = 0; k <= HS_MAXREP; k++){
      const int tail = (k == HS_MAXREP) ? B : B - (k+1)*period;
      if (tail < 0 || so->shape[k] < 0) continue;
      const int U0 = so->shape[k], n = 2 + (tail - U0 > 0 ? tail - U0 : 0);
      for (int e = lane - base; e >= 0 && e < n; e += 64){                    // (n <= 48: at most one entry per lane and list)
        double ent[3];
        table_entry(d, (e == 0) ? 0 : (e == 1 ? 1 : U0 + e - 1), U0, tail, ent);
        double* dst = ent0 + 3*(base + e);
        dst[0] = ent[0]; dst[1] = ent[1]; dst[2] = ent[2];
        if (ent[2] < bnd) bnd = ent[2];
      }
      base += n;
    }
    for (int o = 32; o > 0; o >>= 1){ const double other = __shfl_xor(bnd, o); if (other < bnd) bnd = other; }
    if (lane == 0) ent0[3*base] = bnd;
  }
  // every lane has read the option's fields by now (the loop above is the last reader); one lane makes the offsets pool-wide and clears the flag
  __builtin_amdgcn_wave_barrier();
  if (lane == 0){ so->f64_off = f64_off + (int32_t)d.f64_gen_base; so->tab_off = tab_off + (int32_t)d.f64_gen_base; so->gen = 0; }
}

// One thread per dword of grp_recs[]: the record of layout.h HS_GRP_REC_DWORDS from its descriptor, the option and the option's constants
extern "C" __global__ void __launch_bounds__(256) hs_expand_recs_kernel(const hs_dev_t* dp){
  const hs_dev_t& d = *dp;
  const int64_t gid = (int64_t)blockIdx.x*256 + threadIdx.x;
  const int64_t r = gid >> 6; const int j = (int)(gid & 63);
  if (r >= d.n_recs) return;
  const hs_recdesc_t rd = d.rec_descs[r];
  const hs_stropt_t& so = d.stropts[rd.stropt];
  int32_t v = 0;
  if (j == 0) v = (rd.flags & 0x3ff) | (so.tab_len << 10) | (rd.flags & 0x60000000);
  else if (j == 1) v = rd.re_ord;
  else if (j == 2) v = (so.tail_codes & 0xfff) | (so.B << 12);
  else if (j == 3) v = so.tab_off;
  else if (j == 4) v = rd.nd_row;
  else if (j >= 8 && j <= 8 + HS_MAXREP) v = (so.shape[j-8] & 0xffff) | (so.tab_base[j-8] << 16);
  else if (j >= 16 && j < 56) v = ((const int32_t*)(d.f64pool + so.f64_off))[j-16];
  else if (j == 56 || j == 57) v = ((const int32_t*)(d.f64pool + so.tab_off + 3*so.tab_len))[j-56];
  ((int32_t*)d.grp_recs)[r*HS_GRP_REC_DWORDS + j] = v;
}
