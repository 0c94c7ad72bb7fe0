// trace.hip — Viterbi traceback of one read against one fixed haplotype on gfx950.
//
// Replaces HapAligner::trace_optimal_aln (HapAligner.cpp:711-722): process_read(retrace_aln = true) on ONE haplotype
// (HapAligner.cpp:573-709) = full M/I/D matrices of the left and the right problem, compute_aln_logprob with its
// arg-max seed position (HapAligner.cpp:163-231), HapAligner::retrace (HapAligner.cpp:363-571) from that position, and
// stitch_alignment_trace (AlignmentTraceback.cpp:55-144).
//
//   hs_trace_fill<C>   one wavefront per (request, side) runs a systolic anti-diagonal flank sweep (a traceback is one
//                      read against one allele: nothing but its own columns to put on the lanes).  The reference keeps the full M/I/D matrices and lets retrace compare neighbours; every such
//                      comparison only involves operands the sweep has in registers when it computes the cell, so the sweep
//                      takes retrace's three decisions right there (with its 0.001-nat, direction-dependent tie rules) and
//                      HBM receives ONE BYTE per cell instead of 24 (compact rows: the interior rows of the STR block, which
//                      nothing ever reads, do not exist), plus the last column of M for the seed arg-max.  The STR row is
//                      evaluated one read column per lane by replaying the host-enumerated visiting lists, remembering the
//                      best artifact size and position (HapAligner.cpp:81-97, StutterAlignerClass.cpp:92-95,138-141).
//   hs_trace_walk      one wavefront per request: seed arg-max + log-sum-exp over the lanes, then lane 0 walks the
//                      left matrices and lane 1 the right ones, emitting the operation strings.
//
// trace_optimal_aln positions the haplotype with go_to(), which clears last_changed_, so NOTHING is reused: rows are built
// under the allele's own homopolymer context (prep.h fresh_flank_rows), unlike the forward pass.
// The host then replays the operation strings into the flat hipstr_trace_out_t (the bookkeeping of retrace that touches
// no matrix: flank sequences, SNPs, indels, STR sequence) and stitches them with the haplotype-to-reference strings.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/hipstr_hmm.h"
#include "layout.h"
#include "device_common.h"
#include "prep.h"
#include "api_internal.h"

struct hs_tside_t {               // one side of one request
  int32_t base_off, len, seed;    // the read in the base/quality pools
  int32_t side;                   // 0 = left problem (forward haplotype), 1 = right problem (reversed read and haplotype)
  int32_t n;                      // read columns on this side
  int32_t lead_off, F0;           // rows of the leading flank in this orientation
  int32_t trail_off, F2;
  int32_t stropt;
  int32_t art_off;                // ints: art_size[n] | art_pos[n]
  int32_t ops_off, ops_cap;
  int32_t pad_;
  int32_t lc_off;                 // doubles: last column of M per compact row (F0 + 1 + F2; row F0 is the STR block's last row)
  int32_t pad2_;
  int64_t dec_off;                // bytes: retrace's decisions per cell, (F0 + 1 + F2) x n: bits 0-1 from M, bit 2 from D, bit 3 from I
};

struct hs_tdev_t {
  const hs_tside_t*  sides;       // [2*n_req]: left, right of each request
  const int32_t*     items;       // side indices grouped by columns-per-lane class
  const hs_row_t*    rows;
  const hs_stropt_t* stropts;
  const hs_visit_t*  visits;
  const double*      f64pool;
  const char*        chars;
  const char*        bases;
  const char*        quals;
  const double *int_log, *qual_correct, *qual_error, *m2m, *m2i;
  double             log_thresh;
  uint8_t*           dec;
  double*            lastcol;
  int32_t*           arts;
  char*              ops;
  double*            side_prob;   // [2*n_req]
  double*            ll;          // [n_req]
  int32_t*           max_index;   // [n_req]
  int32_t*           n_ops;       // [2*n_req]
  int32_t*           str_size;    // [2*n_req] artifact size taken in the STR block, or HIPSTR_NO_STR_DATA
  int32_t*           str_pos;     // [2*n_req]
};

namespace {

constexpr double TRACE_LL_TOL = 0.001;     // HapAligner.cpp:345

template <int C> struct TraceLdsStore {     // LDS of one fill wavefront, sized by its columns-per-lane class (n <= 64*C)
  double blc[64*C], blw[64*C];
  double prev[64*C];                       // M of the row before the STR block
  double mr[64*C];                         // M of the STR block's last row
  double Mt[64*C];                         // StutterAlignerClass match table
  double Dl[64*C*HS_MAXREP];
  double In[64*C*HS_MAXREP];
  double terms[HS_NART*64];
  uint8_t rd[64*C];
  uint8_t blk[HS_MAX_STR_BP + 1];
};
struct TraceLds {                          // view of a TraceLdsStore<C>
  double *blc, *blw, *prev, *mr, *Mt, *Dl, *In, *terms;
  uint8_t *rd, *blk;
};

__device__ __forceinline__ double emit_l(const TraceLds& L, int x, uint8_t c){ return L.rd[x] == c ? L.blc[x] : L.blw[x]; }

// Closed form for a "simple" visiting list (hs_stropt_t::shape = U0 >= 0, the same evaluator as the forward pass, hmm_kernels.hip
// simple_eval): every pushed value is lp0 plus a constant.  Pushes, in the reference's order: lp0 | [0 < lim] ln(U0) + lp0 (run of U0
// equal configurations, only if U0 > 0) | lp0 once per plain offset in [U0, lim) | [stop < tail] ln(tail - stop) + lp0.  The running
// likelihood never changes along such a list, so the best position is the start (right-aligned) or the last offset visited
// (left-aligned: ties move it, StutterAlignerClass.cpp:92-95, 138-141) = `stop`.
__device__ __forceinline__ double trace_simple(const hs_tdev_t& d, double lp0, int lim, int U0, int tail, bool left_align, int& best_pos){
  const bool skip = (U0 > 0) && (lim > 0);
  const int nplain = max(0, lim - U0);
  const int stop = (lim <= 0) ? 0 : ((U0 > 0 && lim <= U0) ? U0 : lim);
  const bool has_tail = stop < tail;
  const double v_skip = d.int_log[U0] + lp0;
  const double v_tail = d.int_log[max(tail - stop, 0)] + lp0;
  double mx = lp0;
  if (skip) mx = fmax(mx, v_skip);
  if (has_tail) mx = fmax(mx, v_tail);
  double tot = 0.0;
  { const double dd = lp0 - mx; if (dd > d.log_thresh) tot += (double)(1 + nplain) * (double)f_fasterexp((float)dd); }
  if (skip){ const double dd = v_skip - mx; if (dd > d.log_thresh) tot += (double)f_fasterexp((float)dd); }
  if (has_tail){ const double dd = v_tail - mx; if (dd > d.log_thresh) tot += (double)f_fasterexp((float)dd); }
  best_pos = left_align ? stop : 0;
  return mx + (double)f_fasterlog((float)tot);
}

// StutterAlignerClass::align_pcr_insertion_reverse (StutterAlignerClass.cpp:59-104) for one read column.
__device__ double trace_ins(const hs_tdev_t& d, const TraceLds& L, const hs_stropt_t& so, const double* f64, int n, int len, int j, int D,
                            bool left_align, int& best_pos){
  const int B = so.B, p = so.period, off = n-1-j;
  const double lp0 = f64[HS_NART] + L.In[HS_MAXREP*off + D/p - 1] + (len > D ? L.Mt[off+D] : 0.0);
  const int lim = min(max(len-D, 0), B);
  if (so.shape[HS_MAXREP] >= 0) return trace_simple(d, lp0, lim, so.shape[HS_MAXREP], B, left_align, best_pos);
  Lse lse;
  double best = lp0; best_pos = 0;
  for (int pass = 0; pass < 2; pass++){
    double lp = lp0;
    lse.start(pass, lp);
    lse.push(pass, lp, d.log_thresh);
    const hs_visit_t* e = d.visits + so.ins_off;
    int ni;
    for (;; e++){
      const uint64_t meta = e->meta;
      ni = (int)(meta & 0xffff);
      if (ni >= lim) break;
      const int U = (int)((meta >> 16) & 0xffff);
      int pos = 1 + ni;
      double v;
      if ((meta >> 48) & 1) v = lp;
      else if (U == 0){
        const uint8_t ca = (uint8_t)(meta >> 32), cb = (uint8_t)(meta >> 40);
        for (int idx = -ni-p; idx >= -ni-D; idx -= p){
          lp -= emit_l(L, j+idx, ca);
          lp += emit_l(L, j+idx, cb);
        }
        v = lp;
      } else { v = e->logU + lp; pos = ni + U; }
      lse.push(pass, v, d.log_thresh);
      if (pass == 0 && (lp > best || (left_align && lp == best))){ best_pos = pos; best = lp; }
    }
    if (ni < B) lse.push(pass, d.int_log[B-ni] + lp, d.log_thresh);
  }
  return lse.finish();
}

// StutterAlignerClass::align_pcr_deletion_reverse (StutterAlignerClass.cpp:106-150) for one read column.
__device__ double trace_del(const hs_tdev_t& d, const TraceLds& L, const hs_stropt_t& so, const double* f64, int n, int len, int j, int D,
                            bool left_align, int& best_pos){
  const int B = so.B, p = so.period, off = n-1-j, q = -D/p - 1;
  double lp0 = f64[HS_NART+1+q];
  if (off + D >= 0) lp0 += L.Mt[off+D] - L.Dl[(off+D)*HS_MAXREP + q];
  else for (int k = 0; k > -len; k--) lp0 += emit_l(L, j+k, L.blk[B-1+k+D]);
  if (so.shape[q] >= 0) return trace_simple(d, lp0, len, so.shape[q], B+D, left_align, best_pos);
  Lse lse;
  double best = lp0; best_pos = 0;
  for (int pass = 0; pass < 2; pass++){
    double lp = lp0;
    lse.start(pass, lp);
    lse.push(pass, lp, d.log_thresh);
    const hs_visit_t* e = d.visits + so.del_off[q];
    int ni;
    for (;; e++){
      const uint64_t meta = e->meta;
      ni = (int)(meta & 0xffff);
      if (ni >= len) break;
      const int U = (int)((meta >> 16) & 0xffff);
      int pos = 1 + ni;
      double v;
      if (U == 0){
        const uint8_t ca = (uint8_t)(meta >> 32), cb = (uint8_t)(meta >> 40);
        lp -= emit_l(L, j-ni, ca);
        lp += emit_l(L, j-ni, cb);
        v = lp;
      } else { v = e->logU + lp; pos = ni + U; }
      lse.push(pass, v, d.log_thresh);
      if (pass == 0 && (lp > best || (left_align && lp == best))){ best_pos = pos; best = lp; }
    }
    if (ni < B+D) lse.push(pass, d.int_log[B+D-ni] + lp, d.log_thresh);
  }
  return lse.finish();
}

__device__ __forceinline__ int tri_idx(bool rev, double v1, double v2, double v3){       // HapAligner.cpp:346-358
  if (!rev){ if (v1 > v2+TRACE_LL_TOL) return (v1 > v3+TRACE_LL_TOL ? 0 : 2); return (v2 > v3+TRACE_LL_TOL ? 1 : 2); }
  if (v3 > v2+TRACE_LL_TOL) return (v3 > v1+TRACE_LL_TOL ? 2 : 0);
  return (v2 > v1+TRACE_LL_TOL ? 1 : 0);
}
__device__ __forceinline__ int pair_idx(bool rev, double v1, double v2){                 // HapAligner.cpp:360-361
  if (!rev) return (v1 > v2+TRACE_LL_TOL ? 0 : 1);
  return (v2 > v1+TRACE_LL_TOL ? 1 : 0);
}

// ------------------------------------------------------------------ decisions of one (request, side)
template <int C>
__device__ __forceinline__ void trace_fill_body(TraceLdsStore<C>& store, const hs_tdev_t* __restrict__ dp, int item_begin){
  TraceLds L;
  L.blc = store.blc; L.blw = store.blw; L.prev = store.prev; L.mr = store.mr; L.Mt = store.Mt; L.Dl = store.Dl; L.In = store.In;
  L.terms = store.terms; L.rd = store.rd; L.blk = store.blk;
  const hs_tdev_t& d = *dp;
  const int lane = threadIdx.x;
  const int si = uni(d.items[item_begin + blockIdx.x]);
  const hs_tside_t* S = d.sides + si;
  const int n = uni(S->n), len = uni(S->len), base_off = uni(S->base_off), side = uni(S->side);
  const int F0 = uni(S->F0), F2 = uni(S->F2);
  uint8_t* dec = d.dec + uni(S->dec_off);
  double* lastcol = d.lastcol + uni(S->lc_off);
  const bool rev = side != 0;
  const int nl = (n + C - 1) / C, lastlane = (n - 1) / C;
#ifdef HS_TRACE_TIME        // s_memtime per stage of the first few wavefronts (experiment builds)
  unsigned long long tt[6]; int ti = 0; tt[ti++] = __builtin_amdgcn_s_memtime();
#define HS_TTICK() (tt[ti++] = __builtin_amdgcn_s_memtime())
#else
#define HS_TTICK() ((void)0)
#endif

  uint8_t rd[C]; double blc[C], blw[C];
#pragma unroll
  for (int k = 0; k < C; k++){
    const int j = min(lane*C + k, n-1);
    const int src = base_off + (side ? len - 1 - j : j);
    const uint8_t q = (uint8_t)d.quals[src];
    rd[k] = (uint8_t)d.bases[src];
    blc[k] = d.qual_correct[q]; blw[k] = d.qual_error[q];
    if (lane*C + k < n){ L.rd[j] = rd[k]; L.blc[j] = blc[k]; L.blw[j] = blw[k]; }
  }
  const double tab_m2m = d.m2m[lane & 15], tab_m2i = d.m2i[lane & 15];

  double Mrow[C], Drow[C];
  // rows 1.. of a flank block enter at lane 0, one per step; lane t works on row (step - t); every cell is stored
  auto sweep = [&](const hs_row_t* nr, int nrows){
    const int steps = nrows + nl - 1;
    int chunk = 0;
    uint32_t rowv = (lane < nrows) ? nr[lane] : 0u;
    double oM = 0, oD = 0, oI = 0, om2m = 0, om2i = 0;
    int oMeta = 0;
    for (int st = 0; st < steps; st++){
      int meta0 = 0; double f_m2m = 0, f_m2i = 0;
      if (st < nrows){
        if (st - chunk == 64){ chunk += 64; rowv = (chunk + lane < nrows) ? nr[chunk + lane] : 0u; }
        meta0 = rdlane((int)rowv, st - chunk);
        const int h = (meta0 >> 8) & 15;
        f_m2m = rdlane(tab_m2m, h); f_m2i = rdlane(tab_m2i, h);
      }
      const int meta = shr1(meta0, oMeta);
      const double m2m = shr1(f_m2m, om2m), m2i = shr1(f_m2i, om2i);
      double mdiag = shr1(0.0, oM), ddiag = shr1(0.0, oD), ileft = shr1(0.0, oI);
      if (meta < 0){
        const uint8_t hc = (uint8_t)(meta & 0xff);
        const int64_t ro = (int64_t)((meta >> 12) & 0xfff) * n;
        oM = Mrow[C-1]; oD = Drow[C-1];
#pragma unroll
        for (int kk = 0; kk < C; kk++){
          const double e = (rd[kk] == hc) ? blc[kk] : blw[kk];
          const double c0v = ileft + m2i, c1v = mdiag + m2m, c2v = ddiag + m2i;
          double nM = e + fmax(c0v, fmax(c1v, c2v));
          double nI = blc[kk] + fmax(mdiag + T_I2M, ileft + T_I2I);
          const double dm = Mrow[kk] + T_D2M, ddl = Drow[kk] + T_D2D, ii = ileft + T_I2I, im = mdiag + T_I2M;
          const double nD = fmax(dm, ddl);
          // retrace's choices at this cell (HapAligner.cpp:536-566): from M among (I left, D diag, M diag), from D and from I
          const int code = tri_idx(rev, c0v, c2v, c1v) | (pair_idx(rev, ddl, dm) << 2) | (pair_idx(rev, ii, im) << 3);
          if (kk == 0 && lane == 0){ nM = e; nI = blc[kk]; }     // HapAligner.cpp:123-126
          mdiag = Mrow[kk]; ddiag = Drow[kk]; ileft = nI;
          Mrow[kk] = nM; Drow[kk] = nD;
          const int j = lane*C + kk;
          if (j < n) dec[ro + j] = (uint8_t)code;
          if (j == n-1) lastcol[(meta >> 12) & 0xfff] = nM;
        }
        oI = ileft;
      }
      oMeta = meta; om2m = m2m; om2i = m2i;
    }
  };

  // ---- matrix row 0 (HapAligner.cpp:33-42) and the leading flank
  const hs_row_t* lead = d.rows + uni(S->lead_off);
  {
    const uint8_t c0 = (uint8_t)(uni((int)lead[0]) & 0xff);
    double pre[C];
#pragma unroll
    for (int kk = 0; kk < C; kk++) pre[kk] = 0.0;
    double carry = 0.0;
    for (int t = 0; t < nl; t++){
      const double cin = shr1(0.0, carry);
      if (lane == t){
        double run = (t == 0) ? 0.0 : cin;
#pragma unroll
        for (int kk = 0; kk < C; kk++){ pre[kk] = run; if (lane*C + kk < n) run += blc[kk]; }
        carry = run;
      }
    }
#pragma unroll
    for (int kk = 0; kk < C; kk++){
      Mrow[kk] = ((rd[kk] == c0) ? blc[kk] : blw[kk]) + pre[kk];
      Drow[kk] = IMP;
      if (lane*C + kk == n-1) lastcol[0] = Mrow[kk];
    }
    if (lane == lastlane) d.side_prob[si] = carry;
  }
  if (F0 > 1) sweep(lead + 1, F0 - 1);
#pragma unroll
  for (int kk = 0; kk < C; kk++){ const int j = lane*C + kk; if (j < n) L.prev[j] = Mrow[kk]; }
  HS_TTICK();

  // ---- STR block (HapAligner.cpp:62-109)
  const hs_stropt_t so = d.stropts[uni(S->stropt)];
  const int B = so.B, p = so.period, nd = so.nd;
  const double* f64 = d.f64pool + so.f64_off;
  for (int x = lane; x < B; x += 64) L.blk[x] = (uint8_t)d.chars[so.seq_off + x];
  __syncthreads();
  // StutterAlignerClass::load_read (StutterAlignerClass.cpp:12-53), one table position per lane
  {
    const int maxdel = nd*p, maxins = HS_MAXREP*p;
    for (int i = lane; i < n; i += 64){
      const int e = n-1-i;
      double lp = 0.0;
      int j;
      const int lim = min(n-i, maxdel);
      for (j = 0; j < lim; j++){
        lp += emit_l(L, e-j, L.blk[B-1-j]);
        if ((j+1) % p == 0) L.Dl[i*HS_MAXREP + (j+1)/p - 1] = lp;
      }
      const int lim2 = min(n-i, B);
      for (j = maxdel; j < lim2; j++) lp += emit_l(L, e-j, L.blk[B-1-j]);
      L.Mt[i] = lp;
      double li = 0.0;
      const int lim3 = min(maxins, n-i);
      for (j = 0; j < lim3; j++){
        if (j % p < B) li += emit_l(L, e-j, L.blk[B-1-(j%p)]);
        else           li += L.blc[e-j];
        if ((j+1) % p == 0) L.In[i*HS_MAXREP + (j+1)/p - 1] = li;
      }
      for (; j < maxins; j++) if ((j+1) % p == 0) L.In[i*HS_MAXREP + (j+1)/p - 1] = li;
    }
  }
  __syncthreads();
  HS_TTICK();
  {
    const bool left_align = (side == 0);        // forward: left-align the artifact, reverse: right-align (HapAligner.cpp:69-71)
    int32_t* art_size = d.arts + uni(S->art_off);
    int32_t* art_pos = art_size + n;
    const int64_t ro = (int64_t)F0 * n;
    for (int j = lane; j < n; j += 64){
      int bsize = -10000, bpos = -1;
      double best_ll = IMP;
      for (int t = 0; t < HS_NART; t++){
        const int art = (t - HS_MAXREP)*p;
        const int alen = min(B+art, j+1);
        double term = IMP;
        int apos = -1;
        if (alen >= 0){
          double pr;
          if (art == 0) pr = L.Mt[n-1-j];
          else if (art > 0) pr = trace_ins(d, L, so, f64, n, alen, j, art, left_align, apos);
          else pr = trace_del(d, L, so, f64, n, alen, j, art, left_align, apos);
          const double pre = (j-alen < 0) ? 0.0 : L.prev[j-alen];
          term = f64[t] + pr + pre;
        }
        L.terms[t*64 + lane] = term;
        if (term > best_ll){ bsize = art; bpos = apos; best_ll = term; }
      }
      Lse lse;
      for (int pass = 0; pass < 2; pass++){
        lse.start(pass, L.terms[lane]);
        for (int t = 0; t < HS_NART; t++) lse.push(pass, L.terms[t*64 + lane], d.log_thresh);
      }
      const double v = lse.finish();
      L.mr[j] = v;
      if (j == n-1) lastcol[F0] = v;
      art_size[j] = bsize; art_pos[j] = bpos;
    }
  }
  __syncthreads();
  HS_TTICK();

  // ---- trailing flank: "stutter block must be followed by a match" (HapAligner.cpp:122-139), then the plain recurrence
  const hs_row_t* trail = d.rows + uni(S->trail_off);
  {
    const int t0row = uni((int)trail[0]);
    const uint8_t c0 = (uint8_t)(t0row & 0xff);
    const int h0 = (t0row >> 8) & 15;
    const double a_m2m = d.m2m[h0], a_m2i = d.m2i[h0];
    const int64_t ro = (int64_t)(F0 + 1) * n;
#pragma unroll
    for (int kk = 0; kk < C; kk++){
      const int j = min(lane*C + kk, n-1);
      const double e = (rd[kk] == c0) ? blc[kk] : blw[kk];
      const double mleft = L.mr[max(j - 1, 0)];                 // M of the STR block's last row, previous column
      Mrow[kk] = (j == 0) ? e : e + mleft;
      Drow[kk] = IMP;
      // the neighbours retrace would compare here: I of this row and D of the STR row are IMPOSSIBLE (HapAligner.cpp:130-139)
      const int code = tri_idx(rev, IMP + a_m2i, IMP + a_m2i, mleft + a_m2m) | (pair_idx(rev, IMP + T_D2D, L.mr[j] + T_D2M) << 2)
                     | (pair_idx(rev, IMP + T_I2I, mleft + T_I2M) << 3);
      if (lane*C + kk < n) dec[ro + j] = (uint8_t)code;
      if (lane*C + kk == n-1) lastcol[F0 + 1] = Mrow[kk];
    }
  }
  if (F2 > 1) sweep(trail + 1, F2 - 1);
#ifdef HS_TRACE_TIME
  HS_TTICK();
  if (lane == 0 && blockIdx.x < 4) printf("trace fill side %d (n %d, rows %d + %d, B %d, C %d): lead %llu tables %llu STR row %llu trail %llu cycles\n", si, n, F0, F2, B, C,
                                          tt[1]-tt[0], tt[2]-tt[1], tt[3]-tt[2], tt[4]-tt[3]);
#endif
}
// sides of up to 384 columns (C <= 6): the wavefront's tables are static LDS (under the 64 KiB a kernel may declare)
template <int C>
__global__ void __launch_bounds__(64) hs_trace_fill(const hs_tdev_t* __restrict__ dp, int item_begin){
  __shared__ TraceLdsStore<C> store;
  trace_fill_body<C>(store, dp, item_begin);
}
// The requests of ONE locus are a hundred wavefronts spread over two or three column classes: as launches of their own on side streams the
// classes did not reliably run side by side (end of round 4, rocprofv3 timeline of a 100-request call: 143 us | 245 us beside it | 233 us
// BEHIND it — two of the three streams shared a hardware queue), and a fill kernel is a chain of ~300 dependent steps whatever its size.
// One launch for the classes 1-6 instead: a workgroup picks its class from the launch order's class boundaries and takes the front of the
// largest class's LDS block.  Registers and LDS are the largest class's — irrelevant for a call that fills a fraction of the device; calls
// of many loci keep one launch per class.
struct hs_tcls_t { int32_t end[6]; };          // launch-order index one past the last side of class 1 .. 6
__global__ void __launch_bounds__(64) hs_trace_fill_mixed(const hs_tdev_t* __restrict__ dp, hs_tcls_t cls){
  __shared__ TraceLdsStore<6> store;
  const int b = (int)blockIdx.x;
  if (b < cls.end[0])      trace_fill_body<1>(*(TraceLdsStore<1>*)&store, dp, 0);
  else if (b < cls.end[1]) trace_fill_body<2>(*(TraceLdsStore<2>*)&store, dp, 0);
  else if (b < cls.end[2]) trace_fill_body<3>(*(TraceLdsStore<3>*)&store, dp, 0);
  else if (b < cls.end[3]) trace_fill_body<4>(*(TraceLdsStore<4>*)&store, dp, 0);
  else if (b < cls.end[4]) trace_fill_body<5>(*(TraceLdsStore<5>*)&store, dp, 0);
  else                     trace_fill_body<6>(store, dp, 0);
}
// longer sides, up to the forward pass' 1024 columns (C = 8, 12, 16: 75-146 KiB of tables): dynamic LDS, sized by the launch
// (hipFuncAttributeMaxDynamicSharedMemorySize).  Same body: a read of this length is rare in HipSTR's short-read data and takes the
// lower occupancy; what matters is that the drop-in does not refuse a read the forward pass accepted.
extern __shared__ double hs_trace_dyn_lds[];
template <int C>
__global__ void __launch_bounds__(64) hs_trace_fill_long(const hs_tdev_t* __restrict__ dp, int item_begin){
  trace_fill_body<C>(*(TraceLdsStore<C>*)hs_trace_dyn_lds, dp, item_begin);
}
template <int C> int launch_fill_long(int cnt, hipStream_t ks, const hs_tdev_t* d_args, int first){
  // the attribute belongs to the function on the CURRENT device (hipstr_multi_*: several devices in one process), so it is set on every
  // launch, like api.hip does for the STR kernels: a host-side call of well under a microsecond next to a rare, long kernel
  if (hipFuncSetAttribute((const void*)hs_trace_fill_long<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(TraceLdsStore<C>)) != hipSuccess) return 1;
  hipLaunchKernelGGL(hs_trace_fill_long<C>, dim3(cnt), dim3(64), sizeof(TraceLdsStore<C>), ks, d_args, first);
  return 0;
}

// ------------------------------------------------------------------ seed arg-max, total likelihood and the walk
// HapAligner::retrace (HapAligner.cpp:363-571): only the decisions; the bookkeeping is replayed on the host from `ops`.
// dec: the side's decision bytes — the wavefront's LDS copy when the side fits HS_WALK_LDS (a walk is a chain of a few hundred dependent
// one-byte loads: 60 ns each from LDS, ~700 ns from L2 / HBM), else the matrix in the workspace
__device__ void trace_walk(const hs_tdev_t& d, int si, int B, int max_index, const uint8_t* dec){
  const hs_tside_t* S = d.sides + si;
  const bool rev = S->side != 0;
  const int n = S->n, F0 = S->F0, F2 = S->F2;
  const int32_t* art_size = d.arts + S->art_off;
  const int32_t* art_pos = art_size + n;
  char* ops = d.ops + S->ops_off;
  const int cap = S->ops_cap;
  int k = 0;
  auto push = [&](char c){ if (k < cap) ops[k] = c; k++; };
  d.str_size[si] = HIPSTR_NO_STR_DATA; d.str_pos[si] = -1;
  if (max_index == 0){            // HapAligner.cpp:646-648
    for (int i = 0; i < n; i++) push('S');
    d.n_ops[si] = k;
    return;
  }
  const int blen[3] = { F0, B, F2 };
  int sblock = 0, scoord = max_index;
  for (int b = 0; b < 3; b++){ if (scoord < blen[b]){ sblock = b; break; } scoord -= blen[b]; }
  int block, base;
  if (scoord == 0){ block = sblock-1; base = blen[block]-1; } else { block = sblock; base = scoord-1; }
  const int row = max_index - 1;
  int u = row < F0 ? row : (row < F0+B ? F0 : row - B + 1);
  int seq = n-1, c = n-1, type = 0;           // 0 match, 1 deletion, 2 insertion
  while (block >= 0){
    if (block == 1){
      const int size = art_size[seq], apos = art_pos[seq];
      d.str_size[si] = size; d.str_pos[si] = apos;
      int i = 0;
      for (; i < min(seq+1, apos); i++) push('M');
      if (size < 0) for (int x = 0; x < -size; x++) push('D');
      else for (; i < min(seq+1, apos+size); i++) push('I');
      for (; i < min(B+size, seq+1); i++) push('M');
      if (B + size >= seq+1) break;            // the read does not span the STR block
      u = F0 - 1; c -= B + size; seq -= B + size; type = 0;
    } else {
      const hs_row_t* rows = d.rows + (block == 0 ? S->lead_off : S->trail_off);
      bool done = false;
      while (base >= 0 && seq >= 0){
        const int h = (rows[base] >> 8) & 15;
        push(type == 0 ? 'M' : (type == 1 ? 'D' : 'I'));
        if (type == 0){ seq--; base--; } else if (type == 1) base--; else seq--;
        if (seq == -1 || (base == -1 && block == 0)){
          while (seq != -1){ push('S'); seq--; }
          done = true;
          break;
        }
        const int code = dec[(int64_t)u*n + c];                  // taken by the sweep when it computed cell (u, c)
        if (type == 0){
          const int best = code & 3;
          if (best == 0){ type = 2; c -= 1; }
          else { type = (best == 1) ? 1 : 0; u--; c--; }
        } else if (type == 1){
          type = ((code >> 2) & 1) == 0 ? 1 : 0; u--;
        } else {
          if (((code >> 3) & 1) == 0) c--;
          else { type = 0; u--; c--; }
        }
      }
      if (done) break;
    }
    block--;
    if (block >= 0) base = blen[block] - 1;
  }
  d.n_ops[si] = k;
}

#define HS_WALK_LDS 24576          // bytes of decisions per side a walk keeps in LDS (150-bp reads x 60-row flanks: 9 KB)
__global__ void __launch_bounds__(64) hs_trace_walk(const hs_tdev_t* __restrict__ dp, int req_begin){
  const hs_tdev_t& d = *dp;
  const int lane = threadIdx.x;
  const int q = req_begin + blockIdx.x;
  const hs_tside_t* SL = d.sides + 2*q;
  const hs_tside_t* SR = SL + 1;
  const int nL = uni(SL->n), nR = uni(SR->n), F0 = uni(SL->F0), F2 = uni(SL->F2);
  const int B = uni(d.stropts[uni(SL->stropt)].B);
  const int H = F0 + B + F2;
  const double* LM = d.lastcol + uni(SL->lc_off);          // last column of M per compact row
  const double* RM = d.lastcol + uni(SR->lc_off);
  const int sp = uni(SL->base_off) + uni(SL->seed);
  const uint8_t sc = (uint8_t)d.bases[sp];
  const uint8_t sq = (uint8_t)d.quals[sp];
  const double lc = d.qual_correct[sq], lw = d.qual_error[sq];
  const double prior = -d.int_log[F0 + F2];
  const double spL = d.side_prob[2*q], spR = d.side_prob[2*q+1];
  auto cL = [&](int r){ return r < F0 ? r : (r < F0+B ? F0 : r - B + 1); };
  auto cR = [&](int r){ return r < F2 ? r : (r < F2+B ? F2 : r - B + 1); };
  auto term = [&](int x){       // compute_aln_logprob (HapAligner.cpp:163-231), one seed position
    const uint32_t rw = x < F0 ? d.rows[SL->lead_off + x] : d.rows[SL->trail_off + x - F0 - B];
    const double pe = prior + (sc == (uint8_t)(rw & 0xff) ? lc : lw);
    if (x == 0)   return (pe + spL) + RM[cR(H-2)];
    if (x == H-1) return (pe + spR) + LM[cL(H-2)];
    return (pe + LM[cL(x-1)]) + RM[cR(H-2-x)];
  };
  // the reference pushes x = 0, x = H-1, then the interior positions in order, keeping the FIRST maximum (HapAligner.cpp:184-222)
  double bv = -__builtin_huge_val(); int brank = 0x7fffffff;
  for (int kk = lane; kk < F0 + F2; kk += 64){
    const int x = kk < F0 ? kk : kk + B;
    const double v = term(x);
    const int rank = x == 0 ? 0 : (x == H-1 ? 1 : x + 1);
    if (v > bv || (v == bv && rank < brank)){ bv = v; brank = rank; }
  }
  const double mx = wave_max_d(bv);
  const int rank = (int)-wave_max_d(-(double)(bv == mx ? brank : 0x7fffffff));
  const int max_index = rank == 0 ? 0 : (rank == 1 ? H-1 : rank - 1);
  double tot = 0.0;
  for (int kk = lane; kk < F0 + F2; kk += 64){
    const int x = kk < F0 ? kk : kk + B;
    const double df = term(x) - mx;
    if (df > d.log_thresh) tot += (double)f_fasterexp((float)df);
  }
  tot = wave_sum_d(tot);
  if (lane == 0){ d.ll[q] = mx + (double)f_fasterlog((float)tot); d.max_index[q] = max_index; }
  // the two sides' decision bytes into LDS, 16 bytes per lane and step (hipstr_hmm_trace aligns the matrices to 16 bytes)
  __shared__ uint4 s_dec[2][HS_WALK_LDS/16];
  const uint8_t* decp[2];
#pragma unroll
  for (int sd = 0; sd < 2; sd++){
    const hs_tside_t* S = SL + sd;
    const int bytes = (uni(S->F0) + 1 + uni(S->F2))*uni(S->n);
    const uint8_t* g = d.dec + uni(S->dec_off);
    if (bytes <= HS_WALK_LDS){
      const uint4* g4 = (const uint4*)g;
      for (int i = lane; i < (bytes + 15)/16; i += 64) s_dec[sd][i] = g4[i];
      decp[sd] = (const uint8_t*)s_dec[sd];
    } else decp[sd] = g;
  }
  wave_lds_sync();
  if (lane == 0) trace_walk(d, 2*q, B, max_index, decp[0]);
  else if (lane == 1) trace_walk(d, 2*q+1, B, H-1-max_index, decp[1]);
}

// ------------------------------------------------------------------ host side
#define TR_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess){ \
  hipstr::api_fail(std::string(#call) + ": " + hipGetErrorString(e_)); return 1; } } while (0)

struct DevBufs {
  std::vector<void*> p;
  hipstr::Ctx* ctx = NULL;          // blocks come from (and return to) the context's cache: no hipMalloc / hipFree per call
  // the cache serves other host threads: nothing this call queued may still be running on the blocks when they go back (error paths leave early)
  hipStream_t streams[3] = {NULL, NULL, NULL}; int n_streams = 0;
  void runs_on(hipStream_t st){ for (int i = 0; i < n_streams; i++) if (streams[i] == st) return; if (n_streams < 3) streams[n_streams++] = st; }
  ~DevBufs(){ if (ctx){ for (int i = 0; i < n_streams; i++) hipStreamSynchronize(streams[i]); for (void* x : p) hipstr::dev_free(ctx, x); } }
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
    if (count) TR_HIP(hipMemcpy(*out, src, count*sizeof(T), hipMemcpyHostToDevice));
    return 0;
  }
};

// two side streams + their events per (host thread, device), created at first use and kept
struct AllelePrep {
  std::string seq[2][3];          // block sequences per orientation, in side order
  int lead_off[2], trail_off[2], stropt[2];
};

struct TraceAcc {                 // what AlignmentTrace accumulates (AlignmentTraceback.h:27-34)
  bool str_set = false; int stutter_size = 0; std::string str_seq;
  std::string flank[3];
  int flank_ins = 0, flank_del = 0;
  std::vector<std::pair<int32_t,int32_t>> indels;
  std::vector<std::pair<int32_t,char>> snps;
};

// start coordinate a (possibly reversed) haplotype block reports: HapBlock::reverse() builds HapBlock(end_-1, start_-1, ..),
// RepeatBlock::reverse() keeps (start_, end_) (HapBlock.h:123-133, RepeatBlock.h:48-58)
int32_t side_block_start(const hipstr_batch_t* b, int locus, bool rev, int bi){
  const int fb = rev ? 2-bi : bi;
  if (!rev || fb == 1) return b->blk_start[3*locus + fb];
  return b->blk_end[3*locus + fb]-1;
}

const double MIN_SNP_LOG_PROB_CORRECT = -0.0043648054;     // HapAligner.cpp:24

// The matrix-free half of HapAligner::retrace (HapAligner.cpp:363-571): the device decided the move at every step
// (`ops`); this replays them to collect flank sequences, SNPs, indels and the STR sequence.  rd/lc: the side's read
// (reversed for the right problem).  Returns false if the operation string is inconsistent.
bool replay_side(const hipstr_batch_t* b, int locus, const std::string sseq[3], bool rev, const std::string& rd, const std::vector<double>& lc,
                 const std::string& ops, int size, int apos, int block, int base, TraceAcc& tr){
  const int MATCH = 0, DEL = 1, INS = 2, NONE = -1;
  const int n = rd.size();
  int seq = n-1;
  size_t k = 0;
  while (block >= 0){
    const std::string& bs = sseq[block];
    const int blen = bs.size();
    if (block == 1){
      std::string ss;
      int i = 0;
      for (; i < std::min(seq+1, apos); i++){ ss.push_back(rd[seq-i]); k++; }
      if (size < 0) k += -size;
      else for (; i < std::min(seq+1, apos+size); i++){ ss.push_back(rd[seq-i]); k++; }
      for (; i < std::min(blen+size, seq+1); i++){ ss.push_back(rd[seq-i]); k++; }
      if (!rev) std::reverse(ss.begin(), ss.end());
      tr.str_set = true; tr.stutter_size = size; tr.str_seq = ss;
      if (blen + size >= seq+1) return k == ops.size();
      seq -= blen + size;
    } else {
      int prev = NONE;
      int32_t pos = side_block_start(b, locus, rev, block) + (rev ? -base : base);
      const int32_t inc = rev ? 1 : -1;
      int indel_seq = -1; int32_t indel_position = -1;
      std::string fs;
      const int out_block = rev ? 2-block : block;
      auto flush = [&](){ if (!rev) std::reverse(fs.begin(), fs.end()); tr.flank[out_block] += fs; };
      while (base >= 0 && seq >= 0){
        if (k >= ops.size()) return false;
        const char oc = ops[k++];
        const int type = oc == 'M' ? MATCH : (oc == 'D' ? DEL : (oc == 'I' ? INS : NONE));
        if (type == NONE) return false;
        if (type != prev){
          if (prev == DEL){
            if (rev) tr.indels.push_back(std::make_pair(indel_position, indel_position - pos));
            else     tr.indels.push_back(std::make_pair(pos+1, pos - indel_position));
          } else if (prev == INS)
            tr.indels.push_back(std::make_pair(indel_position + (rev ? 0 : 1), (int32_t)(indel_seq - seq)));
          if (type == DEL || type == INS){ indel_seq = seq; indel_position = pos; }
          prev = type;
        }
        if (type == MATCH){
          if (bs[base] != rd[seq] && lc[seq] > MIN_SNP_LOG_PROB_CORRECT) tr.snps.push_back(std::make_pair(pos, rd[seq]));
          fs.push_back(rd[seq]); seq--; base--; pos += inc;
        } else if (type == DEL){ tr.flank_del++; base--; pos += inc; }
        else { tr.flank_ins++; fs.push_back(rd[seq]); seq--; }
        if (seq == -1 || (base == -1 && block == 0)){
          k += seq + 1;                        // soft clips
          flush();
          return k == ops.size();
        }
      }
      flush();
    }
    block--;
    if (block >= 0) base = (int)sseq[block].size() - 1;
  }
  return k == ops.size();
}

// AlignmentTraceback.cpp:7-52
void stitch_dir(const char* hap_aln, int hlen, const std::string& read_aln, int h_index, int r_index, int inc, std::string& out){
  const int rlen = read_aln.size();
  while (r_index >= 0 && r_index < rlen){
    if (read_aln[r_index] == 'S'){ out.push_back('S'); r_index += inc; continue; }
    if (h_index < 0 || h_index >= hlen) return;
    if (hap_aln[h_index] == 'D'){
      if (read_aln[r_index] == 'I'){ out.push_back('M'); r_index += inc; h_index += inc; }
      else { out.push_back('D'); h_index += inc; }
    }
    else if (read_aln[r_index] == 'I'){ out.push_back('I'); r_index += inc; }
    else if (read_aln[r_index] == 'D'){
      if (hap_aln[h_index] == 'M') out.push_back('D');
      r_index += inc; h_index += inc;
    }
    else { out.push_back(hap_aln[h_index]); r_index += inc; h_index += inc; }
  }
}

bool put_pool(char* pool, int32_t* off, int idx, const std::string& s, int cap){
  off[idx+1] = off[idx];
  if ((int64_t)off[idx] + (int64_t)s.size() > cap) return false;
  memcpy(pool + off[idx], s.data(), s.size());
  off[idx+1] = off[idx] + (int32_t)s.size();
  return true;
}

struct ReqOut {                    // one request's results, built by a worker thread, copied into the flat pools in order
  bool ok = true;
  std::string hap_aln, aln_str;
  TraceAcc acc;
  int32_t aln_start = 0, aln_stop = 0;
  std::vector<std::pair<char,int32_t>> cigar;
};

}  // namespace

extern "C" int hipstr_hmm_trace(const hipstr_batch_t* b, int32_t n_req, const int32_t* req_read, const int32_t* req_allele,
                                const char* const* hap_to_ref, hipstr_trace_out_t* o){
  return hipstr_hmm_trace_seeded(b, n_req, req_read, req_allele, NULL, hap_to_ref, o);
}

extern "C" int hipstr_hmm_trace_seeded(const hipstr_batch_t* b, int32_t n_req, const int32_t* req_read, const int32_t* req_allele,
                                       const int32_t* req_seed, const char* const* hap_to_ref, hipstr_trace_out_t* o){
  hipstr::ApiTimer prof_t(hipstr::PB_TRACE);
  using hipstr::api_fail;
  if (!b || !o || n_req < 0 || (n_req > 0 && (!req_read || !req_allele))) return api_fail("null argument");
  if (b->n_loci < 1) return api_fail("hipstr_hmm_trace needs at least one locus");
  { std::string bad; if (hipstr::validate_tables(b, bad)) return api_fail(bad); }
  const bool timing = getenv("HIPSTR_TRACE_TIMING") != NULL;
  auto now = [](){ return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b){ return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t_begin = now();
  hipstr::ApiTables T;
  if (hipstr::api_device_tables(&T)) return 1;
  const hipstr::HostTables& HT = hipstr::host_tables();
  const int n_loci = b->n_loci;
  const int n_reads = b->read_off[n_loci];
  std::vector<int32_t> opt_base(n_loci + 1, 0);           // first option (over all blocks) of every locus
  for (int l = 0; l < n_loci; l++){
    if (b->period[l] < 1 || b->period[l] > 9) return api_fail("STR period must be in [1,9] (stutter_model.h:38)");
    int cnt = 0;
    for (int k = 0; k < 3; k++){
      if (b->blk_nopts[3*l+k] < 1) return api_fail("haplotype block without options");
      cnt += b->blk_nopts[3*l+k];
    }
    opt_base[l+1] = opt_base[l] + cnt;
  }
  o->hap_aln_off[0] = o->str_seq_off[0] = o->flank_seq_off[0] = o->indel_off[0] = o->snp_off[0] = 0;
  o->cigar_off[0] = o->aln_str_off[0] = 0;
  if (n_req == 0) return 0;

  // ---- per request: locus, seed, allele rows (own homopolymer context), sizes
  hipstr::Prepared P;              // only its row / STR-option pools are used
  std::vector<hs_row_t> rows;
  std::map<int64_t, int> allele_slot;                    // (locus, allele) -> entry of `alleles`
  std::vector<AllelePrep> alleles;
  std::vector<int32_t> seeds(n_req), req_locus(n_req), req_ap(n_req);
  for (int q = 0; q < n_req; q++){
    const int r = req_read[q], k = req_allele[q];
    if (r < 0 || r >= n_reads) return api_fail("request names a read outside the batch");
    const int l = (int)(std::upper_bound(b->read_off, b->read_off + n_loci + 1, r) - b->read_off) - 1;
    const int32_t* nopts = b->blk_nopts + 3*l;
    const int A = nopts[0]*nopts[1]*nopts[2];
    if (k < 0 || k >= A) return api_fail("request names an allele outside its locus");
    const bool given = req_seed && req_seed[q] != HIPSTR_SEED_AUTO;       // trace_optimal_aln's seed_base argument (HapAligner.h:93)
    const int s = given ? req_seed[q] : hipstr::calc_seed_base(b, l, r);
    if (s == -2) return api_fail("Invalid alignment seed or unrecognized CIGAR char (HapAligner.cpp:309,316)");
    if (s < 0) return api_fail("read without a seed base cannot be traced (HapAligner.cpp:586-594)");
    const int len = b->base_off[r+1] - b->base_off[r];
    if (given && (s < 1 || s > len - 2)) return api_fail("seed base must leave at least one base on either side (HapAligner.cpp:316)");
    if (s > 64*HS_MAX_COLS || len-s-1 > 64*HS_MAX_COLS) return api_fail("traceback of a read side longer than 1024 bases is not supported");
    seeds[q] = s; req_locus[q] = l;
    const int64_t key = ((int64_t)l << 32) | (uint32_t)k;
    std::map<int64_t, int>::iterator hit = allele_slot.find(key);
    if (hit != allele_slot.end()){ req_ap[q] = hit->second; continue; }
    AllelePrep ap;
    int32_t oi[3];
    hipstr::allele_options(nopts, k, oi);
    for (int x = 0, cur = opt_base[l]; x < 3; cur += nopts[x], x++){
      const int oidx = cur + oi[x];
      const std::string sq(b->seq + b->opt_off[oidx], b->opt_off[oidx+1] - b->opt_off[oidx]);
      if (sq.empty()) return api_fail(x == 1 ? "empty STR allele is not supported" : "empty flank sequence");
      if (x == 1 && sq.size() > HS_MAX_STR_BP) return api_fail("STR allele longer than 2047 bp is not supported");
      ap.seq[0][x] = sq;
      ap.seq[1][2-x] = std::string(sq.rbegin(), sq.rend());
    }
    if (ap.seq[0][0].size() + 1 + ap.seq[0][2].size() > 4095) return api_fail("flanks longer than 4094 bases in total are not supported");
    for (int sd = 0; sd < 2; sd++){
      std::vector<hs_row_t> lead, trail;
      hipstr::fresh_flank_rows(ap.seq[sd], lead, trail);
      ap.lead_off[sd] = rows.size();  rows.insert(rows.end(), lead.begin(), lead.end());
      ap.trail_off[sd] = rows.size(); rows.insert(rows.end(), trail.begin(), trail.end());
      ap.stropt[sd] = P.stropts.size();
      hipstr::append_stropt(ap.seq[sd][1], b->period[l], b->stutter + 6*l, P);
    }
    req_ap[q] = allele_slot[key] = (int)alleles.size();
    alleles.push_back(ap);
  }

  const auto t_prep = now();
  // ---- static device data
  DevBufs dev;
  dev.runs_on(T.stream);
  hs_tdev_t h; memset(&h, 0, sizeof h);
  const int total_bases = b->base_off[n_reads];
  // the reads' bases and qualities go up with the call: only those of the requested reads when they are a small part of the batch (a few
  // loci of a large batch: the whole batch's 150 MB cost more than the call's kernels)
  std::vector<int32_t> up_off(n_req);                  // a request's read in the uploaded arrays
  std::vector<char> up_bases, up_quals;
  bool compact_reads = false;
  {
    int64_t wanted = 0;
    std::vector<int32_t> at((size_t)n_reads, -1);       // read -> offset in the compact arrays
    for (int q = 0; q < n_req; q++){
      const int r = req_read[q];
      if (at[r] < 0){ at[r] = (int32_t)wanted; wanted += b->base_off[r+1] - b->base_off[r]; }
    }
    if (wanted > 0 && wanted*2 < (int64_t)total_bases){
      compact_reads = true;
      up_bases.resize((size_t)wanted); up_quals.resize((size_t)wanted);
      for (int q = 0; q < n_req; q++){
        const int r = req_read[q], len = b->base_off[r+1] - b->base_off[r];
        up_off[q] = at[r];
        memcpy(up_bases.data() + at[r], b->bases + b->base_off[r], (size_t)len);
        memcpy(up_quals.data() + at[r], b->quals + b->base_off[r], (size_t)len);
      }
    } else
      for (int q = 0; q < n_req; q++) up_off[q] = b->base_off[req_read[q]];
  }
  const char* const src_bases = compact_reads ? up_bases.data() : b->bases;
  const char* const src_quals = compact_reads ? up_quals.data() : b->quals;
  const size_t n_up = compact_reads ? up_bases.size() : (size_t)total_bases;
  hipstr::HostArena st_arena;                     // every table of the call: one pinned block, one copy
  {
    const size_t o_rows = st_arena.add(rows.data(), rows.size()*sizeof(hs_row_t)), o_so = st_arena.add(P.stropts.data(), P.stropts.size()*sizeof(hs_stropt_t)),
                 o_vis = st_arena.add(P.visits.data(), P.visits.size()*sizeof(hs_visit_t)), o_f64 = st_arena.add(P.f64pool.data(), P.f64pool.size()*sizeof(double)),
                 o_chars = st_arena.add(P.chars.data(), P.chars.size()), o_bases = st_arena.add(src_bases, n_up), o_quals = st_arena.add(src_quals, n_up);
    if (st_arena.reserve(T.ctx)) return api_fail("out of device or pinned host memory");
    if (st_arena.send(T.stream)) return 1;
    h.rows = st_arena.at<hs_row_t>(o_rows); h.stropts = st_arena.at<hs_stropt_t>(o_so); h.visits = st_arena.at<hs_visit_t>(o_vis);
    h.f64pool = st_arena.at<double>(o_f64); h.chars = st_arena.at<char>(o_chars); h.bases = st_arena.at<char>(o_bases); h.quals = st_arena.at<char>(o_quals);
  }
  h.int_log = T.int_log; h.qual_correct = T.qual_correct; h.qual_error = T.qual_error; h.m2m = T.m2m; h.m2i = T.m2i;
  h.log_thresh = HT.log_thresh;

  // ---- chunks of requests whose matrices fit the workspace budget
  // (the query costs as much as a small call's kernels: only a call whose matrices could exceed 256 MiB asks — an upper bound from the
  //  longest read and the flank lengths of the first allele is enough to tell)
  int64_t budget = (int64_t)256 << 20;                                                   // bytes of decision matrices per chunk
  {
    int64_t rough = 0;
    for (int q = 0; q < n_req; q++){
      const int r = req_read[q]; const AllelePrep& ap = alleles[req_ap[q]];
      rough += (int64_t)(ap.seq[0][0].size() + ap.seq[0][2].size() + 2)*(b->base_off[r+1] - b->base_off[r]);
    }
    if (rough > budget){
      size_t free_b = 0, total_b = 0;
      TR_HIP(hipMemGetInfo(&free_b, &total_b));
      budget = std::min<int64_t>((int64_t)8 << 30, (int64_t)(free_b / 4));
    }
  }
  if (const char* e = getenv("HIPSTR_TRACE_WS_MIB")) budget = std::max<int64_t>(1, atoll(e)) << 20;
  std::vector<hs_tside_t> sides(2*(size_t)n_req);
  std::vector<int64_t> need(n_req);
  for (int q = 0; q < n_req; q++){
    const int r = req_read[q];
    const AllelePrep& ap = alleles[req_ap[q]];
    const int len = b->base_off[r+1] - b->base_off[r];
    need[q] = 0;
    for (int sd = 0; sd < 2; sd++){
      hs_tside_t& S = sides[2*q+sd];
      memset(&S, 0, sizeof S);
      S.base_off = up_off[q]; S.len = len; S.seed = seeds[q]; S.side = sd;
      S.n = sd ? len - seeds[q] - 1 : seeds[q];
      S.lead_off = ap.lead_off[sd]; S.F0 = ap.seq[sd][0].size();
      S.trail_off = ap.trail_off[sd]; S.F2 = ap.seq[sd][2].size();
      S.stropt = ap.stropt[sd];
      S.ops_cap = S.n + S.F0 + S.F2 + (int)ap.seq[sd][1].size() + 2*HS_MAXREP*b->period[req_locus[q]] + 16;
      need[q] += ((int64_t)(S.F0 + 1 + S.F2)*S.n + 15) & ~(int64_t)15;
    }
    if (need[q] > budget) return api_fail("one traceback needs more workspace than the device offers");
  }

  // workspaces are sized for the largest chunk once and reused
  int64_t max_mat = 0, max_art = 0, max_ops = 0, max_lc = 0; int max_nq = 0;
  {
    int64_t mat = 0, art = 0, ops = 0, lc = 0; int nq = 0;
    for (int q = 0; q < n_req; q++){
      if (mat + need[q] > budget){ mat = art = ops = lc = 0; nq = 0; }
      mat += need[q]; art += 2*(int64_t)(sides[2*q].n + sides[2*q+1].n); ops += sides[2*q].ops_cap + sides[2*q+1].ops_cap; nq++;
      lc += 2*(int64_t)(sides[2*q].F0 + 1 + sides[2*q].F2);
      max_mat = std::max(max_mat, mat); max_art = std::max(max_art, art); max_ops = std::max(max_ops, ops); max_lc = std::max(max_lc, lc);
      max_nq = std::max(max_nq, nq);
    }
  }
  if (max_art > 0x7fffffff || max_ops > 0x7fffffff || max_lc > 0x7fffffff) return api_fail("too many requests for one call; split the request list");
  hs_tdev_t hc = h;
  if (dev.alloc(&hc.dec, max_mat) || dev.alloc(&hc.lastcol, max_lc) || dev.alloc(&hc.arts, max_art) || dev.alloc(&hc.side_prob, 2*(size_t)max_nq)) return 1;
  // what comes back — score, seed position, operation counts, artifact size and position per side, the operation strings — is one
  // device block with the layout of the pinned block it is copied to: one copy per chunk
  const size_t o_ll = 0, o_mxi = o_ll + (size_t)max_nq*8, o_nops = o_mxi + (size_t)max_nq*4, o_ssz = o_nops + 2*(size_t)max_nq*4, o_spos = o_ssz + 2*(size_t)max_nq*4,
               o_ops = (o_spos + 2*(size_t)max_nq*4 + 15) & ~(size_t)15, o_end = o_ops + (size_t)(max_ops ? max_ops : 1);
  char* d_res;
  if (dev.alloc(&d_res, o_end)) return 1;
  hc.ll = (double*)(d_res + o_ll); hc.max_index = (int32_t*)(d_res + o_mxi); hc.n_ops = (int32_t*)(d_res + o_nops);
  hc.str_size = (int32_t*)(d_res + o_ssz); hc.str_pos = (int32_t*)(d_res + o_spos); hc.ops = d_res + o_ops;

  const auto t_static = now();
  double ms_alloc = 0, ms_kernel = 0, ms_d2h = 0, ms_replay = 0;
  for (int q0 = 0; q0 < n_req; ){
    const auto c0 = now();
    int q1 = q0; int64_t mat = 0; int64_t n_art = 0, n_ops = 0, n_lc = 0;
    while (q1 < n_req && mat + need[q1] <= budget){
      for (int sd = 0; sd < 2; sd++){
        hs_tside_t& S = sides[2*q1+sd];
        S.dec_off = mat; mat += ((int64_t)(S.F0 + 1 + S.F2)*S.n + 15) & ~(int64_t)15;       // (16-byte pieces: hs_trace_walk copies a matrix to LDS in uint4s)
        S.lc_off = (int32_t)n_lc; n_lc += S.F0 + 1 + S.F2;
        S.art_off = (int32_t)n_art; n_art += 2*S.n;
        S.ops_off = (int32_t)n_ops; n_ops += S.ops_cap;
      }
      q1++;
    }
    const int nq = q1 - q0;
    // launch order: sides grouped by columns-per-lane class
    std::vector<int32_t> items; int cls_begin[HS_MAX_COLS+1];
    for (int cl = 1; cl <= HS_MAX_COLS; cl++){
      cls_begin[cl-1] = items.size();
      for (int si = 2*q0; si < 2*q1; si++) if ((sides[si].n + 63)/64 == cl) items.push_back(si - 2*q0);
    }
    cls_begin[HS_MAX_COLS] = items.size();
    hipstr::HostArena ch_arena;                     // this chunk's sides, launch order and argument block
    hs_tdev_t hcc = hc;
    const size_t o_sides = ch_arena.add(sides.data() + 2*q0, 2*(size_t)nq*sizeof(hs_tside_t)), o_items = ch_arena.add(items.data(), items.size()*sizeof(int32_t)),
                 o_args = ch_arena.add(&hcc, sizeof hcc);
    // (the argument block points into the arena it travels in: sizes first, then the pointers, then the copy)
    if (ch_arena.reserve(T.ctx)) return api_fail("out of device or pinned host memory");
    hcc.sides = ch_arena.at<hs_tside_t>(o_sides); hcc.items = ch_arena.at<int32_t>(o_items);
    if (ch_arena.send(T.stream)) return 1;
    const hs_tdev_t* d_args = ch_arena.at<hs_tdev_t>(o_args);
    const auto c1 = now();
    // the fill kernels of the column classes: one launch per class for calls of many loci, ONE launch for the requests of a locus or two (hs_trace_fill_mixed)
    int n_cls = 0; for (int cl = 1; cl <= 6; cl++) n_cls += (cls_begin[cl] - cls_begin[cl-1]) > 0;
    constexpr bool mixed_on = true;
    const bool mixed = mixed_on && n_cls > 1 && nq <= 4096;        // small call, several classes: one launch (hs_trace_fill_mixed)
    if (mixed){
      hs_tcls_t cls;
      for (int cl = 1; cl <= 6; cl++) cls.end[cl-1] = cls_begin[cl];
      hipLaunchKernelGGL(hs_trace_fill_mixed, dim3(cls_begin[6]), dim3(64), 0, T.stream, d_args, cls);
    }
    for (int cl = mixed ? 7 : 1; cl <= HS_MAX_COLS; cl++){
      const int cnt = cls_begin[cl] - cls_begin[cl-1];
      if (cnt == 0) continue;
      hipStream_t ks = T.stream;
      switch (cl){
        case 1: hipLaunchKernelGGL(hs_trace_fill<1>, dim3(cnt), dim3(64), 0, ks, d_args, cls_begin[cl-1]); break;
        case 2: hipLaunchKernelGGL(hs_trace_fill<2>, dim3(cnt), dim3(64), 0, ks, d_args, cls_begin[cl-1]); break;
        case 3: hipLaunchKernelGGL(hs_trace_fill<3>, dim3(cnt), dim3(64), 0, ks, d_args, cls_begin[cl-1]); break;
        case 4: hipLaunchKernelGGL(hs_trace_fill<4>, dim3(cnt), dim3(64), 0, ks, d_args, cls_begin[cl-1]); break;
        case 5: hipLaunchKernelGGL(hs_trace_fill<5>, dim3(cnt), dim3(64), 0, ks, d_args, cls_begin[cl-1]); break;
        case 6: hipLaunchKernelGGL(hs_trace_fill<6>, dim3(cnt), dim3(64), 0, ks, d_args, cls_begin[cl-1]); break;
        case 7: case 8: if (launch_fill_long<8>(cnt, ks, d_args, cls_begin[cl-1])) return api_fail("hipFuncSetAttribute (traceback LDS) failed"); break;
        case 9: case 10: case 11: case 12: if (launch_fill_long<12>(cnt, ks, d_args, cls_begin[cl-1])) return api_fail("hipFuncSetAttribute (traceback LDS) failed"); break;
        default: if (launch_fill_long<16>(cnt, ks, d_args, cls_begin[cl-1])) return api_fail("hipFuncSetAttribute (traceback LDS) failed"); break;
      }
    }
    hipLaunchKernelGGL(hs_trace_walk, dim3(nq), dim3(64), 0, T.stream, d_args, 0);
    TR_HIP(hipGetLastError());
    const auto c2 = now();
    // results through one pinned block (a pageable destination costs a staging copy per call and per array), in one copy behind the kernels
    char* hostblk = (char*)hipstr::pin_alloc(T.ctx, o_end);
    if (!hostblk) return api_fail("out of pinned host memory");
    struct PinGuard { hipstr::Ctx* c; void* p; ~PinGuard(){ hipstr::pin_free(c, p); } } pin_guard{T.ctx, hostblk};
    TR_HIP(hipMemcpyAsync(hostblk, d_res, o_ops + (size_t)(n_ops ? n_ops : 1), hipMemcpyDeviceToHost, T.stream));
    TR_HIP(hipstr::wait_stream(T.stream));
    const double* ll = (const double*)(hostblk + o_ll); const int32_t* mxi = (const int32_t*)(hostblk + o_mxi);
    const int32_t* nops = (const int32_t*)(hostblk + o_nops); const int32_t* ssz = (const int32_t*)(hostblk + o_ssz); const int32_t* spos = (const int32_t*)(hostblk + o_spos);
    const char* opsbuf_p = hostblk + o_ops;
    const auto c3 = now();

    // ---- replay (HapAligner.cpp:642-707) + stitch, request by request; independent, so spread over host threads
    std::vector<ReqOut> res(nq);
    auto work = [&](int qa, int qb){
      for (int q = qa; q < qb; q++){
        ReqOut& R = res[q-q0];
        const int r = req_read[q], sb = seeds[q], l = req_locus[q];
        const AllelePrep& ap = alleles[req_ap[q]];
        const int len = b->base_off[r+1] - b->base_off[r];
        const char* bases = b->bases + b->base_off[r];
        const char* quals = b->quals + b->base_off[r];
        const int max_index = mxi[q-q0];
        const int blen3[3] = { (int)ap.seq[0][0].size(), (int)ap.seq[0][1].size(), (int)ap.seq[0][2].size() };
        const int H = blen3[0] + blen3[1] + blen3[2];
        // left retrace, then the seed base joins the flank it sits in, then the right retrace (HapAligner.cpp:642-684)
        std::string side_ops[2];
        int seed_block = 0;
        for (int x = 0, crd = max_index; x < 3; x++){ if (crd < blen3[x]){ seed_block = x; break; } crd -= blen3[x]; }
        for (int sd = 0; sd < 2 && R.ok; sd++){
          const hs_tside_t& S = sides[2*q+sd];
          const int cnt = nops[2*(q-q0)+sd];
          if (cnt > S.ops_cap){ R.ok = false; break; }
          side_ops[sd].assign(opsbuf_p + S.ops_off, cnt);
          if (sd == 1 && seed_block != 1) R.acc.flank[seed_block].push_back(bases[sb]);
          const int mx = sd ? H-1-max_index : max_index;
          if (mx == 0) continue;                         // this side is all soft clips
          std::string rd(S.n, ' '); std::vector<double> lc(S.n);
          for (int j = 0; j < S.n; j++){
            const int src = sd ? len-1-j : j;
            rd[j] = bases[src]; lc[j] = HT.qual_correct[(uint8_t)quals[src]];
          }
          int blk = 0, crd = mx;
          for (int x = 0; x < 3; x++){ const int bl = ap.seq[sd][x].size(); if (crd < bl){ blk = x; break; } crd -= bl; }
          int block, base;
          if (crd == 0){ block = blk-1; base = (int)ap.seq[sd][block].size()-1; } else { block = blk; base = crd-1; }
          if (!replay_side(b, l, ap.seq[sd], sd != 0, rd, lc, side_ops[sd], ssz[2*(q-q0)+sd], spos[2*(q-q0)+sd], block, base, R.acc)) R.ok = false;
        }
        if (!R.ok) continue;
        R.hap_aln.assign(side_ops[0].rbegin(), side_ops[0].rend());
        R.hap_aln.push_back('M');
        R.hap_aln += side_ops[1];
        if (hap_to_ref == NULL) continue;
        // ---- stitch_alignment_trace (AlignmentTraceback.cpp:55-144)
        const std::string& full = R.hap_aln;
        const char* h2r = hap_to_ref[b->hap_off[l] + req_allele[q]];
        const int hlen = (int)strlen(h2r);
        int hap_index = max_index, hai = 0; int32_t seed_pos = b->blk_start[3*l];
        while (hap_index > 0 && hai < hlen){
          if (h2r[hai] == 'M' || h2r[hai] == 'I') hap_index--;
          if (h2r[hai] == 'M' || h2r[hai] == 'D') seed_pos++;
          hai++;
        }
        while (hai < hlen && h2r[hai] == 'D') hai++;
        int sbase = sb, rai = 0;
        while (sbase > 0 && rai < (int)full.size()){
          if (full[rai] == 'M' || full[rai] == 'I' || full[rai] == 'S') sbase--;
          rai++;
        }
        while (rai < (int)full.size() && full[rai] == 'D') rai++;
        std::string la, ra;
        stitch_dir(h2r, hlen, full, hai-1, rai-1, -1, la);
        std::reverse(la.begin(), la.end());
        stitch_dir(h2r, hlen, full, hai+1, rai+1, 1, ra);
        std::string fa = la + "M" + ra;
        for (size_t i = 0; i < fa.size(); i++){ if (fa[i] == 'I') fa[i] = 'S'; else break; }
        int32_t start = seed_pos, stop = seed_pos;
        for (char ch : la) if (ch == 'D' || ch == 'M') start--;
        for (char ch : ra) if (ch == 'D' || ch == 'M') stop++;
        R.aln_start = start; R.aln_stop = stop;
        char cc = fa[0]; int num = 1;
        for (size_t i = 1; i <= fa.size(); i++){
          if (i == fa.size() || fa[i] != cc){
            R.cigar.push_back(std::make_pair(cc, (int32_t)num));
            if (i < fa.size()){ cc = fa[i]; num = 1; }
          } else num++;
        }
        int ri = 0;
        for (char ch : fa){
          if (ch == 'S') ri++;
          else if (ch == 'M' || ch == 'I') R.aln_str.push_back(bases[ri++]);
          else R.aln_str.push_back('-');
        }
      }
    };
    // the replay is ~1.7 us of string work per request; the host threads are a persistent pool (a few microseconds to wake), so a
    // locus' hundred requests are already worth sharing: blocks of 24 requests
    int nthreads = std::max(1, std::min(hipstr::host_threads(), nq / 24));
    if (const char* e = getenv("HIPSTR_TRACE_THREADS")) nthreads = std::max(1, std::min(atoi(e), std::max(1, nq / 16)));
    auto run_parallel = [&](const std::function<void(int,int)>& fn){
      if (nthreads <= 1){ fn(q0, q1); return; }
      hipstr::parallel_for(nthreads, nthreads, [&](int t){ fn(q0 + (int)((int64_t)nq*t/nthreads), q0 + (int)((int64_t)nq*(t+1)/nthreads)); });
    };
    const auto r0t = now();
    run_parallel(work);
    const auto r1t = now();
    if (hipstr::api_profile_on()) hipstr::api_profile_add(hipstr::PB_TRACE_REPLAY, std::chrono::duration<double>(r1t - r0t).count());
    // ---- the caller's flat pools: offsets in request order (serial prefix sums), then the bytes (parallel again)
    for (int q = q0; q < q1; q++){
      const ReqOut& R = res[q-q0];
      if (!R.ok) return api_fail("internal error: inconsistent traceback operation string");
      o->hap_aln_off[q+1] = o->hap_aln_off[q] + (int32_t)R.hap_aln.size();
      o->str_seq_off[q+1] = o->str_seq_off[q] + (int32_t)(R.acc.str_set ? R.acc.str_seq.size() : 0);
      o->flank_seq_off[2*q+1] = o->flank_seq_off[2*q] + (int32_t)R.acc.flank[0].size();
      o->flank_seq_off[2*q+2] = o->flank_seq_off[2*q+1] + (int32_t)R.acc.flank[2].size();
      o->indel_off[q+1] = o->indel_off[q] + (int32_t)R.acc.indels.size();
      o->snp_off[q+1] = o->snp_off[q] + (int32_t)R.acc.snps.size();
      o->cigar_off[q+1] = o->cigar_off[q] + (int32_t)R.cigar.size();
      o->aln_str_off[q+1] = o->aln_str_off[q] + (int32_t)R.aln_str.size();
      const int64_t cap = o->cap_chars;
      if ((int64_t)o->hap_aln_off[q] + (int64_t)R.hap_aln.size() > cap || (int64_t)o->str_seq_off[q] + (int64_t)R.acc.str_seq.size() > cap ||
          (int64_t)o->flank_seq_off[2*q] + (int64_t)(R.acc.flank[0].size() + R.acc.flank[2].size()) > cap ||
          (int64_t)o->indel_off[q] + (int64_t)R.acc.indels.size() > cap || (int64_t)o->snp_off[q] + (int64_t)R.acc.snps.size() > cap ||
          (int64_t)o->cigar_off[q] + (int64_t)R.cigar.size() > cap || (int64_t)o->aln_str_off[q] + (int64_t)R.aln_str.size() > cap)
        return api_fail("hipstr_trace_out_t pools are too small (cap_chars)");
    }
    auto copy_out = [&](int qa, int qb){
      for (int q = qa; q < qb; q++){
        const ReqOut& R = res[q-q0];
        o->ll[q] = ll[q-q0]; o->max_index[q] = mxi[q-q0];
        memcpy(o->hap_aln + o->hap_aln_off[q], R.hap_aln.data(), R.hap_aln.size());
        o->stutter_size[q] = R.acc.str_set ? R.acc.stutter_size : HIPSTR_NO_STR_DATA;
        if (R.acc.str_set) memcpy(o->str_seq + o->str_seq_off[q], R.acc.str_seq.data(), R.acc.str_seq.size());
        memcpy(o->flank_seq + o->flank_seq_off[2*q], R.acc.flank[0].data(), R.acc.flank[0].size());
        memcpy(o->flank_seq + o->flank_seq_off[2*q+1], R.acc.flank[2].data(), R.acc.flank[2].size());
        o->flank_ins[q] = R.acc.flank_ins; o->flank_del[q] = R.acc.flank_del;
        for (size_t i = 0; i < R.acc.indels.size(); i++){ o->indel_pos[o->indel_off[q] + i] = R.acc.indels[i].first; o->indel_size[o->indel_off[q] + i] = R.acc.indels[i].second; }
        for (size_t i = 0; i < R.acc.snps.size(); i++){ o->snp_pos[o->snp_off[q] + i] = R.acc.snps[i].first; o->snp_base[o->snp_off[q] + i] = R.acc.snps[i].second; }
        o->aln_start[q] = R.aln_start; o->aln_stop[q] = R.aln_stop;
        for (size_t i = 0; i < R.cigar.size(); i++){ o->cigar_op[o->cigar_off[q] + i] = R.cigar[i].first; o->cigar_len[o->cigar_off[q] + i] = R.cigar[i].second; }
        memcpy(o->aln_str + o->aln_str_off[q], R.aln_str.data(), R.aln_str.size());
      }
    };
    const auto r2t = now();
    run_parallel(copy_out);
    const auto r3t = now();
    // a request's results are a dozen small heap blocks: let the threads that made them give them back, not this one at scope exit
    run_parallel([&](int qa, int qb){ for (int q = qa; q < qb; q++){ ReqOut none; std::swap(none, res[q-q0]); } });
    if (timing) fprintf(stderr, "  replay of %d: setup %.3f work %.3f offsets %.3f copy %.3f release %.3f (threads %d)\n", nq, ms(c3, r0t), ms(r0t, r1t), ms(r1t, r2t), ms(r2t, r3t), ms(r3t, now()), nthreads);
    q0 = q1;
    const auto c4 = now();
    ms_alloc += ms(c0, c1); ms_kernel += ms(c1, c2); ms_d2h += ms(c2, c3); ms_replay += ms(c3, c4);
  }
  if (timing)
    fprintf(stderr, "hipstr_hmm_trace: %d requests; prep %.3f ms, static upload+alloc %.3f, chunk upload %.3f, kernels %.3f, d2h %.3f, replay %.3f, total %.3f\n",
            n_req, ms(t_begin, t_prep), ms(t_prep, t_static), ms_alloc, ms_kernel, ms_d2h, ms_replay, ms(t_begin, now()));
  return 0;
}
