// hmm_kernels.hip — hand-written gfx950 kernels for HipSTR's read-to-haplotype HMM forward score.
//
// One workgroup = one pooled read (x a chunk of candidate alleles); two wavefronts:
// wave 0 solves the LEFT problem (read prefix vs forward haplotype), wave 1 the RIGHT
// problem (reversed read suffix vs reversed haplotype) — the split of HapAligner::process_read
// (HapAligner.cpp:606-628).  Per side:
//
//   phase A  leading flank block: max-plus M/I/D recurrence (HapAligner.cpp:114-156) swept along
//            anti-diagonals as a systolic array — lane t owns C consecutive read columns in
//            registers, haplotype rows enter at lane 0 and flow lane-to-lane with
//            v_mov_b32_dpp wave_shr:1, no LDS on the critical path.  Computed once per read
//            and cached across alleles that share the block (the reference's "reuse_alns").
//   phase B  STR block (HapAligner.cpp:62-109 + StutterAlignerClass.cpp): one lane per read
//            column, 13 artifact sizes, artifact position marginalised by replaying a
//            host-enumerated visiting list broadcast with v_readlane.
//   phase C  trailing flank block: same sweep as A, per allele.
//   combine  compute_aln_logprob (HapAligner.cpp:163-231) as a wave reduction.
//
// Arithmetic is IEEE double add/max in exactly the reference's operation order; the
// reference's float log-sum-exp approximations (mathops.cpp:86-106, fastonebigheader.h)
// are bit-replicated, so results are bit-identical to the CPU path.  Compile with
// -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "layout.h"

namespace {

constexpr double IMP = HS_IMPOSSIBLE;
constexpr double T_I2I = -1.0, T_I2M = -0.4586751453870818910216436;   // AlignmentModel.h:7-10
constexpr double T_D2D = -1.0, T_D2M = -0.4586751453870818910216436;

// ------------------------------------------------------------------ wave primitives
__device__ __forceinline__ int shr1(int old, int src){
  return __builtin_amdgcn_update_dpp(old, src, 0x138 /* wave_shr:1 */, 0xf, 0xf, false);
}
__device__ __forceinline__ double shr1(double old, double src){
  const int lo = shr1(__double2loint(old), __double2loint(src));
  const int hi = shr1(__double2hiint(old), __double2hiint(src));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int rdlane(int v, int l){ return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ double rdlane(double v, int l){
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ uint64_t rdlane(uint64_t v, int l){
  return ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
}
__device__ __forceinline__ int uni(int v){ return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void wave_lds_sync(){
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ int wave_max_i(int v){
  for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ double wave_max_d(double v){
  for (int m = 32; m >= 1; m >>= 1) v = fmax(v, __shfl_xor(v, m));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v){
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// ------------------------------------------------------------------ float approximations (bit-exact)
__device__ __forceinline__ float f_fasterexp(float p){           // fastonebigheader.h:206-218
  const float y = __fmul_rn(1.442695040f, p);
  const float c = (y < -126.0f) ? -126.0f : y;
  return __uint_as_float((uint32_t)__fmul_rn(8388608.0f, __fadd_rn(c, 126.94269504f)));
}
__device__ __forceinline__ float f_fasterlog(float x){           // fastonebigheader.h:348-358
  float y = (float)__float_as_uint(x);
  y = __fmul_rn(y, 8.2629582881927490e-8f);
  return __fsub_rn(y, 87.989971088f);
}

// streaming form of fast_log_sum_exp(vector) (mathops.cpp:97-106): pass 0 finds the max,
// pass 1 accumulates.  The float terms are summed in double, which is exact for any order.
struct Lse {
  double mx, tot;
  __device__ __forceinline__ void start(int pass, double first){ if (pass == 0) mx = first; else tot = 0.0; }
  __device__ __forceinline__ void push(int pass, double v, double thr){
    if (pass == 0) mx = fmax(mx, v);
    else { const double d = v - mx; if (d > thr) tot += (double)f_fasterexp((float)d); }
  }
  __device__ __forceinline__ double finish() const { return mx + (double)f_fasterlog((float)tot); }
};

// ------------------------------------------------------------------ per-workgroup view of LDS
struct Lds {
  double2* bq;      // [Lc] (log P(correct), log P(error)) per read column, side regions back to back
  double*  rowP;    // [Lc] M of the haplotype row preceding the STR block
  double*  Mt;      // [Lc] StutterAligner match_probs_
  double*  MR;      // [Lc] M of the STR block's last row
  double*  Dl;      // [6][Lc] StutterAligner del_probs_
  double*  lastcol; // [2][lds_flank+1] last read column of M per compact haplotype row
  double*  misc;    // [4] side_prob L/R
  uint8_t* rd;      // [Lc] read bases
  int Lc, nflank;
};

struct Side {       // wave-uniform description of one side of one read
  int n;            // read columns
  int o;            // offset of this side's region in the per-column LDS arrays
  int side;
};

__device__ __forceinline__ double emit(uint8_t r, uint8_t c, double2 q){ return r == c ? q.x : q.y; }

// ------------------------------------------------------------------ systolic sweep over NORMAL flank rows
// Rows rows[0..nrows) enter at lane 0, one per step; lane t works on row (step - t).
// Mrow/Drow hold M/D of the previous haplotype row for this lane's C columns and are
// updated in place; lastcol[u] receives M[row][n-1].
template <int C>
__device__ __forceinline__ void sweep(const hs_dev_t& d, const hs_row_t* __restrict__ rows, int nrows, const Side& s,
                                      const uint8_t (&rd)[C], const double (&blc)[C], const double (&blw)[C],
                                      double (&Mrow)[C], double (&Drow)[C], double* lastcol, double tab_m2m, double tab_m2i){
  if (nrows <= 0 || (d.debug_skip & 2)) return;
  const int lane = threadIdx.x & 63;
  const int nl = (s.n + C - 1) / C, lastlane = (s.n - 1) / C, klast = (s.n - 1) % C;
  const int steps = nrows + nl - 1;
  int chunk = 0;
  uint32_t rowv = (lane < nrows) ? rows[lane] : 0u;
  double oM = 0, oD = 0, oI = 0, om2m = 0, om2i = 0;
  int oMeta = 0;
  for (int st = 0; st < steps; st++){
    int meta0 = 0; double f_m2m = 0, f_m2i = 0;
    if (st < nrows){
      if (st - chunk == 64){ chunk += 64; rowv = (chunk + lane < nrows) ? rows[chunk + lane] : 0u; }
      meta0 = rdlane((int)rowv, st - chunk);
      const int h = (meta0 >> 8) & 15;
      f_m2m = rdlane(tab_m2m, h); f_m2i = rdlane(tab_m2i, h);
    }
    const int meta = shr1(meta0, oMeta);
    const double m2m = shr1(f_m2m, om2m), m2i = shr1(f_m2i, om2i);
    double mdiag = shr1(0.0, oM), ddiag = shr1(0.0, oD), ileft = shr1(0.0, oI);
    if (meta < 0){   // valid bit is the sign bit
      const uint8_t hc = (uint8_t)(meta & 0xff);
      oM = Mrow[C-1]; oD = Drow[C-1];
      double mlast = 0;
#pragma unroll
      for (int k = 0; k < C; k++){
        const double e = (rd[k] == hc) ? blc[k] : blw[k];
        const double c0 = ileft + m2i, c1 = mdiag + m2m, c2 = ddiag + m2i;
        double nM = e + fmax(c0, fmax(c1, c2));
        double nI = blc[k] + fmax(mdiag + T_I2M, ileft + T_I2I);
        const double nD = fmax(Mrow[k] + T_D2M, Drow[k] + T_D2D);
        if (k == 0 && lane == 0){ nM = e; nI = blc[k]; }     // HapAligner.cpp:123-126
        mdiag = Mrow[k]; ddiag = Drow[k]; ileft = nI;
        Mrow[k] = nM; Drow[k] = nD;
        if (k == klast) mlast = nM;
      }
      oI = ileft;
      if (lane == lastlane) lastcol[(meta >> 12) & 0xfff] = mlast;
    }
    oMeta = meta; om2m = m2m; om2i = m2i;
  }
}

// ------------------------------------------------------------------ phase B: the STR block
struct StrCtx {
  int B, p, nd;
  int blkv;          // lane x holds block chars 4x..4x+3
  double cst;        // lane t<20 holds f64pool[f64_off+t]: pmf[13] | prior_ins | prior_del[6]
  const hs_visit_t* visits;
  const hs_stropt_t* so;
};
__device__ __forceinline__ uint8_t blk_at(const StrCtx& c, int x){    // x wave-uniform
  return (uint8_t)(((uint32_t)rdlane(c.blkv, x >> 2) >> ((x & 3)*8)) & 0xff);
}

// Marginalisation over the artifact position (StutterAlignerClass.cpp:59-104 insertion, :106-150 deletion).
// The loop over block offsets is the same for every read column, so the host enumerated it (hs_visit_t) and
// the wave replays it in lock step; a lane drops out once the offset reaches its own bound `lim`.
//   lp0     value for the artifact at the block's right end (position 0)
//   nsub    read bases whose emission changes when the artifact moves one base left: D/p for an insertion, 1 for a deletion
//   stride  distance between those bases: p for an insertion, 0 for a deletion
//   tail    number of remaining equal-likelihood configurations is (tail - offset): B (insertion) or B+D (deletion)
__device__ __forceinline__ double visit_eval(const hs_dev_t& d, const Lds& L, int o, int j, double lp0, int lim, int limmax,
                                             const hs_visit_t* __restrict__ list, int llen, int nsub, int stride, int tail){
  const int lane = threadIdx.x & 63;
  Lse acc;
  for (int pass = 0; pass < 2; pass++){
    double lp = lp0;
    acc.start(pass, lp0);
    acc.push(pass, lp0, d.log_thresh);
    int nistop = 0; bool stopped = false;
    int vbase = 0;
    hs_visit_t vv = list[min(lane, llen-1)];
    for (int v = 0; v < llen; v++){
      if (v - vbase == 64){ vbase += 64; vv = list[min(vbase + lane, llen-1)]; }
      const uint64_t meta = rdlane(vv.meta, v - vbase);
      const int ni = (int)(meta & 0xffff);
      if (ni >= limmax){ if (!stopped) nistop = ni; break; }
      const bool act = ni < lim;
      if (!act && !stopped){ nistop = ni; stopped = true; }
      const int U = (int)((meta >> 16) & 0xffff);
      if ((meta >> 48) & 1){ if (act) acc.push(pass, lp, d.log_thresh); }
      else if (U == 0){
        const uint8_t ca = (uint8_t)(meta >> 32), cb = (uint8_t)(meta >> 40);
        for (int m = 1; m <= nsub; m++){
          const int pos = o + max(j - ni - m*stride, 0);
          const uint8_t r = L.rd[pos]; const double2 bq = L.bq[pos];
          if (act){ lp -= emit(r, ca, bq); lp += emit(r, cb, bq); }
        }
        if (act) acc.push(pass, lp, d.log_thresh);
      } else {
        const double logU = rdlane(vv.logU, v - vbase);
        if (act) acc.push(pass, logU + lp, d.log_thresh);
      }
    }
    if (nistop < tail) acc.push(pass, d.int_log[tail - nistop] + lp, d.log_thresh);
  }
  return acc.finish();
}

// Fills L.MR[o + j] = M[R][j] for every column of the side (HapAligner.cpp:62-104) and lastcol[STR slot].
__device__ __forceinline__ void phase_str(const hs_dev_t& d, const Lds& L, const Side& s, int str_opt, int flead){
  const int lane = threadIdx.x & 63;
  StrCtx c;
  c.so = d.stropts + str_opt;
  c.B = uni(c.so->B); c.p = uni(c.so->period); c.nd = uni(c.so->nd);
  c.visits = d.visits;
  c.blkv = ((const int*)(d.chars + uni(c.so->seq_off)))[min(lane, (c.B + 3)/4 - 1)];
  c.cst = d.f64pool[uni(c.so->f64_off) + min(lane, 19)];
  const int B = c.B, p = c.p, n = s.n, o = s.o;
  const int ncyc = (n + 63) / 64;
  const hs_visit_t* ins_list = d.visits + uni(c.so->ins_off);
  const int ins_len = uni(c.so->ins_len);

  // --- StutterAlignerClass::load_read (StutterAlignerClass.cpp:12-53): match_probs_ and del_probs_
  for (int kk = 0; kk < ncyc; kk++){
    const int j = min(lane + 64*kk, n-1);
    double lp = 0.0;
    const int tmax = min(B, n);
    const int ndp = c.nd * p;
    for (int t = 0; t < tmax; t++){
      const uint8_t bc = blk_at(c, B-1-t);
      const int pos = o + max(j - t, 0);
      const double e = emit(L.rd[pos], bc, L.bq[pos]);
      if (t <= j){
        lp += e;
        if (t < ndp && (t+1) % p == 0) L.Dl[((t+1)/p - 1)*L.Lc + o + j] = lp;
      }
    }
    L.Mt[o + j] = lp;
  }
  wave_lds_sync();

  for (int kk = 0; kk < ncyc; kk++){
    const int jraw = lane + 64*kk;
    const bool actj = jraw < n;
    const int j = min(jraw, n-1);
    const int jmax = min(n-1, 64*kk + 63);        // largest column of this chunk: bounds are monotone in j
    // The 13 artifact terms are produced by ONE runtime loop (no artifact, insertions +p..+6p, deletions -p..-6p) and kept
    // in a rotating register window; fast_log_sum_exp (mathops.cpp:97-106) does not depend on their order.
    double terms[HS_NART];
#pragma unroll
    for (int t = 0; t < HS_NART; t++) terms[t] = IMP;
    double li = 0.0;                               // running ins_probs_ sum (StutterAlignerClass.cpp:40-51)
    for (int it = 0; it < HS_NART; it++){
      double term = IMP;
      if (it == 0){                                // no artifact (StutterAlignerClass.cpp:55-57)
        const int len = min(B, j + 1);
        const double pre = (j - len < 0) ? 0.0 : L.rowP[o + j - len];
        term = (rdlane(c.cst, HS_MAXREP) + L.Mt[o + j]) + pre;
      } else if (it <= HS_MAXREP){                 // insertion of D = (q+1) p
        const int q = it - 1, D = (q+1)*p;
        for (int m = 0; m < p; m++){               // extend the insertion table by one repeat unit
          const int t = q*p + m;
          const int pos = o + max(j - t, 0);
          const double2 bq = L.bq[pos];
          const double e = (m < B) ? emit(L.rd[pos], blk_at(c, B-1-m), bq) : bq.x;
          if (t <= j) li += e;
        }
        const int len = min(B + D, j + 1);
        const double lp0 = (rdlane(c.cst, 13) + li) + ((len > D) ? L.Mt[o + max(j - D, 0)] : 0.0);
        const int lim = actj ? min(max(0, len - D), B) : 0;
        const int limmax = min(max(0, min(B + D, jmax + 1) - D), B);
        const double S = visit_eval(d, L, o, j, lp0, lim, limmax, ins_list, ins_len, q+1, p, B);
        const double pre = (j - len < 0) ? 0.0 : L.rowP[o + j - len];
        term = (rdlane(c.cst, HS_MAXREP + 1 + q) + S) + pre;
      } else {                                     // deletion of aD = (q+1) p bases
        const int q = it - 1 - HS_MAXREP, aD = (q+1)*p;
        if (B - aD >= 0){
          const int len = min(B - aD, j + 1);
          double lp0 = rdlane(c.cst, 14 + q);
          const bool direct = (j + aD <= n - 1);
          if (direct) lp0 += L.Mt[o + j + aD] - L.Dl[q*L.Lc + o + j + aD];
          if (jmax + aD > n - 1){                  // some column of the chunk ends within aD of the read end
            const int tmax = min(B - aD, n);
            for (int t = 0; t < tmax; t++){
              const uint8_t bc = blk_at(c, B-1-t-aD);
              const int pos = o + max(j - t, 0);
              const double e = emit(L.rd[pos], bc, L.bq[pos]);
              if (!direct && t < len) lp0 += e;
            }
          }
          const int lim = actj ? len : 0;
          const int limmax = min(B - aD, jmax + 1);
          const double S = visit_eval(d, L, o, j, lp0, lim, limmax, d.visits + uni(c.so->del_off[q]), uni(c.so->del_len[q]), 1, 0, B - aD);
          const double pre = (j - len < 0) ? 0.0 : L.rowP[o + j - len];
          term = (rdlane(c.cst, HS_MAXREP - 1 - q) + S) + pre;
        }
      }
#pragma unroll
      for (int t = 0; t + 1 < HS_NART; t++) terms[t] = terms[t+1];
      terms[HS_NART-1] = term;
    }
    Lse acc;
    for (int pass = 0; pass < 2; pass++){
      acc.start(pass, terms[0]);
#pragma unroll
      for (int t = 0; t < HS_NART; t++) acc.push(pass, terms[t], d.log_thresh);
    }
    if (actj){
      const double mr = acc.finish();
      L.MR[o + j] = mr;
      if (j == n-1) L.lastcol[s.side*(L.nflank+1) + flead] = mr;
    }
  }
  wave_lds_sync();
}

// One flank block of one side: is_lead selects phase A (row 0 + leading flank, result cached in
// LDS across alleles) or phase C (trailing flank of the current allele).
template <int C>
__device__ __forceinline__ void run_flank(const hs_dev_t& d, const Lds& L, const Side& s, int rowset_id, bool is_lead,
                                          double tab_m2m, double tab_m2i){
  const int lane = threadIdx.x & 63;
  uint8_t rd[C]; double blc[C], blw[C];
#pragma unroll
  for (int k = 0; k < C; k++){
    const int j = min(lane*C + k, s.n-1);
    rd[k] = L.rd[s.o + j];
    const double2 q = L.bq[s.o + j];
    blc[k] = q.x; blw[k] = q.y;
  }
  hs_rowset_t rs;
  rs.off = uni(d.rowsets[rowset_id].off); rs.len = uni(d.rowsets[rowset_id].len);
  const hs_row_t* rows = d.rows + rs.off;
  const uint32_t r0 = rows[0];
  const uint8_t c0 = (uint8_t)(r0 & 0xff);
  double* lastcol = L.lastcol + s.side*(L.nflank+1);
  double Mrow[C], Drow[C];
  if (is_lead){
    // matrix row 0 (HapAligner.cpp:33-42).  left_prob is a strictly sequential sum in the reference,
    // so it is passed lane to lane rather than scanned.
    const int nl = (s.n + C - 1) / C;
    double pre[C];
#pragma unroll
    for (int k = 0; k < C; k++) pre[k] = 0.0;
    double carry = 0.0;
    for (int t = 0; t < nl; t++){
      const double cin = shr1(0.0, carry);
      if (lane == t){
        double run = (t == 0) ? 0.0 : cin;
#pragma unroll
        for (int k = 0; k < C; k++){ pre[k] = run; if (lane*C + k < s.n) run += blc[k]; }
        carry = run;
      }
    }
#pragma unroll
    for (int k = 0; k < C; k++){
      Mrow[k] = ((rd[k] == c0) ? blc[k] : blw[k]) + pre[k];
      Drow[k] = IMP;
    }
    if (lane == (s.n-1)/C) L.misc[s.side] = carry;
  } else {
    // "stutter block must be followed by a match" (HapAligner.cpp:122-139)
#pragma unroll
    for (int k = 0; k < C; k++){
      const int j = min(lane*C + k, s.n-1);
      const double e = (rd[k] == c0) ? blc[k] : blw[k];
      Mrow[k] = (j == 0) ? e : e + L.MR[s.o + j - 1];
      Drow[k] = IMP;
    }
  }
  if (lane == (s.n-1)/C){
    double v = 0;
#pragma unroll
    for (int k = 0; k < C; k++) if (k == (s.n-1)%C) v = Mrow[k];
    lastcol[(r0 >> 12) & 0xfff] = v;
  }
  sweep<C>(d, rows + 1, rs.len - 1, s, rd, blc, blw, Mrow, Drow, lastcol, tab_m2m, tab_m2i);
  if (is_lead){
#pragma unroll
    for (int k = 0; k < C; k++){ const int j = lane*C + k; if (j < s.n) L.rowP[s.o + j] = Mrow[k]; }
    wave_lds_sync();
  }
}

}  // namespace

extern __shared__ double hs_lds_raw[];

// LDS bytes per workgroup for a batch whose longest read has lds_len bases and whose longest
// allele has lds_flank flank bases (keep in sync with the carve below).
extern "C" size_t hs_forward_lds_bytes(int lds_len, int lds_flank){
  const size_t Lc = ((size_t)lds_len + 3) & ~(size_t)1;
  return Lc*16 + Lc*8*3 + Lc*8*HS_MAXREP + 2*((size_t)lds_flank+1)*8 + 4*8 + ((Lc + 15) & ~(size_t)15);
}

#ifndef HS_MIN_WAVES
#define HS_MIN_WAVES 5   // waves per SIMD the register allocator must leave room for (measured best: profiles/r01_notes.md)
#endif
extern "C" __global__ void __launch_bounds__(128, HS_MIN_WAVES)
hs_forward_kernel(const hs_dev_t* __restrict__ dp){
  const hs_dev_t& d = *dp;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = d.active[blockIdx.x];
  const hs_read_t rdesc = d.reads[r];
  const hs_locus_t loc = d.loci[rdesc.locus];

  Lds L;
  L.Lc = (d.lds_len + 3) & ~1; L.nflank = d.lds_flank;
  L.bq = (double2*)hs_lds_raw;
  L.rowP = (double*)(L.bq + L.Lc);
  L.Mt = L.rowP + L.Lc; L.MR = L.Mt + L.Lc; L.Dl = L.MR + L.Lc;
  L.lastcol = L.Dl + HS_MAXREP*L.Lc;
  L.misc = L.lastcol + 2*(L.nflank+1);
  L.rd = (uint8_t*)(L.misc + 4);

  const int nL = rdesc.seed, nR = rdesc.len - rdesc.seed - 1;
  Side s; s.side = w; s.n = w ? nR : nL; s.o = w ? ((nL + 1) & ~1) : 0;
  // stage this side of the read: bases, log P(correct), log P(error); the right side reversed (HapAligner.cpp:606-609)
  for (int j = lane; j < s.n; j += 64){
    const int src = rdesc.base_off + (w ? rdesc.len - 1 - j : j);
    const uint8_t q = (uint8_t)d.quals[src];
    L.rd[s.o + j] = (uint8_t)d.bases[src];
    L.bq[s.o + j] = make_double2(d.qual_correct[q], d.qual_error[q]);
  }
  const double tab_m2m = d.m2m[lane & 15], tab_m2i = d.m2i[lane & 15];
  const uint8_t seed_c = (uint8_t)d.bases[rdesc.base_off + rdesc.seed];
  const uint8_t seed_q = (uint8_t)d.quals[rdesc.base_off + rdesc.seed];
  const double seed_lc = d.qual_correct[seed_q], seed_lw = d.qual_error[seed_q];
  wave_lds_sync();

  const int C = (s.n + 63) >> 6;
  const int k0 = blockIdx.y * d.allele_chunk, k1 = min(loc.n_alleles, k0 + d.allele_chunk);
  double* out = d.aln_probs + loc.out_off + (int64_t)(r - loc.read_begin)*loc.n_alleles;
  int cur_lead = -1;
  for (int k = k0; k < k1; k++){
    const hs_allele_t* alp = d.alleles + loc.hap_begin + k;
    if (!uni(alp->realign)) continue;
    const int lead_id = uni(alp->lead_rows[w]), trail_id = uni(alp->trail_rows[w]), str_id = uni(alp->str_opt[w]);
    const bool run_lead = lead_id != cur_lead;
    cur_lead = lead_id;
    for (int ph = run_lead ? 0 : 1; ph < 2; ph++){
      if (ph == 1 && !(d.debug_skip & 1)) phase_str(d, L, s, str_id, uni(d.rowsets[lead_id].len));
      const int rsid = ph == 0 ? lead_id : trail_id;
      switch (C){
        case 1:  run_flank<1>(d, L, s, rsid, ph == 0, tab_m2m, tab_m2i); break;
        case 2:  run_flank<2>(d, L, s, rsid, ph == 0, tab_m2m, tab_m2i); break;
        case 3:  run_flank<3>(d, L, s, rsid, ph == 0, tab_m2m, tab_m2i); break;
        default: run_flank<4>(d, L, s, rsid, ph == 0, tab_m2m, tab_m2i); break;
      }
    }
    __syncthreads();
    if (w == 0 && !(d.debug_skip & 4)){
      // compute_aln_logprob (HapAligner.cpp:163-231): log-sum-exp over the haplotype positions the seed base can sit on
      const int N = uni(alp->n_flank);
      const hs_rowset_t lead = d.rowsets[alp->lead_rows[0]], trail = d.rowsets[alp->trail_rows[0]];
      const int F0 = uni(lead.len);
      const double prior = -d.int_log[N];
      const double* lcL = L.lastcol; const double* lcR = L.lastcol + (L.nflank+1);
      Lse acc;
      for (int pass = 0; pass < 2; pass++){
        if (pass == 0) acc.mx = -1.0e300; else acc.tot = 0.0;
        for (int y = lane; y < N; y += 64){
          const uint8_t hc = (uint8_t)((y < F0 ? d.rows[lead.off + y] : d.rows[trail.off + y - F0]) & 0xff);
          const double e = (seed_c == hc) ? seed_lc : seed_lw;
          double a, b;
          if (y == 0)        { a = L.misc[0];   b = lcR[N-1]; }
          else if (y == N-1) { a = L.misc[1];   b = lcL[N-1]; }
          else if (y < F0)   { a = lcL[y-1];    b = lcR[N-1-y]; }
          else               { a = lcL[y];      b = lcR[N-2-y]; }
          acc.push(pass, ((prior + e) + a) + b, d.log_thresh);
        }
        if (pass == 0) acc.mx = wave_max_d(acc.mx); else acc.tot = wave_sum_d(acc.tot);
      }
      if (lane == 0) out[k] = acc.finish();
    }
    __syncthreads();
  }
}
