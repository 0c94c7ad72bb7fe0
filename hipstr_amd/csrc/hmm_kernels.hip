// hmm_kernels.hip — hand-written gfx950 kernels for HipSTR's read-to-haplotype HMM forward score.
//
// A pooled read is split at its seed base into a LEFT problem (read prefix vs the forward haplotype) and a
// RIGHT problem (reversed read suffix vs the reversed haplotype) — HapAligner::process_read
// (HapAligner.cpp:606-628).  Each side walks the haplotype [leading flank | STR block | trailing flank].
// The three block kinds have very different register and LDS appetites, so each is its own kernel, tuned
// separately, handing a few hundred bytes per (read, allele, side) to the next one through HBM workspaces:
//
//   hs_col_kernel             per-column emission logs of every read in side orientation.
//   hs_lead_kernel_coop       leading flank: matrix row 0 + max-plus M/I/D recurrence (HapAligner.cpp:33-42, 114-156), once
//                             per read and distinct flank, shared by all alleles (what the reference's "reuse_alns" does one
//                             allele at a time), with READS as lanes: the flank rows are wave-uniform, a lane carries the
//                             M/I/D of its own read, the matrix is swept in bands of rows held in registers, a band per
//                             wavefront of the workgroup.                                  -> rowP, last column, side_prob
//   hs_nd_kernel              STR block (HapAligner.cpp:62-109 + StutterAlignerClass.cpp): one lane per read column, 13 artifact
//   hs_str_group_kernel_p     sizes; the artifact position marginalised by a tabulated closed form (periodic blocks: _p, reads of a
//   hs_str_group_kernel[_pw,  locus side packed into one workgroup's lanes), by the piecewise closed forms of interrupted repeats
//    _rp], hs_str_kernel[_generic]   (_pw, _rp) or by a replay of a host-enumerated visiting list (a workgroup per read).        -> MR
//   hs_trail_kernel_coop      trailing flank: the same banded sweep with ALLELES as lanes: all alleles of a locus share the
//                             read and (per group) the flank rows, so a lane needs no neighbour at all; the workgroup streams
//                             through its items without a barrier (round 6).                                  -> last column
//   hs_flank_systolic         the flank blocks of small launches: a wavefront per alignment, rows as lanes.
//   hs_combine_kernel         compute_aln_logprob (HapAligner.cpp:163-231): log-sum-exp over seed positions.
//
// Arithmetic is IEEE double add/max in exactly the reference's operation order; the reference's float
// log-sum-exp approximations (mathops.cpp:86-106, fastonebigheader.h) are bit-replicated, so results are
// bit-identical to the CPU path.  Compile with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <type_traits>

#ifdef __HIP_DEVICE_COMPILE__
#define HS_P(T) __attribute__((address_space(1))) T*      // layout.h: the argument block's pointers are global-address-space pointers here
#endif
#include "layout.h"
#include "device_common.h"

namespace {

// Decoded view of one (active read, side): everything wave-uniform.
struct SideView {
  int ai, r, side, n, nL, len, base_off;
  const hs_locus_t* loc;
  int64_t ws_mr, ws_lt, ws_lead, ws_col;
};
__device__ __forceinline__ SideView side_view(const hs_dev_t& d, int ai, int side){
  SideView v;
  v.ai = ai; v.side = side;
  v.r = uni(d.active[ai]);
  const hs_read_t* rd = d.reads + v.r;
  v.len = uni(rd->len); v.nL = uni(rd->seed); v.base_off = uni(rd->base_off);
  v.n = side ? v.len - v.nL - 1 : v.nL;
  v.loc = d.loci + uni(rd->locus);
  v.ws_mr = uni(d.ws[ai].mr); v.ws_lt = uni(d.ws[ai].lt); v.ws_lead = uni(d.ws[ai].lead[side]); v.ws_col = uni(d.ws[ai].col);
  return v;
}
// lead workspace record of (side, slot): rowP[n] | last column of the leading-flank rows [lead_flank] | side_prob
__device__ __forceinline__ double* lead_record(const hs_dev_t& d, const SideView& v, int slot){
  const int stride = v.n + uni(v.loc->lead_flank[v.side]) + 1;
  return d.ws_lead + v.ws_lead + (int64_t)slot*stride;
}

// ------------------------------------------------------------------ flank blocks: what the sweeps share
// Trailing flank: work item = (read side, group of <= 64 alleles sharing the trailing-flank rowset), lane = allele.  Every quantity that
// depends on the read column or on the haplotype row — base, log P(correct/error), flank base, transition logs, hence the emission — is
// wave-uniform; a lane only carries its own M / I / D values, so the recurrence (HapAligner.cpp:144-153) needs no cross-lane traffic.
// When a group has <= 32 alleles, 64/npad reads of the same locus and side are packed into one wavefront.  Leading flank: the lanes are
// READS of one locus and side (the rows are still wave-uniform); the first band's top boundary is matrix row 0 (HapAligner.cpp:33-42:
// emission plus the running sum of log P(correct), which also yields side_prob at the read's last column), and the last band writes M
// of its bottom row — rowP, what the STR block starts from — for every column.  The sweeps themselves: band_sweep_coop below (the serial
// one-wavefront-per-item form of rounds 1-2, HIPSTR_FLANK_COOP=0, went in round 6: no test used it).
// ------------------------------------------------------------------ leading flank: reads as lanes, the same banded sweep
// Per-column emission logs of an active read in side orientation (left side columns, then the reversed right side, HapAligner.cpp:606-609):
// [len-1][3] doubles = log P(correct), log P(error), base.  One wavefront per read; read by hs_lead_kernel and hs_trail_kernel.
__global__ void __launch_bounds__(64) hs_col_kernel(const hs_dev_t* __restrict__ dp, int active_begin, int n_clear){
  const hs_dev_t& d = *dp;
  // the re-do flags and the item counters of the pass (a hipMemsetAsync before: two fill kernels and a gap in every small call).  Every chunk's
  // launch clears all of them: the chunks run one after the other on one stream, an earlier chunk's flags and counters are used up by then
  for (int i = blockIdx.x*64 + threadIdx.x; i < n_clear; i += gridDim.x*64) d.redo[i] = 0;
  const int ai = active_begin + blockIdx.x;
  const hs_read_t rd = d.reads[d.active[ai]];
  double* col = d.ws_col + d.ws[ai].col;
  for (int c = threadIdx.x; c < rd.len - 1; c += 64){
    const int src = rd.base_off + (c < rd.seed ? c : rd.len - 1 - (c - rd.seed));
    const uint8_t q = (uint8_t)d.quals[src];
    col[3*c] = d.qual_correct[q]; col[3*c+1] = d.qual_error[q]; col[3*c+2] = (double)(uint8_t)d.bases[src];
  }
}

// ------------------------------------------------------------------ flank blocks, cooperative form: bands as the wavefronts of one workgroup
// The banded sweep above runs the bands of an item one after the other and hands the boundary row of every column from band to band
// through an HBM scratch row (1 KB per column per hand-over: ~150 KB per item at the north-star shape, the largest traffic of the
// whole pass).  Here the HS_COOP_WAVES wavefronts of a workgroup take one band each and run them as a pipeline, wavefront w working
// on column t - w at step t: the boundary of a column travels through a two-slot LDS ring (16 B x 64 lanes), a workgroup barrier
// per step keeps the wavefronts one column apart, and an item takes nmax + nb - 1 steps instead of nb * nmax.  Fewer rows per
// wavefront also means the row constants fit the SGPR file and the state fits 3 wavefronts per SIMD.  Same cells, same operation
// order per cell: bit-identical to the serial sweep.  Blocks with more rows than HS_COOP_WAVES x HS_COOP_ROWS run in rounds of
// HS_COOP_WAVES bands, the boundary between rounds going through the scratch row as before.
#ifndef HS_COOP_WAVES
#define HS_COOP_WAVES 4
#endif
#ifndef HS_COOP_ROWS
#define HS_COOP_ROWS 15
#endif
#ifndef HS_COOP_OCC
#define HS_COOP_OCC 3         // wavefronts per SIMD the register allocation aims at
#endif

// LDS operations of this wavefront are complete (and its global stores, when it hands a boundary over through memory), then the
// workgroup barrier.  Unlike __syncthreads() this does not wait for the global LOADS in flight — the next column's prefetch.
__device__ __forceinline__ void coop_barrier(bool drain_stores){
  if (drain_stores) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// The ring is addressed through LDS-typed pointers: with generic ones the compiler folds "boundary from the scratch row or from the
// ring" into one flat_load of a selected address, which counts against the vector-memory counter and makes every column wait for
// the next column's prefetch.
typedef const __attribute__((address_space(3))) double* hs_lds_cd2;      // (M, D) pairs, read and written as doubles
typedef __attribute__((address_space(3))) double* hs_lds_d2;
// The transition logs of the band's rows (m2m, m2i: 4 SGPRs per row, 60 for 15 rows) do not fit the SGPR file next to everything else;
// left to the compiler they are parked in VGPR lanes and cost two v_readlane_b32 (8 VALU cycles) and a few v_mov per cell.  With
// HS_COOP_LDS_CONSTS the wavefront keeps them in an LDS table (16 B per row) and reads a row's pair back as one broadcast
// ds_read_b128, issued two rows ahead of its use: the LDS pipe is idle in this kernel, the VALU is what binds.  The loads are inline
// assembly (volatile: not hoisted back out of the column loop into registers) with their own s_waitcnt; LDS operations of a wavefront
// complete in order, so the compiler's own counts for the ring accesses only ever wait longer, never less.
#ifndef HS_COOP_LDS_CONSTS
#define HS_COOP_LDS_CONSTS 1
#endif
#ifndef HS_COOP_LDS_EMIT
#define HS_COOP_LDS_EMIT 1        // trailing flank: emissions from a per-column LDS table instead of a compare and two selects per cell
#endif
#ifndef HS_COOP_LDS_DEPTH
#define HS_COOP_LDS_DEPTH 3       // rows between a pair's request and its use (<= 6)
#endif
typedef double hs_d2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ hs_d2v kload(uint32_t addr, int off){      // off: a constant once the row loop is unrolled
  hs_d2v q;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(addr), "i"(off) : "memory");
  return q;
}
template <int CNT> __device__ __forceinline__ void kwait(hs_d2v& q){ asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(q) : "n"(CNT)); }
// Trailing flank, emission of a cell: (read base == haplotype base) ? log P(correct) : log P(error) is the same number in every lane of a
// read.  With EL the lanes of a read write the band's emissions of a column — lane s of the read that of row s (and of row s + 8 where a read
// has only 8 lanes) — to a small LDS table once per column (one column ahead, from the prefetched column values), and a cell reads its own
// with one ds_read_b64 at [column parity][the lane's read][row]: the row is the instruction's immediate offset, so a cell costs no VALU
// instruction at all (until round 6 the table was indexed by the row's base code: a v_add_u32 per cell and rows of A, C, G, T, N only;
// before that v_cmp + 2 v_cndmask per cell).  Needs >= 8 lanes per read (allele groups of 8 and more); otherwise the select stays.
#define HS_ETAB_ROWS 17          // doubles per read in the table: 16 row slots + 1 (the reads' entries of one row fall into different banks)
#define HS_ETAB_PAR (8*HS_ETAB_ROWS)     // doubles per column parity: up to 8 reads per wavefront
__device__ __forceinline__ double eload(uint32_t addr, int off){      // off: a constant once the row loop is unrolled
  double e;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(e) : "v"(addr), "i"(off) : "memory");
  return e;
}
template <int CNT> __device__ __forceinline__ void ewait(double& e, hs_d2v& q){ asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(e), "+v"(q) : "n"(CNT)); }
template <int CNT> __device__ __forceinline__ void ewait1(double& e){ asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(e) : "n"(CNT)); }
// Hand-over between the band wavefronts of a workgroup (round 6).  Until round 5 the four wavefronts met at an s_barrier after every read
// column — 78 barriers per item, each waiting for the slowest of four wavefronts that share their SIMDs with two other workgroups: a fifth
// of the wavefronts' cycles (SQ_WAIT_INST_ANY 0.26, VALU pipe 0.79 with 12 of 14.5 instructions per cell FP64).  Now a band publishes the
// number of columns it has finished in an LDS word (prog[w]) after storing the column's boundary (M, D) pairs into a ring of HS_RING
// columns, and waits only for what it needs: its upper neighbour to have finished the column it is about to start, and its lower neighbour
// to have consumed the ring slot it is about to overwrite (HS_RING columns back).  LDS operations of a wavefront execute in order and the
// LDS has no cache: a wavefront that sees the counter sees the boundary values stored before it.  A band runs ahead of the band below it
// by up to HS_RING columns; nothing else synchronises inside an item.  Same cells, same operations: bit-identical.
#ifndef HS_RING
#define HS_RING 8        // columns of boundary values a band may be ahead of the band below it (power of two)
#endif
typedef __attribute__((address_space(3))) int* hs_lds_i;
#ifdef HS_FTIME      // timing experiment: cycles a band wavefront spends waiting for its neighbours / sweeping / between items, printed by a few workgroups
__device__ unsigned long long g_ft[1024][8][6];
#define HS_FT_ADD(k, v) do { if (lane == 0) g_ft[blockIdx.x & 1023][w][k] += (v); } while (0)
#define HS_FT_NOW() __builtin_amdgcn_s_memtime()
#else
#define HS_FT_ADD(k, v) do {} while (0)
#define HS_FT_NOW() 0ull
#endif
// (the counters' addresses and values are wave-uniform: they stay in scalar registers and pass through short-lived vector registers inside
// the statement — the sweep has none to spare: 15 rows x (M, I, D) + constants fill the 168 a wavefront may have at three per SIMD)
__device__ __forceinline__ int prog_read(uint32_t addr_uniform){
  int v;
  asm volatile("v_mov_b32 %0, %1\n\tds_read_b32 %0, %0\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "s"(addr_uniform) : "memory");
  return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void prog_write(uint32_t addr_uniform, int v_uniform){        // (every lane stores the same word: no exec juggling)
  int t0, t1;
  asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\tds_write_b32 %0, %1" : "=&v"(t0), "=&v"(t1) : "s"(addr_uniform), "s"(v_uniform) : "memory");
}
template <int NR, bool FIRST, bool LAST, bool LEAD, bool EL>
__device__ __forceinline__ void band_sweep_coop(const hs_dev_t& d, int lane, bool live, int n, int nmax, const double* __restrict__ col,
                                                const hs_row_t* __restrict__ rows, int row0, int c0, const double* __restrict__ mr,
                                                double* __restrict__ bnd, bool topg, bool botg, hs_lds_cd2 lds_top, hs_lds_d2 lds_bot,
                                                double* __restrict__ lt, double* __restrict__ rowp, double* __restrict__ side_out,
                                                int w, hs_lds_i prog, int base, int nsteps, int wlast, hs_lds_d2 ktab, hs_lds_d2 etab, int npad){
  static_assert(!EL || (HS_COOP_LDS_CONSTS != 0 && !LEAD), "the emission table rides on the constants' request pipeline");
  constexpr bool KL = HS_COOP_LDS_CONSTS != 0;
  int hc[NR]; double m2m[KL ? 1 : NR], m2i[KL ? 1 : NR];
  const uint32_t kaddr = (uint32_t)(uintptr_t)ktab;           // LDS byte address of this wavefront's table
  if (KL){
    const int rr = min(lane, NR - 1);
    const int meta_l = (int)rows[row0 + rr];
    const double vm = d.m2m[(meta_l >> 8) & 15], vi = d.m2i[(meta_l >> 8) & 15];
    if (lane < NR){ ktab[2*rr] = vm; ktab[2*rr + 1] = vi; }
    wave_lds_sync();
  }
#pragma unroll
  for (int r = 0; r < NR; r++){
    const int meta = uni((int)rows[row0 + r]);
    hc[r] = meta & 0xff;                                               // (EL: not used — the table's writers compare)
    if (!KL){ m2m[KL ? 0 : r] = uni(d.m2m[(meta >> 8) & 15]); m2i[KL ? 0 : r] = uni(d.m2i[(meta >> 8) & 15]); }
  }
  double Mp[NR], Qp[NR], Ip[NR];          // per row: M[r][j], Q[r] = max(I[r][j], D[r-1][j]) and — ahead by a column — I[r][j+1]
  double nx_blc = col[0], nx_blw = col[1], nx_rd = col[2];
  double nx_mr = 0.0;
  // EL: table of two column parities x (64 / npad reads) x HS_ETAB_ROWS entries; lane `slot` of a read writes the entry of row `slot`
  // (npad == 8: and of row slot + 8), comparing the column's read base with that row's base
  static_assert(!EL || NR <= 16, "rows of a band in the emission table");
  const int e_sub = EL ? lane / npad : 0, e_slot = EL ? lane - e_sub*npad : 0;
  int e_char = 0;
  if (EL) e_char = ((int)rows[row0 + min(e_slot, NR - 1)] & 0xff) | (((int)rows[row0 + min(e_slot + 8, NR - 1)] & 0xff) << 8);
  const int e_wr = e_sub*HS_ETAB_ROWS + e_slot;
  const uint32_t e_rd = EL ? (uint32_t)(uintptr_t)etab + (uint32_t)(8*HS_ETAB_ROWS)*(uint32_t)e_sub : 0;      // + parity * 8 HS_ETAB_PAR + 8 row
  auto e_write = [&](int par, double rdv, double blc, double blw){
    const int rdb = (int)rdv;
    if (e_slot < NR) etab[par*HS_ETAB_PAR + e_wr] = (rdb == (e_char & 0xff)) ? blc : blw;
    if (NR > 8 && npad == 8 && e_slot + 8 < NR) etab[par*HS_ETAB_PAR + e_wr + 8] = (rdb == (e_char >> 8)) ? blc : blw;
  };
  if (EL){ e_write(0, nx_rd, nx_blc, nx_blw); wave_lds_sync(); }
  double diagM = 0;
  double pre = 0.0;                              // LEAD: left_prob, a strictly sequential sum in the reference
  const uint32_t a_me = (uint32_t)uni((int)(uintptr_t)(prog + w)), a_top = (uint32_t)uni((int)(uintptr_t)(prog + (w > 0 ? w - 1 : 0))), a_bot = (uint32_t)uni((int)(uintptr_t)(prog + w + 1));
  // base: the columns this workgroup's bands had finished before this sweep — the counters are never reset where sweeps follow each other
  // without a barrier (hs_trail_kernel_coop); a column's number, its ring slot and the values waited for are base + j
  int top_seen = 0, bot_seen = 0;                // columns the upper / the lower neighbour is known to have finished
  const unsigned long long t_sweep0_ = HS_FT_NOW();
  // BAR (the leading flanks: few items, a short chain of them per workgroup — what counts there is the latency of a hand-over, and a
  // barrier wakes its waiters faster than a polled counter: 3.8 against 4.8 ms per NS pass): the bands meet at a barrier after every
  // column, band w one column behind band w - 1, as until round 5; the counters are not used.
  constexpr bool BAR = LEAD;
#pragma unroll 1       // (left to itself the compiler peels and unrolls the column loop: 2000 scratch operations in a kernel that has 168 registers and needs them all)
  for (int t = 0; t < (BAR ? nsteps : nmax); t++){
    const int j = BAR ? t - w : t;
    if (!BAR || (j >= 0 && j < nmax)){
      const double blcj = nx_blc, blwj = nx_blw; const int rdj = (int)nx_rd;
      const double cur_mr = nx_mr;
      double2 cur_b = make_double2(0.0, 0.0);
      if (!FIRST){
        if (topg){
          // (streamed rounds: the previous round's last band — wavefront wlast — stored this column's boundary in the workgroup's HBM scratch and
          //  published base - nmax + j + 1 behind it; no barrier between the rounds)
          if (!BAR && wlast >= 0 && top_seen <= base - nmax + j)
            while ((top_seen = prog_read((uint32_t)uni((int)(uintptr_t)(prog + wlast)))) <= base - nmax + j) __builtin_amdgcn_s_sleep(1);
          cur_b = *(const double2*)(bnd + ((size_t)j*64 + lane)*2);
        } else {
          if (!BAR && top_seen <= base + j){
            const unsigned long long t0_ = HS_FT_NOW(); int spins_ = 0;
            while ((top_seen = prog_read(a_top)) <= base + j){ __builtin_amdgcn_s_sleep(1); spins_++; }
            HS_FT_ADD(0, HS_FT_NOW() - t0_); if (spins_) HS_FT_ADD(4, 1);
          }
          const int e = 2*(((base + j) & (HS_RING - 1))*64 + lane); cur_b = make_double2(lds_top[e], lds_top[e + 1]);
        }
      }
      {
        const int jn = min(j + 1, n - 1);           // a lane past its own read end keeps re-reading its last column
        nx_blc = col[3*jn]; nx_blw = col[3*jn+1]; nx_rd = col[3*jn+2];
        if (FIRST && !LEAD) nx_mr = mr[jn - 1 >= 0 ? jn - 1 : 0];
      }
      double upM, upD;
      if (FIRST){
        const double e0 = (rdj == c0) ? blcj : blwj;
        if (LEAD){
          upM = e0 + pre;
          pre += blcj;
          if (j == n-1 && live) *side_out = pre;                     // side_prob: the whole side hangs off the haplotype
        } else upM = (j == 0) ? e0 : e0 + cur_mr;
        upD = IMP;
        if (j == n-1 && live) lt[0] = upM;
      } else { upM = cur_b.x; upD = cur_b.y; }
      const double topM = upM;
      // Round 6: 11 instead of 12 FP64 operations per cell.  I[r][j+1] = blc[j+1] + max(M[r-1][j] + T_I2M, I[r][j] + T_I2I) and
      // D[r][j] = max(M[r-1][j] + T_D2M, D[r-1][j] + T_D2D) start from the SAME sum — T_I2M and T_D2M are one constant
      // (AlignmentModel.h:7-10) — which the sweep used to form twice, once per column.  So the D pass of column j also forms row r's I of
      // column j + 1 (the next column's blc is already here: it is prefetched at the top of the column) and, before I[r][j] is
      // overwritten, Q[r] = max(I[r][j], D[r-1][j]) — what M[r][j+1] takes its second addend from.  D itself is not kept across columns any
      // more (only Q needs it): the state stays three values per row.  Same operations on the same operands, one of them shared:
      // bit-identical.
      auto dstep = [&](int r, double Icur){      // row r of the top-down pass; Icur = I[r][j]; upM / upD = M / D of the row above in this column
        const double X = upM + T_D2M;
        const double nD = fmax(X, upD + T_D2D);
        Qp[r] = fmax(Icur, upD);
        Ip[r] = nx_blc + fmax(X, Icur + T_I2I);
        upM = Mp[r]; upD = nD;
      };
      if (j == 0){
#pragma unroll
        for (int r = 0; r < NR; r++){           // first read column (HapAligner.cpp:123-126): M = e, I = blc, D from the row above
          double e;
          if (EL){ e = eload(e_rd, 8*r); ewait1<0>(e); }      // column 0: parity 0
          else e = (rdj == hc[r]) ? blcj : blwj;
          Mp[r] = e;
          dstep(r, blcj);
        }
      } else {
        // M bottom-up in place (row r takes M[r-1] of the previous column), then D, Q and the next column's I top-down through the new M
        if (KL){
          // rows NR-1 .. 0; the pair of row r (and, EL, its emission) is requested while row r + KD is computed
          constexpr int KD = HS_COOP_LDS_DEPTH, KM = KD + 1, PER = EL ? 2 : 1;       // PER: requests per row
          hs_d2v kq[KM]; double eq[EL ? KM : 1];
          const uint32_t e_col = EL ? e_rd + (uint32_t)(8*HS_ETAB_PAR)*(uint32_t)(j & 1) : 0;
#pragma unroll
          for (int a = 0; a < KD; a++) if (NR - 1 - a >= 0){
            const int r = NR - 1 - a >= 0 ? NR - 1 - a : 0;
            kq[r % KM] = kload(kaddr, 16*r);
            if (EL) eq[EL ? r % KM : 0] = eload(e_col, 8*r);
          }
#pragma unroll
          for (int r = NR - 1; r >= 0; r--){
            if (r >= KD){
              kq[(r - KD) % KM] = kload(kaddr, 16*(r - KD));
              if (EL) eq[EL ? (r - KD) % KM : 0] = eload(e_col, 8*(r - KD));
            }
            const int young = (r >= KD ? KD : r) * PER;             // requests issued after this row's
            if (EL) switch (young){
              case 0: ewait<0>(eq[EL ? r % KM : 0], kq[r % KM]); break; case 2: ewait<2>(eq[EL ? r % KM : 0], kq[r % KM]); break;
              case 4: ewait<4>(eq[EL ? r % KM : 0], kq[r % KM]); break; case 6: ewait<6>(eq[EL ? r % KM : 0], kq[r % KM]); break;
              case 8: ewait<8>(eq[EL ? r % KM : 0], kq[r % KM]); break; case 10: ewait<10>(eq[EL ? r % KM : 0], kq[r % KM]); break;
              default: ewait<12>(eq[EL ? r % KM : 0], kq[r % KM]); break;
            } else switch (young){
              case 0: kwait<0>(kq[r % KM]); break; case 1: kwait<1>(kq[r % KM]); break; case 2: kwait<2>(kq[r % KM]); break;
              case 3: kwait<3>(kq[r % KM]); break; case 4: kwait<4>(kq[r % KM]); break; case 5: kwait<5>(kq[r % KM]); break;
              default: kwait<6>(kq[r % KM]); break;
            }
            const double e = EL ? eq[EL ? r % KM : 0] : ((rdj == hc[r]) ? blcj : blwj);
            const double dM = (r == 0) ? diagM : Mp[r > 0 ? r-1 : 0];
            Mp[r] = e + fmax(dM + kq[r % KM].x, Qp[r] + kq[r % KM].y);
          }
        } else {
#pragma unroll
        for (int r = NR - 1; r >= 0; r--){
          const double e = (rdj == hc[r]) ? blcj : blwj;
          const double dM = (r == 0) ? diagM : Mp[r > 0 ? r-1 : 0];
          Mp[r] = e + fmax(dM + m2m[KL ? 0 : r], Qp[r] + m2i[KL ? 0 : r]);
        }
        }
#pragma unroll
        for (int r = 0; r < NR; r++) dstep(r, Ip[r]);
      }
      if (!LAST){
        if (botg) *(double2*)(bnd + ((size_t)j*64 + lane)*2) = make_double2(upM, upD);
        else {
          if (!BAR && base + j - bot_seen >= HS_RING){
            const unsigned long long t0_ = HS_FT_NOW(); int spins_ = 0;
            while (base + j - (bot_seen = prog_read(a_bot)) >= HS_RING){ __builtin_amdgcn_s_sleep(1); spins_++; }
            HS_FT_ADD(1, HS_FT_NOW() - t0_); if (spins_) HS_FT_ADD(5, 1);
          }
          const int e = 2*(((base + j) & (HS_RING - 1))*64 + lane); lds_bot[e] = upM; lds_bot[e + 1] = upD;
        }
      } else if (LEAD){ if (j < n && live) rowp[j] = upM; }
      // this column is done: its boundary (if a band below reads it from the ring) is stored, and the ring slot the band above filled for it
      // has been read — both in front of the counter's store, and a wavefront's LDS operations execute in order
      if (!BAR && !(FIRST && LAST)){
        if (botg) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the boundary stored in HBM has arrived before the next round's first band is told
        else asm volatile("" ::: "memory");
        prog_write(a_me, base + j + 1);
      }
      if (EL) e_write((j + 1) & 1, nx_rd, nx_blc, nx_blw);       // the next column's emissions (its values were requested at the top of this one)
      diagM = topM;                              // top boundary of this column = diagonal of the band's first row next column
      if (j == n-1 && live){
#pragma unroll
        for (int r = 0; r < NR; r++) lt[row0 + r] = Mp[r];           // last read column of this lane's read
      }
    }
    if (BAR) coop_barrier(botg);
  }
  HS_FT_ADD(2, HS_FT_NOW() - t_sweep0_); HS_FT_ADD(3, (unsigned long long)nmax);
}

template <int NR, bool LEAD, bool EL>
__device__ __forceinline__ void band_dispatch_coop(bool first, bool last, const hs_dev_t& d, int lane, bool live, int n, int nmax, const double* col,
                                                   const hs_row_t* rows, int row0, int c0, const double* mr, double* bnd, bool topg, bool botg,
                                                   hs_lds_cd2 lds_top, hs_lds_d2 lds_bot, double* lt, double* rowp, double* side_out, int w, hs_lds_i prog, int base, int nsteps, int wlast, hs_lds_d2 ktab,
                                                   hs_lds_d2 etab, int npad){
  if (first){ if (last) band_sweep_coop<NR, true, true, LEAD, EL>(d, lane, live, n, nmax, col, rows, row0, c0, mr, bnd, topg, botg, lds_top, lds_bot, lt, rowp, side_out, w, prog, base, nsteps, wlast, ktab, etab, npad);
              else      band_sweep_coop<NR, true, false, LEAD, EL>(d, lane, live, n, nmax, col, rows, row0, c0, mr, bnd, topg, botg, lds_top, lds_bot, lt, rowp, side_out, w, prog, base, nsteps, wlast, ktab, etab, npad); }
  else      { if (last) band_sweep_coop<NR, false, true, LEAD, EL>(d, lane, live, n, nmax, col, rows, row0, c0, mr, bnd, topg, botg, lds_top, lds_bot, lt, rowp, side_out, w, prog, base, nsteps, wlast, ktab, etab, npad);
              else      band_sweep_coop<NR, false, false, LEAD, EL>(d, lane, live, n, nmax, col, rows, row0, c0, mr, bnd, topg, botg, lds_top, lds_bot, lt, rowp, side_out, w, prog, base, nsteps, wlast, ktab, etab, npad); }
}

// The rounds of one item: `n_rows` haplotype rows (after the block's first row) cut into bands, W bands per round, one per wavefront.
// prog: the bands' column counters (above).  Two regimes:
//   * gcol == NULL (leading flanks): the counters are zero when the item starts — the caller clears them between the two barriers of its
//     item fetch — and are cleared again between the rounds of a flank deeper than one round holds;
//   * gcol != NULL (trailing flanks): the counters count the columns of every sweep the workgroup has run (*gcol, the same number in every
//     wavefront); nothing is cleared, a wavefront sets its counter to the new total after every round whether it had a band or not, and
//     neither consecutive items nor the rounds of a deep flank have a barrier between them: the last band of a round stores its boundary in
//     the workgroup's HBM scratch, drains the store and only then publishes the column; the next round's first band waits for that counter
//     column by column (band_sweep_coop, topg / wlast).
// (Leading flanks between rounds: stores drained, then the barrier, as before.)
template <int R, int W, bool LEAD, bool EL = false>
__device__ __forceinline__ void coop_rounds(const hs_dev_t& d, int w, int lane, bool live, int n, int nmax, const double* col, const hs_row_t* rows, int n_rows, int c0,
                                            const double* mr, double* bnd, double2 (*ring)[HS_RING*64], int* prog_s, int* gcol, double* lt, double* rowp, double* side_out, double2 (*ktabs)[24], double (*etabs)[2*HS_ETAB_PAR] = NULL, int npad = 64){
  // as many bands as there are wavefronts whenever the rows allow it (all wavefronts busy), more rounds only for blocks deeper than one round holds
  const int rounds = (n_rows + R*W - 1) / (R*W);
  const int nbands = min(n_rows, rounds*W);
  const int nr_base = n_rows / nbands, nr_rem = n_rows - nr_base*nbands;
  hs_lds_i prog = (hs_lds_i)prog_s;
  for (int g = 0; g < rounds; g++){
    if (g > 0 && !gcol){
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the boundary the previous round's last band stored for this round's first
      __syncthreads();
      if (lane == 0) prog_s[w] = 0;
      __syncthreads();
    }
    // (streamed: no barrier between rounds either — the next round's first band waits for the previous round's last band column by column)
    const int wlast = (gcol && g > 0) ? W - 1 : -1;
    const int base = gcol ? *gcol : 0;
    const int nb_round = min(W, nbands - g*W);
    const int nsteps = nmax + nb_round - 1;              // (LEAD: the round's barrier steps, the same count in every wavefront)
    const int b = g*W + w;
    if (w < nb_round){
      const int nr = nr_base + (b < nr_rem ? 1 : 0);
      const int row0 = 1 + b*nr_base + min(b, nr_rem);
      const bool first = (b == 0), last = (b + 1 == nbands);
      const bool topg = (w == 0) && (g > 0), botg = (w + 1 == nb_round) && !last;
      hs_lds_cd2 lds_top = (hs_lds_cd2)ring[w > 0 ? w - 1 : 0]; hs_lds_d2 lds_bot = (hs_lds_d2)ring[w];
      hs_lds_d2 ktab = (hs_lds_d2)ktabs[w];
      hs_lds_d2 etab = EL ? (hs_lds_d2)etabs[w] : (hs_lds_d2)ktabs[w];
      switch (nr){
#define HS_COOP_CASE(N_) case N_: if (N_ <= R) band_dispatch_coop<(N_ <= R ? N_ : 1), LEAD, EL>(first, last, d, lane, live, n, nmax, col, rows, row0, c0, mr, bnd, topg, botg, lds_top, lds_bot, lt, rowp, side_out, w, prog, base, nsteps, wlast, ktab, etab, npad); break;
        HS_COOP_CASE(1) HS_COOP_CASE(2) HS_COOP_CASE(3) HS_COOP_CASE(4) HS_COOP_CASE(5) HS_COOP_CASE(6) HS_COOP_CASE(7) HS_COOP_CASE(8)
        HS_COOP_CASE(9) HS_COOP_CASE(10) HS_COOP_CASE(11) HS_COOP_CASE(12) HS_COOP_CASE(13) HS_COOP_CASE(14) HS_COOP_CASE(15) HS_COOP_CASE(16)
        HS_COOP_CASE(17) HS_COOP_CASE(18) HS_COOP_CASE(19) HS_COOP_CASE(20)
#undef HS_COOP_CASE
        default: if (LEAD) for (int t = 0; t < nsteps; t++) coop_barrier(false); break;
      }
    } else if (LEAD) for (int t = 0; t < nsteps; t++) coop_barrier(false);       // a wavefront without a band in this round keeps the step count
    if (gcol){
      *gcol = base + nmax;
      asm volatile("" ::: "memory");
      prog_write((uint32_t)uni((int)(uintptr_t)(prog + w)), base + nmax);      // (also by a wavefront without a band: its neighbours' later waits count from here)
    }
  }
}

// Trailing flanks, round 6: the workgroup STREAMS through its items.  Until round 5 an item was fetched behind a workgroup barrier, every
// wavefront walked the item's chain of dependent loads (item -> group -> read -> locus -> workspace -> allele -> rowset: ~9 columns' worth
// of latency per 75-column item) and the four-band pipeline filled and drained around each item (tools: HS_FTIME build, profiles/r06_notes.md:
// 14 % of the first band's cycles outside its sweeps, 13 % of the last band's waiting for the band above).  Now
//   * the LAST wavefront of the workgroup — the one that waits for the pipeline to fill anyway — takes the next item from the chunk's
//     counter and walks its loads while the bands above it are already sweeping the current one; what an item's sweeps need goes into one of
//     two LDS records (uniform words + per-lane n and the three workspace offsets), published by a counter (s_q[0]);
//   * every wavefront reads the record of item k when it gets there, sweeps, and goes on to item k + 1 without a barrier: the bands' column
//     counters count through the items (coop_rounds, gcol), so the first band starts the next item while the last is still three columns
//     behind in this one — the pipeline never drains.
// The only barriers left are those between the rounds of a flank deeper than one round (HBM hand-over), passed by every wavefront alike.
// Same cells, same operations, same order per cell: bit-identical.
struct TrailRec {          // what the sweeps of one item need; uniform words first
  int item_ok;             // 0: past the last item (the stream ends)
  int n_rows;              // rows after the block's first row (0: the single "must be followed by a match" row, HapAligner.cpp:130-139)
  int nmax, npad, nreads, nm, el_ok, c0, rs_off, pad;
  int n[64];
  long long col[64], mr[64], lt[64];
};
template <int R, int W, int OCC>
__global__ void __launch_bounds__(64*W, OCC) hs_trail_kernel_coop(const hs_dev_t* __restrict__ dp, int item_begin, int item_end, int chunk){
  const hs_dev_t& d = *dp;
  const int lane = threadIdx.x & 63, w = uni((int)(threadIdx.x >> 6));
  __shared__ double2 ring[W][HS_RING*64];
  __shared__ int s_prog[W + 1];             // columns finished per band wavefront (band_sweep_coop); [W]: never written, the last band's lower neighbour
  __shared__ int s_q[2 + W];                // [0] records published; [2 + w] items whose record wavefront w has read
  __shared__ double2 ktabs[W][24];          // per wavefront: (m2m, m2i) of its band's rows (HS_COOP_LDS_CONSTS)
  __shared__ double etabs[W][2*HS_ETAB_PAR];  // per wavefront: emissions of the current and the next column per (read, row of the band)
  __shared__ TrailRec s_rec[2];
  double* const bnd = d.ws_band + (size_t)blockIdx.x * d.band_cols * 64 * 2;
  int32_t* const ctr = d.redo + d.n_active + chunk;
#ifdef HS_FTIME
  if (lane == 0) for (int k = 0; k < 6; k++) g_ft[blockIdx.x & 1023][w][k] = 0;
  const unsigned long long t_k0_ = HS_FT_NOW(); int n_items_ = 0;
#endif
  const uint32_t a_q = (uint32_t)uni((int)(uintptr_t)(__attribute__((address_space(3))) int*)&s_q[0]);
  // The record of the k-th item of this workgroup, written by the last wavefront (all 64 lanes: lane = the sweep's lane)
  auto prefetch = [&](int k){
    TrailRec& rc = s_rec[k & 1];
    int it_i = 0;
    if (lane == 0) it_i = atomicAdd(ctr, 1);
    const int item = item_begin + uni(it_i);
    if (item >= item_end){ if (lane == 0) rc.item_ok = 0; }
    else {
      const hs_item_t* it = d.items + item;
      const int side = uni(it->side), nreads = uni(it->rowset);
      const hs_tgroup_t* g = d.tgroups + uni(it->slot);
      const int nm = uni(g->n_members);
      int npad = 1; while (npad < nm) npad <<= 1;
      const int sub = lane / npad, slot = lane - sub*npad;
      const int ai = d.tpack[uni(it->active) + min(sub, nreads-1)];
      const int r = d.active[ai];
      const hs_read_t rdv = d.reads[r];
      const hs_locus_t* loc = d.loci + uni(rdv.locus);
      const int nL = rdv.seed, n = side ? rdv.len - rdv.seed - 1 : rdv.seed;
      const hs_ws_t wsr = d.ws[ai];
      const int k_al = d.tmembers[uni(g->member_off) + min(slot, nm-1)];
      const hs_allele_t* al = d.alleles + uni(loc->hap_begin) + k_al;
      const int ord = al->re_ord;
      const int rowset = uni(g->rowset);
      const int rs_off = uni(d.rowsets[rowset].off), rs_len = uni(d.rowsets[rowset].len);
      const hs_row_t* rows = d.rows + rs_off;
      // emissions through the LDS table: 8 lanes per read to write a column's entries (two rows each, then)
      const bool el_ok = (HS_COOP_LDS_EMIT != 0) && (npad >= 8);
      const int nmax = uni(wave_max_i(n));         // (a cross-lane maximum: every lane takes part)
      const int c0 = uni((int)rows[0]) & 0xff;
      rc.n[lane] = n;
      rc.col[lane] = wsr.col + 3*(int64_t)(side ? nL : 0);
      rc.mr[lane] = wsr.mr + (int64_t)ord*(rdv.len-1) + (side ? nL : 0);
      rc.lt[lane] = wsr.lt + (int64_t)ord*uni(loc->lt_stride) + (side ? d.rowsets[al->trail_rows[0]].len : 0);
      if (lane == 0){
        rc.item_ok = 1; rc.n_rows = rs_len - 1; rc.nmax = nmax; rc.npad = npad; rc.nreads = nreads; rc.nm = nm; rc.el_ok = el_ok ? 1 : 0;
        rc.c0 = c0; rc.rs_off = rs_off;
      }
    }
    asm volatile("" ::: "memory");           // the record's stores are issued before the counter's (a wavefront's LDS operations execute in order)
    prog_write(a_q, k + 1);
  };
  if (threadIdx.x < W + 1) s_prog[threadIdx.x] = 0;
  if (threadIdx.x < 2 + W) s_q[threadIdx.x] = 0;
  __syncthreads();
  int gcol = 0;                              // columns this workgroup's sweeps have covered so far: the same in every wavefront
  for (int k = 0;; k++){
    if (w == W - 1){
      if (k == 0) prefetch(0);
      // item k + 1 goes into the record item k - 1 had: every wavefront must have read that one (it has, long since, unless items without sweeps let this one run ahead)
      for (int ww = 0; ww < W - 1; ww++)
        while (prog_read(a_q + 4u*(uint32_t)(2 + ww)) < k) __builtin_amdgcn_s_sleep(1);
    } else {
      while (prog_read(a_q) <= k) __builtin_amdgcn_s_sleep(1);
    }
    const TrailRec& rc = s_rec[k & 1];
    if (uni(rc.item_ok) == 0) break;
    const int n_rows = uni(rc.n_rows), nmax = uni(rc.nmax), npad = uni(rc.npad), nreads = uni(rc.nreads), nm = uni(rc.nm), c0 = uni(rc.c0);
    const bool el_ok = uni(rc.el_ok) != 0;
    const hs_row_t* rows = d.rows + uni(rc.rs_off);
    const int n = rc.n[lane];
    const double* col = d.ws_col + rc.col[lane];
    const double* mr = d.ws_mr + rc.mr[lane];
    double* lt = d.ws_lt + rc.lt[lane];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    prog_write(a_q + 4u*(uint32_t)(2 + w), k + 1);          // this wavefront has read the record of item k
    if (w == W - 1) prefetch(k + 1);                            // ... and the last one walks the next item's loads while the bands above it sweep this one
#ifdef HS_FTIME
    n_items_++;
#endif
    const int sub = lane / npad, slot = lane - sub*npad;
    const bool live = (sub < nreads) && (slot < nm);
    if (n_rows == 0){     // the block is the single "must be followed by a match" row (HapAligner.cpp:130-139)
      if (w == 0){
        const int j = n - 1;
        const double blcj = col[3*j], blwj = col[3*j+1]; const int rdj = (int)col[3*j+2];
        const double e0 = (rdj == c0) ? blcj : blwj;
        if (live) lt[0] = (j == 0) ? e0 : e0 + mr[max(j-1, 0)];
      }
      continue;
    }
    if (el_ok) coop_rounds<R, W, false, true>(d, w, lane, live, n, nmax, col, rows, n_rows, c0, mr, bnd, ring, s_prog, &gcol, lt, NULL, NULL, ktabs, etabs, npad);
    else
    coop_rounds<R, W, false>(d, w, lane, live, n, nmax, col, rows, n_rows, c0, mr, bnd, ring, s_prog, &gcol, lt, NULL, NULL, ktabs);
  }
#ifdef HS_FTIME
  if ((blockIdx.x % 191) == 7 && lane == 0){
    const unsigned long long* f = g_ft[blockIdx.x & 1023][w];
    printf("trail wg %d wave %d: kernel %llu cycles, %d items, sweeps %llu (columns %llu), wait top %llu (%llu spins>0), wait bottom %llu (%llu)\n", (int)blockIdx.x, w,
           HS_FT_NOW() - t_k0_, n_items_, f[2], f[3], f[0], f[4], f[1], f[5]);
  }
#endif
}

template <int R, int W, int OCC>
__global__ void __launch_bounds__(64*W, OCC) hs_lead_kernel_coop(const hs_dev_t* __restrict__ dp, int item_begin, int item_end, int chunk){
  const hs_dev_t& d = *dp;
  const int lane = threadIdx.x & 63, w = uni((int)(threadIdx.x >> 6));
  __shared__ double2 ring[W][HS_RING*64];
  __shared__ int s_prog[W + 1];                 // columns finished per band wavefront (band_sweep_coop); [W]: never written, the last band's lower neighbour
  __shared__ double2 ktabs[W][24];          // per wavefront: (m2m, m2i) of its band's rows (HS_COOP_LDS_CONSTS)
  __shared__ int s_item;
  double* const bnd = d.ws_band + (size_t)blockIdx.x * d.band_cols * 64 * 2;
  int32_t* const ctr = d.redo + d.n_active + chunk;
  for (;;){
    if (threadIdx.x == 0) s_item = atomicAdd(ctr, 1);
    __syncthreads();
    const int item = item_begin + uni(s_item);
    if (threadIdx.x <= W) s_prog[threadIdx.x] = 0;           // every wavefront is past the previous item's sweeps (the barrier above)
    __syncthreads();
    if (item >= item_end) break;
    const hs_item_t* it = d.items + item;
    const int side = uni(it->side) & 1, slot = uni(it->side) >> 1, nreads = uni(it->slot);
    const bool live = lane < nreads;
    const int ai = d.tpack[uni(it->active) + min(lane, nreads-1)];
    const hs_read_t rdv = d.reads[d.active[ai]];
    const hs_locus_t* loc = d.loci + uni(rdv.locus);
    const int nL = rdv.seed, n = side ? rdv.len - rdv.seed - 1 : rdv.seed;
    const int nmax = uni(wave_max_i(n));
    const hs_ws_t wsr = d.ws[ai];
    const int lead_flank = uni(loc->lead_flank[side]);
    double* rec = d.ws_lead + wsr.lead[side] + (int64_t)slot*(n + lead_flank + 1);     // lead_record() of this lane's read
    double* lastcol = rec + n;
    double* side_out = rec + n + lead_flank;
    const double* col = d.ws_col + wsr.col + 3*(int64_t)(side ? nL : 0);
    const int rowset = uni(it->rowset);
    const int rs_off = uni(d.rowsets[rowset].off), rs_len = uni(d.rowsets[rowset].len);
    const hs_row_t* rows = d.rows + rs_off;
    const int c0 = uni((int)rows[0]) & 0xff;
    if (rs_len - 1 == 0){     // a one-base flank: matrix row 0 is all there is
      if (w == 0){
        double pre = 0.0;
        for (int j = 0; j < nmax; j++){
          const int jc = min(j, n - 1);
          const double blcj = col[3*jc], blwj = col[3*jc+1]; const int rdj = (int)col[3*jc+2];
          const double m0 = ((rdj == c0) ? blcj : blwj) + pre;
          pre += blcj;
          if (j < n && live) rec[j] = m0;
          if (j == n-1 && live){ lastcol[0] = m0; *side_out = pre; }
        }
      }
      continue;
    }
    coop_rounds<R, W, true>(d, w, lane, live, n, nmax, col, rows, rs_len - 1, c0, NULL, bnd, ring, s_prog, NULL, lastcol, rec, side_out, ktabs);
  }
}

// ------------------------------------------------------------------ flank blocks, systolic form: one wavefront per alignment, rows as lanes
// The sweeps above put reads (leading flank) or alleles (trailing flank) on the lanes and every row of a flank on ONE workgroup: a
// column costs what 60 rows cost on one CU (~0.85 us) whatever the bands, and a one-locus call — two items, 255 idle CUs — waits
// ~120 columns x 0.85 us for each of its two sweeps (two thirds of its 0.3 ms: profiles/r03_notes.md, r04_notes.md).  For launches of a
// few items the matrix of every (read side[, allele]) goes to a wavefront of its own instead: lane = haplotype row, the wavefront moves
// along the anti-diagonals (lane l works on column t - l at step t), a cell takes M and D of the row above from the lane below it with
// one wave_shr each and keeps the diagonal from the step before.  No barrier, no ring, no sharing — 64 times the work per alignment
// of the leading flank (it is computed per read here too, but with one row per lane) and a pipeline fill of 64 steps, which is why this
// is the latency shape only — and n + rows steps of one short dependent chain each.  The row constants (base, transition logs) are
// per-lane registers, the per-column operands come from an LDS copy of the read side.  Same cells, same operations per cell as
// band_sweep: bit-identical.  Flanks of more than 64 rows run in bands of 64 rows, the boundary row kept in LDS.
#define HS_SYS_MAXCOLS 256
#define HS_SYS_ITEMS 96u
template <bool LEAD>
__global__ void __launch_bounds__(64) hs_flank_systolic(const hs_dev_t* __restrict__ dp, int item_begin){
  const hs_dev_t& d = *dp;
  const int lane = threadIdx.x, pair = blockIdx.y;
  const hs_item_t* it = d.items + item_begin + blockIdx.x;
  int side, ai, rowset;
  const hs_tgroup_t* g = NULL; int tslot = 0, lslot = 0;
  if (LEAD){
    side = uni(it->side) & 1; lslot = uni(it->side) >> 1;
    if (pair >= uni(it->slot)) return;
    ai = uni(d.tpack[uni(it->active) + pair]); rowset = uni(it->rowset);
  } else {
    side = uni(it->side);
    g = d.tgroups + uni(it->slot);
    const int nm = uni(g->n_members), sub = pair / nm;
    tslot = pair - sub*nm;
    if (sub >= uni(it->rowset)) return;
    ai = uni(d.tpack[uni(it->active) + sub]); rowset = uni(g->rowset);
  }
  const hs_read_t rdv = d.reads[uni(d.active[ai])];
  const hs_locus_t* loc = d.loci + uni(rdv.locus);
  const int nL = uni(rdv.seed), len = uni(rdv.len), n = side ? len - nL - 1 : nL;
  if (n <= 0) return;
  const hs_ws_t wsr = d.ws[ai];
  const double* col = d.ws_col + uni(wsr.col) + 3*(int64_t)(side ? nL : 0);
  const int rs_off = uni(d.rowsets[rowset].off), rs_len = uni(d.rowsets[rowset].len);
  const hs_row_t* rows = d.rows + rs_off;
  const int c0 = uni((int)rows[0]) & 0xff;
  const double* mr = NULL; double* lt; double* rowp = NULL; double* side_out = NULL;
  if (LEAD){
    const int lead_flank = uni(loc->lead_flank[side]);
    double* rec = d.ws_lead + uni(side ? wsr.lead[1] : wsr.lead[0]) + (int64_t)lslot*(n + lead_flank + 1);
    rowp = rec; lt = rec + n; side_out = rec + n + lead_flank;
  } else {
    const int k = uni(d.tmembers[uni(g->member_off) + tslot]);
    const hs_allele_t* al = d.alleles + uni(loc->hap_begin) + k;
    const int ord = uni(al->re_ord);
    mr = d.ws_mr + uni(wsr.mr) + (int64_t)ord*(len - 1) + (side ? nL : 0);
    lt = d.ws_lt + uni(wsr.lt) + (int64_t)ord*uni(loc->lt_stride) + (side ? uni(d.rowsets[uni(al->trail_rows[0])].len) : 0);
  }
  __shared__ double2 s_bq[HS_SYS_MAXCOLS];               // (log P(correct), log P(error)) per read column
  __shared__ double2 s_tb[2][HS_SYS_MAXCOLS];            // (M, D) of the row above the current band per column | the current band's last row (they swap)
  __shared__ int s_rd[HS_SYS_MAXCOLS];
  // ---- the read side and the block's first row (HapAligner.cpp:33-42 / :130-139): M of row 0 at every column
  for (int j = lane; j < n; j += 64){
    const double blc = col[3*j], blw = col[3*j + 1]; const int rdj = (int)col[3*j + 2];
    s_bq[j] = make_double2(blc, blw); s_rd[j] = rdj;
    const double e0 = (rdj == c0) ? blc : blw;
    if (!LEAD) s_tb[0][j] = make_double2((j == 0) ? e0 : e0 + mr[j - 1], IMP);
  }
  wave_lds_sync();
  if (LEAD){
    if (lane == 0){                    // left_prob is a strictly sequential sum in the reference: one lane, n additions
      double pre = 0.0;
      for (int j = 0; j < n; j++){
        const double2 bq = s_bq[j];
        s_tb[0][j] = make_double2(((s_rd[j] == c0) ? bq.x : bq.y) + pre, IMP);
        pre += bq.x;
      }
      *side_out = pre;
    }
    wave_lds_sync();
  }
  const int n_rows = rs_len - 1;
  if (lane == 0) lt[0] = s_tb[0][n - 1].x;
  if (n_rows == 0){                    // the block is its first row only
    if (LEAD) for (int j = lane; j < n; j += 64) rowp[j] = s_tb[0][j].x;
    return;
  }
  int par = 0;
  for (int b0 = 0; b0 < n_rows; b0 += 64, par ^= 1){
    const int nrb = min(64, n_rows - b0);
    const bool last_band = b0 + 64 >= n_rows;
    const int myrow = 1 + b0 + min(lane, nrb - 1);
    const bool arow = lane < nrb, last_lane = lane == nrb - 1;
    const int meta = (int)rows[myrow];
    const int hc = meta & 0xff;
    const double m2m = d.m2m[(meta >> 8) & 15], m2i = d.m2i[(meta >> 8) & 15];
    const double2* const s_top = s_tb[par];
    double2* const s_out = s_tb[par ^ 1];
    double curM = 0.0, curD = 0.0, curI = 0.0, diagM = 0.0, diagD = 0.0, ltv = 0.0, xdiag = 0.0 + T_I2M;      // xdiag = diagM + T_I2M, carried from the step before
    const int nsteps = n + nrb - 1;
    // A lane outside its read's columns (before its first, behind its last) or beyond the band's rows computes on clamped operands and
    // nothing of it reaches a cell that counts: column 0 takes nothing from the lane's own past, a cell's neighbours are a column
    // behind it in the row above.  So the state is updated unconditionally — no mask, no branch — and only the results are picked:
    // the last read column's M (ltv) and the band's last row (s_out).  The operands of step t + 1 are requested while step t is
    // computed (a step is one short dependent chain: an LDS round trip in front of it would be most of its time).
    int jn = -lane;                                       // the lane's column at the step being prefetched
    auto clampj = [&](int j){ return min(max(j, 0), n - 1); };
    double2 nx_bq = s_bq[clampj(jn)], nx_top = s_top[0]; int nx_rd = s_rd[clampj(jn)];
    for (int t = 0; t < nsteps; t++){
      const double2 bq = nx_bq, top = nx_top; const int rdj = nx_rd;
      const int j = jn;
      jn++;
      { const int jq = clampj(jn); nx_bq = s_bq[jq]; nx_rd = s_rd[jq]; nx_top = s_top[min(t + 1, n - 1)]; }
      const double upM = shr1(top.x, curM), upD = shr1(top.y, curD);
      const double e = (rdj == hc) ? bq.x : bq.y;
      // (m2d == m2i: max(a + x, b + x) == max(a, b) + x exactly, as in band_sweep); column 0: HapAligner.cpp:123-126
      const bool first = (j == 0);
      const double nM = first ? e : e + fmax(diagM + m2m, fmax(curI, diagD) + m2i);
      // (round 6, as in band_sweep_coop: M of the row above + T_I2M of this column's I is the sum the previous step's D formed — T_I2M and T_D2M are one constant)
      const double xup = upM + T_D2M;
      const double nI = first ? bq.x : bq.x + fmax(xdiag, curI + T_I2I);
      const double nD = fmax(xup, upD + T_D2D);
      xdiag = xup;
      curM = nM; curI = nI; curD = nD;
      ltv = (j == n - 1) ? nM : ltv;                       // last read column
      if (last_lane && j >= 0 && j < n) s_out[j] = make_double2(nM, nD);      // the band's last row: the next band's top, or rowP
      diagM = upM; diagD = upD;
    }
    if (arow) lt[myrow] = ltv;
    wave_lds_sync();
    if (last_band && LEAD) for (int j = lane; j < n; j += 64) rowp[j] = s_out[j].x;
  }
}

// ------------------------------------------------------------------ the STR block
struct StrCtx {
  int B, p, nd;
  double cst;        // lane t<20 holds f64pool[f64_off+t]: pmf[13] | prior_ins | prior_del[6]
  const hs_stropt_t* so;
  const uint8_t* blk;   // block bases of the current allele, in LDS
};
__device__ __forceinline__ uint8_t blk_at(const StrCtx& c, int x){ return c.blk[x]; }   // x wave-uniform: LDS broadcast

struct StrLds {     // one side of one read
  double2* bq;      // [n] (log P(correct), log P(error)) per read column
  double*  rowP;    // [n] M of the haplotype row preceding the STR block (this allele's leading flank)
  double*  Mt;      // [n] StutterAligner match_probs_
  double*  Dl;      // [6][ld] StutterAligner del_probs_
  uint8_t* rd;      // [n] read bases
  const double* ilog;   // [HS_ILOG_LDS] LDS copy of int_log(0..): ln of block-length-sized integers
  double*  nd;      // [HS_ND_TOTAL] deletion start values of the columns within |D| of the read end, the six sizes back to back
  double*  cstl;    // [20] pmf[13] | prior_ins | prior_del[6] of the current allele
  double*  tab;     // [2][HS_TAB_CAP] tabulated closed form of the current allele's simple lists: A | G (layout.h tab_*)
  uint8_t* blk;     // [blk_len] block bases of the current allele
  int ld;
};
// (HS_ND_TOTAL, HS_WAVE_LDS: layout.h — the host checks a locus' LDS need with the same numbers)

// Marginalisation over the artifact position (StutterAlignerClass.cpp:59-104 insertion, :106-150 deletion).
// The loop over block offsets is the same for every read column, so the host enumerated it (hs_visit_t) and
// the wave replays it in lock step; a lane drops out once the offset reaches its own bound `lim`.
//   lp0     value for the artifact at the block's right end (position 0)
//   nsub    read bases whose emission changes when the artifact moves one base left: D/p for an insertion, 1 for a deletion
//   stride  distance between those bases: p for an insertion, 0 for a deletion
//   tail    number of remaining equal-likelihood configurations is (tail - offset): B (insertion) or B+D (deletion)
__device__ __forceinline__ double visit_eval(const hs_dev_t& d, const StrLds& L, int j, double lp0, int lim, int limmax,
                                             const hs_visit_t& bundle, int rel, const hs_visit_t* __restrict__ list, int llen,
                                             int nsub, int stride, int tail){
  // `bundle` holds the first 64 visiting-list entries of this STR option, one per lane (all seven lists back to back);
  // the list evaluated here starts at bundle entry `rel` and at `list` in memory (used only beyond the bundle).
  Lse acc;
  for (int pass = 0; pass < 2; pass++){
    double lp = lp0;
    acc.start(pass, lp0);
    acc.push(pass, lp0, d.log_thresh);
    int nistop = 0; bool stopped = false;
    for (int v = 0; v < llen; v++){
      uint64_t meta; double logU;
      if (rel + v < 64){ meta = rdlane(bundle.meta, rel + v); logU = rdlane(bundle.logU, rel + v); }
      else { const hs_visit_t e = list[v]; meta = rdlane(e.meta, 0); logU = rdlane(e.logU, 0); }
      const int ni = (int)(meta & 0xffff);
      if (ni >= limmax){ if (!stopped) nistop = ni; break; }
      const bool act = ni < lim;
      if (!act && !stopped){ nistop = ni; stopped = true; }
      const int U = (int)((meta >> 16) & 0xffff);
      if ((meta >> 48) & 1){ if (act) acc.push(pass, lp, d.log_thresh); }
      else if (U == 0){
        const uint8_t ca = (uint8_t)(meta >> 32), cb = (uint8_t)(meta >> 40);
        for (int m = 1; m <= nsub; m++){
          const int pos = j - ni - m*stride;            // negative only on lanes that are past their bound (not accumulated)
          const uint8_t r = L.rd[pos]; const double2 bq = L.bq[pos];
          if (act){ lp -= emit(r, ca, bq); lp += emit(r, cb, bq); }
        }
        if (act) acc.push(pass, lp, d.log_thresh);
      } else {
        if (act) acc.push(pass, logU + lp, d.log_thresh);
      }
    }
    if (nistop < tail) acc.push(pass, L.ilog[tail - nistop] + lp, d.log_thresh);
  }
  return acc.finish();
}

// Closed form of visit_eval for a "simple" visiting list (hs_stropt_t::shape = U0 >= 0): every pushed value is
// lp0 plus a constant, so the log-sum-exp needs no list traversal.  Pushes, in the reference's order:
//   lp0 | [0 < lim] ln(U0) + lp0 (run of U0 equal configurations, only if U0 > 0) | lp0 once per plain offset in
//   [U0, lim) | [stop < tail] ln(tail - stop) + lp0, stop = first visited offset >= lim.
__device__ __forceinline__ double simple_eval(const hs_dev_t& d, const StrLds& L, double lp0, int lim, int U0, int tail){
  const bool skip = (U0 > 0) && (lim > 0);
  const int start = U0;                                         // first plain offset
  const int nplain = max(0, lim - start);
  const int stop = (lim <= 0) ? 0 : ((U0 > 0 && lim <= U0) ? U0 : lim);
  const bool has_tail = stop < tail;
  const double v_skip = L.ilog[U0] + lp0;
  const double v_tail = L.ilog[max(tail - stop, 0)] + lp0;
  double mx = lp0;
  if (skip) mx = fmax(mx, v_skip);
  if (has_tail) mx = fmax(mx, v_tail);
  // branch-free: an absent or thresholded term adds +0.0, which leaves the (non-negative) sum unchanged bit for bit
  const double d0 = lp0 - mx, d1 = v_skip - mx, d2 = v_tail - mx;
  double tot = (d0 > d.log_thresh) ? (double)(1 + nplain) * (double)f_fasterexp((float)d0) : 0.0;   // equal float terms: the product is exact
  tot += (skip && d1 > d.log_thresh) ? (double)f_fasterexp((float)d1) : 0.0;
  tot += (has_tail && d2 > d.log_thresh) ? (double)f_fasterexp((float)d2) : 0.0;
  return mx + (double)f_fasterlog((float)tot);
}

// Closed form for a "piecewise simple" visiting list (hs_stropt_t::shape = HS_SHAPE_PIECEWISE; descriptor slots written by prep.cpp):
//   [run?] break [run?] (break [run?])  plain ... plain  terminal
// The running likelihood only changes at the (one or two) break entries, so there are at most three levels L0, L1, L2 and the pushes of
// visit_eval are: L0 | ln U + L of a run entry | L after a break | L of the last level once per plain entry | the tail term — each
// present only if its offset is below the lane's bound.  Same values, same float log-sum-exp as the replay, without replaying.
//   pwA/pwB: the 70 descriptor slots of the STR option, one per lane (64 + 6); k: list index (0..5 deletion lists, 6 insertion list)
__device__ __forceinline__ double pw_eval(const hs_dev_t& d, const StrLds& L, int j, double lp0, int lim, double pwA, double pwB, int k,
                                          int nsub, int stride, int tail){
  auto lo = [&](int sl){ const int g = k*HS_PW_SLOTS + sl; return g < 64 ? rdlane(__double2loint(pwA), g) : rdlane(__double2loint(pwB), g - 64); };
  auto hi = [&](int sl){ const int g = k*HS_PW_SLOTS + sl; return g < 64 ? rdlane(__double2hiint(pwA), g) : rdlane(__double2hiint(pwB), g - 64); };
  auto dbl = [&](int sl){ return __hiloint2double(hi(sl), lo(sl)); };
  const int nseg = lo(0), term_ni = hi(0);
  const int r0 = lo(1), U0 = hi(1), r1 = lo(3), U1 = hi(3), r2 = lo(5), U2 = hi(5);
  const int b0 = lo(7), c0 = hi(7), b1 = lo(8), c1 = hi(8);
  const int pa = lo(9), pb = hi(9);
  // the levels
  const double L0 = lp0;
  double L1 = L0;
  const bool a_b0 = b0 < lim;                              // nseg >= 1 always
  {
    const uint8_t ca = (uint8_t)c0, cb = (uint8_t)(c0 >> 8);
    for (int m = 1; m <= nsub; m++){
      const int pos = j - b0 - m*stride;
      const uint8_t r = L.rd[pos]; const double2 bq = L.bq[pos];
      if (a_b0){ L1 -= emit(r, ca, bq); L1 += emit(r, cb, bq); }
    }
  }
  double L2 = L1;
  const bool a_b1 = (nseg >= 2) && (b1 < lim);
  if (nseg >= 2){
    const uint8_t ca = (uint8_t)c1, cb = (uint8_t)(c1 >> 8);
    for (int m = 1; m <= nsub; m++){
      const int pos = j - b1 - m*stride;
      const uint8_t r = L.rd[pos]; const double2 bq = L.bq[pos];
      if (a_b1){ L2 -= emit(r, ca, bq); L2 += emit(r, cb, bq); }
    }
  }
  const double Llast = (nseg >= 2) ? L2 : L1;
  const double Lfin = a_b1 ? L2 : (a_b0 ? L1 : L0);         // what the lane's running value is when its replay stops
  const bool a_r0 = (U0 > 0) && (r0 < lim), a_r1 = (U1 > 0) && (r1 < lim), a_r2 = (U2 > 0) && (r2 < lim);
  const double v_r0 = dbl(2) + L0, v_r1 = dbl(4) + L1, v_r2 = dbl(6) + L2;
  const int np = min(max(lim - pa, 0), pb - pa);
  // first offset of the list at or beyond the bound (the replay's nistop)
  int ns = term_ni;
  if (pb > pa) ns = (lim < pb) ? max(lim, pa) : ns;
  if (U2 > 0) ns = (r2 >= lim) ? r2 : ns;
  if (nseg >= 2) ns = (b1 >= lim) ? b1 : ns;
  if (U1 > 0) ns = (r1 >= lim) ? r1 : ns;
  ns = (b0 >= lim) ? b0 : ns;
  if (U0 > 0) ns = (r0 >= lim) ? r0 : ns;
  const bool a_t = ns < tail;
  const double v_t = L.ilog[max(tail - ns, 0)] + Lfin;
  double mx = L0;
  mx = a_r0 ? fmax(mx, v_r0) : mx;  mx = a_b0 ? fmax(mx, L1) : mx;  mx = a_r1 ? fmax(mx, v_r1) : mx;
  mx = a_b1 ? fmax(mx, L2) : mx;    mx = a_r2 ? fmax(mx, v_r2) : mx;  mx = (np > 0) ? fmax(mx, Llast) : mx;
  mx = a_t ? fmax(mx, v_t) : mx;
  auto term = [&](bool on, double v){ const double dd = v - mx; return (on && dd > d.log_thresh) ? (double)f_fasterexp((float)dd) : 0.0; };
  double tot = term(true, L0);
  tot += term(a_r0, v_r0); tot += term(a_b0, L1); tot += term(a_r1, v_r1); tot += term(a_b1, L2); tot += term(a_r2, v_r2);
  tot += (double)np * term(np > 0, Llast);                   // equal float terms: the product is exact
  tot += term(a_t, v_t);
  return mx + (double)f_fasterlog((float)tot);
}


// pw_eval for the grouped layout (hs_str_group_kernel_pw): the same values pushed into the same float log-sum-exp, with
//   * the list's ten descriptor slots in scalar registers (`pws`: scalar loads from the f64 pool, the option is the same for every lane);
//   * the emission of read column c against block base b = the column's entry of base b's plane of the group's emission table
//     (E[plane][column]: what emit() picks from the read base and the two quality logs);
//   * the terms a descriptor rules out for every lane (no run in a segment, one break only, no plain entries) skipped by scalar branches —
//     an absent term adds +0.0 to the non-negative sum in pw_eval.
// xx: the lane's column in the group's tables;  Eb: byte address (LDS) of E[0][0];  plane stride XC*8 bytes.
struct PwSlots { int v[2*HS_PW_SLOTS]; };

// visit_eval for the grouped layout (round 4): a list that has no closed form — three and more interruptions of the repeat — replayed
// entry by entry like visit_eval, but inside hs_str_group_kernel_pw: the entries are the same for every lane (scalar loads from the visit
// pool), the emission of (column, block base) is one LDS read from the group's table, a lane drops out at its own bound.  Same pushes,
// same float log-sum-exp (two passes: the maximum, then the sum of the float exponentials in double).
template <int XC>
__device__ __forceinline__ double visit_eval_grp(const hs_visit_t* __restrict__ list_g, int llen, const double* ilog, double log_thresh, int Eb, int xx,
                                              double lp0, int lim, int limmax, int nsub, int stride, int tail){
  auto ldb = [&](int byte_addr) -> double { return *(const __attribute__((address_space(3))) double*)(uintptr_t)(uint32_t)byte_addr; };
  // the entries through the constant address space: the address is the same for every lane, so these are scalar loads (s_load_dwordx4 from
  // the scalar cache, shared by the passes and the workgroup's wavefronts) — as vector loads + readfirstlane each entry was a dependent trip to L2
  typedef const __attribute__((address_space(4))) hs_visit_t* hs_visit_k;
  const hs_visit_k list = (hs_visit_k)(uintptr_t)list_g;
  Lse acc;
  double lp = lp0;
  int nistop = 0;
  for (int pass = 0; pass < 2; pass++){
    lp = lp0;
    acc.start(pass, lp0);
    acc.push(pass, lp0, log_thresh);
    nistop = 0; bool stopped = false;
    for (int v = 0; v < llen; v++){
      const uint64_t meta = list[v].meta;
      const int ni = (int)(meta & 0xffff);
      if (ni >= limmax){ if (!stopped) nistop = ni; break; }
      const bool act = ni < lim;
      if (!act && !stopped){ nistop = ni; stopped = true; }
      const int U = (int)((meta >> 16) & 0xffff);
      if ((meta >> 48) & 1){ if (act) acc.push(pass, lp, log_thresh); }
      else if (U == 0){
        const int pla = (((int)(meta >> 33)) & 3) * (XC*8), plb = (((int)(meta >> 41)) & 3) * (XC*8);      // base code ((c >> 1) & 3) -> plane of the emission table
        double t = lp;
        for (int m = 1; m <= nsub; m++){
          const int ca = Eb + 8*max(xx - ni - m*stride, 0);        // a lane past its bound may point in front of its read: not used
          const double ea = ldb(ca + pla), eb = ldb(ca + plb);
          t -= ea; t += eb;
        }
        lp = act ? t : lp;
        if (act) acc.push(pass, lp, log_thresh);
      } else {
        const double logU = list[v].logU;
        if (act) acc.push(pass, logU + lp, log_thresh);
      }
    }
    if (nistop < tail) acc.push(pass, ilog[max(tail - nistop, 0)] + lp, log_thresh);
  }
  return acc.finish();
}

// ------------------------------------------------------------------ the closed forms of the grouped kernels' interrupted lists (round 5's lean forms; the round-4
// evaluators pw_eval_grp / pwk_eval_grp they replaced and the list LOOP they were first written for are gone: measured slower, profiles/r05_notes.md).
// The piecewise closed form with up to HS_PWK_MAX breaks (prep.cpp piecewise_k, layout.h HS_PWK_SLOTS): lists of blocks with two or three
// interruptions — [run?] break [run?] break ... [run?] plain ... plain terminal.  The running likelihood changes only at the breaks, so the
// pushes of visit_eval_grp are: L0 | ln U_s + L_s of segment s' run | L_{s+1} behind break s | L_nseg once per plain entry | the tail term —
// each present only if its offset is below the lane's bound.  One pass over the segments builds the levels (kept in registers), the maximum
// and the first offset at or beyond the bound; a second one sums the float exponentials, two per packed operation.  Same values into the
// same float log-sum-exp as the replay (the float terms are summed in double: exact in any order).  What keeps the instruction count down:
//   * the emission of (column xx - off, block base c) is one v_add: A0 = byte address of the lane's column in plane 0 of the emission
//     table, minus a scalar 8 off - plane(c).  No clamp: the table lies 16 KB into the carve (deletion table, rowP, match_probs_ in front),
//     a bound is at most 1024 + 36 columns, so a lane past its bound reads some double in front of its read that the level's select drops;
//   * an absent value is NEG in its HIGH word only (one v_cndmask);
//   * the float exponential's bits without v_cvt_u32_f32: for a term that passes the threshold y = 1.442695040f dd + 126.94269504f lies in
//     [116.9, 126.95] ⊂ [64, 128), so 2^23 y is the integer (2^23 + mantissa) 2^6 — (bits(y) << 6) + 2^31 (mod 2^32) — exactly what
//     fasterexp converts (fastonebigheader.h:206-218); a term under the threshold is switched off as before;
//   * the switch acts on the 32-bit float, not on its double.
// Same values into the same float log-sum-exp as the entry-by-entry replay (tests/cpp/pwk_form_test.cpp, GPU suite + fuzzers).
typedef int hs_i2k __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(4))) hs_i2k* hs_slot_k;
typedef float hs_f2 __attribute__((ext_vector_type(2)));
struct LeanAcc {
  double tot; double mx; double log_thresh;
  __device__ __forceinline__ static double neg_unless(bool on, double v){       // v where on, else a value near -1e300 (high word of NEG, low word of v)
    return __hiloint2double(on ? __double2hiint(v) : (int)0xFE37E43Cu, __double2loint(v));
  }
  // two pushed values: a (counted wa times: 1.0 but for the plain entries) where on_a and over the threshold, b where over the threshold
  __device__ __forceinline__ void pair(double a, bool on_a, double b, double wa, bool weighted){
    const double dd0 = a - mx, dd1 = b - mx;
    const bool t0 = on_a && dd0 > log_thresh, t1 = dd1 > log_thresh;
    hs_f2 x; x.x = (float)dd0; x.y = (float)dd1;
    const hs_f2 y = x * 1.442695040f + 126.94269504f;
    uint32_t e0 = (__float_as_uint(y.x) << 6) + 0x80000000u, e1 = (__float_as_uint(y.y) << 6) + 0x80000000u;
    e0 = t0 ? e0 : 0u; e1 = t1 ? e1 : 0u;
    asm volatile("" : "+v"(e0), "+v"(e1));                      // (the selects stay on the 32-bit values)
    if (weighted) tot += wa * (double)__uint_as_float(e0); else tot += (double)__uint_as_float(e0);
    tot += (double)__uint_as_float(e1);
  }
  __device__ __forceinline__ void single(double a, bool on_a){
    const double dd = a - mx;
    const float y = (float)dd * 1.442695040f + 126.94269504f;
    uint32_t e = (__float_as_uint(y) << 6) + 0x80000000u;
    e = (on_a && dd > log_thresh) ? e : 0u;
    asm volatile("" : "+v"(e));
    tot += (double)__uint_as_float(e);
  }
  __device__ __forceinline__ void one(double b){
    const double dd = b - mx;
    const float y = (float)dd * 1.442695040f + 126.94269504f;
    uint32_t e = (__float_as_uint(y) << 6) + 0x80000000u;
    e = (dd > log_thresh) ? e : 0u;
    asm volatile("" : "+v"(e));
    tot += (double)__uint_as_float(e);
  }
};
__device__ __forceinline__ double lds_f64(int byte_addr){ return *(const __attribute__((address_space(3))) double*)(uintptr_t)(uint32_t)byte_addr; }
__device__ __forceinline__ int plane_of(int ch, int plane_bytes){ return ((ch >> 1) & 3) * plane_bytes; }

// The level behind a break: t - e(column - m stride, base a) + e(column - m stride, base b) for m = 1..nsub, in that order
// (StutterAlignerClass.cpp:78-82, :128-129).
__device__ __forceinline__ double emis_chain(double t, int A0, int sa, int sb, int nsub, int stride8){
  int ao = 0;
  for (int m = 1; m <= nsub; m++){
    ao += stride8;
    const double ea = lds_f64(A0 + (sa - ao)), eb = lds_f64(A0 + (sb - ao));
    t -= ea; t += eb;
  }
  return t;
}

// One or two breaks, ten slots (the grouped layout's pw_eval).  A0: see above; stride8 = 8 x stride.
template <int XC>
__device__ __forceinline__ double pw_eval_lean(const PwSlots& S, const double* ilog, double log_thresh, int A0, double lp0, int lim,
                                               int nsub, int stride8, int tail){
  auto sl = [&](int i){ hs_i2k r; r.x = S.v[2*i]; r.y = S.v[2*i + 1]; return r; };
  const hs_i2k d0 = sl(0), d1 = sl(1), d2 = sl(2), d3 = sl(3), d4 = sl(4), d5 = sl(5), d6 = sl(6), d7 = sl(7), d8 = sl(8), d9 = sl(9);
  const int nseg = d0.x, term_ni = d0.y;
  const int r0 = d1.x, U0 = d1.y, r1 = d3.x, U1 = d3.y, r2 = d5.x, U2 = d5.y;
  const int b0 = d7.x, c0 = d7.y, b1 = d8.x, c1 = d8.y, pa = d9.x, pb = d9.y;
  const double L0 = lp0;
  double L1, L2;
  const bool a_b0 = b0 < lim;
  {
    const int sa = plane_of(c0, XC*8) - 8*b0, sb = plane_of(c0 >> 8, XC*8) - 8*b0;
    const double t = emis_chain(L0, A0, sa, sb, nsub, stride8);
    L1 = a_b0 ? t : L0;
  }
  L2 = L1;
  bool a_b1 = false;
  if (nseg >= 2){
    a_b1 = b1 < lim;
    const int sa = plane_of(c1, XC*8) - 8*b1, sb = plane_of(c1 >> 8, XC*8) - 8*b1;
    const double t = emis_chain(L1, A0, sa, sb, nsub, stride8);
    L2 = a_b1 ? t : L1;
  }
  const int np = min(max(lim - pa, 0), pb - pa);
  int ns = term_ni;
  if (pb > pa) ns = (lim < pb) ? max(lim, pa) : ns;
  if (U2 > 0) ns = (r2 >= lim) ? r2 : ns;
  if (nseg >= 2) ns = (b1 >= lim) ? b1 : ns;
  if (U1 > 0) ns = (r1 >= lim) ? r1 : ns;
  ns = (b0 >= lim) ? b0 : ns;
  if (U0 > 0) ns = (r0 >= lim) ? r0 : ns;
  constexpr double NEG = -1.0e300;
  const double v1 = (U0 > 0) ? LeanAcc::neg_unless(r0 < lim, __hiloint2double(d2.y, d2.x) + L0) : NEG;
  const double v3 = (U1 > 0) ? LeanAcc::neg_unless(r1 < lim, __hiloint2double(d4.y, d4.x) + L1) : NEG;
  const double v5 = (U2 > 0) ? LeanAcc::neg_unless(r2 < lim, __hiloint2double(d6.y, d6.x) + L2) : NEG;
  const double v7 = LeanAcc::neg_unless(ns < tail, ilog[max(tail - ns, 0)] + L2);
  double mx = fmax(L0, L1);                                    // (a level that is not the lane's equals the one before it)
  mx = fmax(mx, L2);
  if (U0 > 0) mx = fmax(mx, v1);
  if (U1 > 0) mx = fmax(mx, v3);
  if (U2 > 0) mx = fmax(mx, v5);
  mx = fmax(mx, v7);
  LeanAcc acc; acc.log_thresh = log_thresh; acc.tot = 0.0; acc.mx = mx;
  if (U0 > 0) acc.pair(L0, true, v1, 1.0, false); else acc.single(L0, true);
  if (U1 > 0) acc.pair(L1, a_b0, v3, 1.0, false); else acc.single(L1, a_b0);
  if (nseg >= 2){ if (U2 > 0) acc.pair(L2, a_b1, v5, 1.0, false); else acc.single(L2, a_b1); }
  else if (U2 > 0) acc.one(v5);
  if (pb > pa) acc.pair(L2, np > 0, v7, (double)np, true);
  else acc.one(v7);
  return mx + (double)f_fasterlog((float)acc.tot);
}
template <int XC>
__device__ __forceinline__ double pw_eval_lean(const double* __restrict__ slots_g, const double* ilog, double log_thresh, int A0, double lp0, int lim,
                                               int nsub, int stride8, int tail){
  const hs_slot_k D = (hs_slot_k)(uintptr_t)slots_g;
  PwSlots S;
#pragma unroll
  for (int t = 0; t < HS_PW_SLOTS; t++){ const hs_i2k q = D[t]; S.v[2*t] = q.x; S.v[2*t + 1] = q.y; }
  return pw_eval_lean<XC>(S, ilog, log_thresh, A0, lp0, lim, nsub, stride8, tail);
}


// Three to HS_PWK_MAX breaks, 24 slots: the list's slots are fetched together (three wide scalar loads, one
// wait) and stay in scalar registers through both passes.
template <int XC, int NSEG>      // NSEG > 0: the list's number of breaks, known at compile time; 0: read from the slots
__device__ __forceinline__ double pwk_eval_lean_n(const hs_i2k (&ds)[HS_PWK_SLOTS], const double* ilog, double log_thresh, int A0, double lp0, int lim,
                                                  int nsub, int stride8, int tail){
  const int nseg = NSEG > 0 ? NSEG : ds[0].x, term_ni = ds[0].y, pa = ds[1].x, pb = ds[1].y;
  __builtin_assume(nseg >= 3 && nseg <= HS_PWK_MAX);          // (prep.cpp piecewise_k: what makes a list this shape) — the first three segments without scalar guards
  constexpr double NEG = -1.0e300;
  double Lv[HS_PWK_MAX + 1];
  Lv[0] = lp0;
  double mx = lp0;
  unsigned u = (unsigned)(term_ni - lim);                     // first offset at or beyond the bound, minus the bound (the terminal entry is one)
#pragma unroll
  for (int s = 0; s <= HS_PWK_MAX; s++){
    if (s < HS_PWK_MAX) Lv[s + 1] = Lv[s];
    if (s <= nseg){
      const hs_i2k run = ds[2 + 3*s];
      if (run.y > 0){
        const double v = __hiloint2double(ds[3 + 3*s].y, ds[3 + 3*s].x) + Lv[s];
        const bool act = run.x < lim;
        mx = fmax(mx, LeanAcc::neg_unless(act, v));
        u = min(u, (unsigned)(run.x - lim));
      }
      if (s < HS_PWK_MAX && s < nseg){
        const hs_i2k brk = ds[4 + 3*s];
        const int b = brk.x;
        const int sa = plane_of(brk.y, XC*8) - 8*b, sb = plane_of(brk.y >> 8, XC*8) - 8*b;
        const double t = emis_chain(Lv[s], A0, sa, sb, nsub, stride8);
        const bool act = b < lim;
        Lv[s + 1] = act ? t : Lv[s];                             // an unreached level equals the one before it: harmless in the maximum
        mx = fmax(mx, Lv[s + 1]);
        u = min(u, (unsigned)(b - lim));
      }
    }
  }
  double Llast = Lv[0];
#pragma unroll
  for (int s = 1; s <= HS_PWK_MAX; s++) Llast = (s <= nseg) ? Lv[s] : Llast;       // (scalar condition)
  const int np = min(max(lim - pa, 0), pb - pa);
  if (pb > pa) u = min(u, (lim < pb) ? (unsigned)max(pa - lim, 0) : 0xffffffffu);
  const int ns = (int)u + lim;
  const double v_t = LeanAcc::neg_unless(ns < tail, ilog[max(tail - ns, 0)] + Llast);
  mx = fmax(mx, v_t);
  LeanAcc acc; acc.log_thresh = log_thresh; acc.tot = 0.0; acc.mx = mx;
#pragma unroll
  for (int s = 0; s <= HS_PWK_MAX; s++){
    if (s <= nseg){
      const bool on = (s == 0) ? true : (ds[4 + 3*(s - 1)].x < lim);
      const hs_i2k run = ds[2 + 3*s];
      if (run.y > 0){
        const double v = __hiloint2double(ds[3 + 3*s].y, ds[3 + 3*s].x) + Lv[s];
        acc.pair(Lv[s], on, LeanAcc::neg_unless(run.x < lim, v), 1.0, false);
      } else acc.single(Lv[s], on);
    }
  }
  if (pb > pa) acc.pair(Llast, np > 0, v_t, (double)np, true);                    // equal float terms: the product is exact
  else acc.one(v_t);
  return mx + (double)f_fasterlog((float)acc.tot);
}
template <int XC>
__device__ __forceinline__ double pwk_eval_lean(const double* __restrict__ slots_g, const double* ilog, double log_thresh, int A0, double lp0, int lim,
                                                int nsub, int stride8, int tail){
  const hs_slot_k D = (hs_slot_k)(uintptr_t)slots_g;
  hs_i2k ds[HS_PWK_SLOTS];
#pragma unroll
  for (int t = 0; t < HS_PWK_SLOTS; t++) ds[t] = D[t];
  return pwk_eval_lean_n<XC, 0>(ds, ilog, log_thresh, A0, lp0, lim, nsub, stride8, tail);
}

}  // namespace

#ifndef HS_STR_WAVES
#define HS_STR_WAVES 4      // workgroups (2 wavefronts) per SIMD pair the register allocation aims at
#endif
extern __shared__ double hs_lds_raw[];


// LDS bytes of one hs_str_kernel workgroup (both sides of a read) for a batch whose longest read has lds_len bases.
extern "C" size_t hs_str_lds_bytes(int lds_len, int max_B){ return hs_str_kernel_lds_bytes(lds_len, max_B); }

// Workgroup = one active read: wave 0 the left side, wave 1 the right side (independent; they share only the LDS
// carve, whose per-column arrays are exactly len-1 long in total).  Writes M of the STR block's last row for every
// realigned allele of the chunk to the MR workspace.
// Two instantiations share the body, each with the register budget of its own evaluators:
//   MODE 0 (hs_str_kernel)          alleles whose visiting lists are all simple and tabulated: positions [0, n_tab) of the side's order
//   MODE 1 (hs_str_kernel_generic)  the other alleles, positions [n_tab, n_re): closed forms the long way and list replay
template <int MODE>
__device__ __forceinline__ void str_body(const hs_dev_t& d, int active_begin, int only_long){
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const SideView v = side_view(d, active_begin + blockIdx.x, w);
  const int n = v.n;
  const int Lc = (d.lds_len + 3) & ~1;
  const int o = w ? ((v.nL + 1) & ~1) : 0;
  StrLds L;
  {
    // Dl first: the read-end deletion loop below indexes bq / rd / blk with (start - t) and relies on masking instead of clamping,
    // so up to n entries BEFORE an array may be touched (never used); everything in front of bq, blk and rd is this workgroup's LDS
    double* Dl = (double*)hs_lds_raw;
    double2* bq = (double2*)(Dl + HS_MAXREP*Lc);
    double* rowP = (double*)(bq + Lc);
    double* Mt = rowP + Lc;
    const int ilog_len = (d.max_B + 9) & ~1, blk_len = (d.max_B + 19) & ~15;
    double* ilog = Mt + Lc;
    double* ndb = ilog + ilog_len;                          // per wave: nd[HS_ND_TOTAL] | cstl[24] | tab[2][HS_TAB_CAP]
    uint8_t* blkb = (uint8_t*)(ndb + 2*HS_WAVE_LDS);
    uint8_t* rdb = blkb + 2*blk_len;
    L.bq = bq + o; L.rowP = rowP + o; L.Mt = Mt + o; L.Dl = Dl + o; L.rd = rdb + o; L.ilog = ilog; L.ld = Lc;
    L.nd = ndb + w*HS_WAVE_LDS; L.cstl = L.nd + HS_ND_TOTAL; L.tab = L.cstl + 24; L.blk = blkb + w*blk_len;
    for (int i = threadIdx.x; i < ilog_len; i += 128) ilog[i] = d.int_log[i];
  }
  __syncthreads();
  const int ncyc = (n + 63) / 64;
  // alleles in the side's processing order (nested STR blocks follow each other); a chunk of positions per workgroup
  const int n_tab = uni(v.loc->n_tab[w]);
  // MODE 1: the piecewise alleles [n_tab, n_pw) of a side that fits a group are hs_str_group_kernel_pw's (pw_grouped)
  const int n_gen = (MODE == 1 && only_long /* = pw_grouped */ && n <= HS_GRP_COLS) ? uni(v.loc->n_rp[w]) : n_tab;
  const int r_lo = MODE == 0 ? 0 : n_gen, r_hi = MODE == 0 ? n_tab : uni(v.loc->n_re);
  const int i0 = r_lo + blockIdx.y * d.allele_chunk, i1 = min(r_hi, i0 + d.allele_chunk);
  const int32_t* order = d.str_order + uni(v.loc->order_off[w]);
  // MODE 1 also re-does, the long way, the tabulated alleles of this read for which hs_str_kernel left HS_REDO marks (a lane's
  // lp0 was too large for the table's guarantee): positions [j0, j1) of the tabulated range, looked at only if the read is flagged
  const int ai = active_begin + blockIdx.x;
  const bool redo = (MODE == 1) && uni(d.redo[ai]) != 0;
  const int j0 = blockIdx.y * d.allele_chunk, j1 = redo ? min(MODE == 1 ? n_gen : n_tab, j0 + d.allele_chunk) : j0;
  if (i0 >= i1 && j0 >= j1) return;           // nothing of this kind for this side (after the barrier: the other side may have work)
  if (MODE == 0 && only_long && n <= HS_GRP_COLS) return;     // this side's columns fit a group: hs_str_group_kernel has it
  for (int j = lane; j < n; j += 64){
    const int src = v.base_off + (w ? v.len - 1 - j : j);
    const uint8_t q = (uint8_t)d.quals[src];
    L.rd[j] = (uint8_t)d.bases[src];
    L.bq[j] = make_double2(d.qual_correct[q], d.qual_error[q]);
  }
  int cur_slot = -1, prev_B = 0;
  const int n_own = max(i1 - i0, 0), n_all = n_own + ((MODE == 1) ? max(j1 - j0, 0) : 0);
  for (int it = 0; it < n_all; it++){
    const bool own = it < n_own;                      // false: a candidate of the re-do scan
    const int i = own ? i0 + it : j0 + (it - n_own);
    const int oe = uni(order[i]);
    const bool chained = own && (it > 0) && ((oe >> 30) & 1);
    const hs_allele_t* al = d.alleles + uni(v.loc->hap_begin) + (oe & 0x1fffffff);
    const int slot = uni(al->lead_slot[w]), str_opt = uni(al->str_opt[w]);
    double* mr_out = d.ws_mr + v.ws_mr + (int64_t)uni(al->re_ord)*(v.len-1) + (w ? v.nL : 0);
    if (MODE == 1 && !own){
      bool marked = false;                            // hs_str_group_kernel marks the columns of a wavefront, which may be any stretch of the side
      for (int kk = 0; kk < ncyc; kk++){ const int jj = lane + 64*kk; marked |= (jj < n) && (mr_out[min(jj, n-1)] == HS_REDO); }
      if (!__any(marked)) continue;
    }
    wave_lds_sync();                // the previous allele's readers of rowP/Mt/Dl are done
    if (slot != cur_slot){          // M of the row before the STR block, from the leading-flank kernel
      const double* rec = lead_record(d, v, slot);
      for (int j = lane; j < n; j += 64) L.rowP[j] = rec[j];
      cur_slot = slot;
    }

    StrCtx c;
    c.so = d.stropts + str_opt;
    // the option's scalars come through the scalar cache in three loads (the record is wave-uniform and read-only; the compiler cannot
    // prove the latter next to this kernel's stores and would fetch every field with a vector load and a v_readfirstlane)
    typedef int hs_i8v __attribute__((ext_vector_type(8)));
    typedef int hs_i2v __attribute__((ext_vector_type(2)));
    hs_i8v so_head; hs_i2v so_tab; int so_ndeq;
    {
      const uint64_t sop = (uint64_t)(uintptr_t)(const hs_stropt_t*)c.so;
      asm volatile("s_load_dwordx8 %0, %3, 0x0\n\ts_load_dwordx2 %1, %3, 0x68\n\ts_load_dword %2, %3, 0x8c\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(so_head), "=&s"(so_tab), "=&s"(so_ndeq) : "s"(sop) : "memory");
    }
    static_assert(offsetof(hs_stropt_t, tab_off) == 0x68 && offsetof(hs_stropt_t, nd_eq) == 0x8c && offsetof(hs_stropt_t, ins_len) == 24, "hs_stropt_t layout");
    const int so_seq_off = so_head[0], so_f64_off = so_head[4];
    c.B = so_head[1]; c.nd = so_head[2]; c.p = so_head[3];
    const int nd_eq = so_ndeq;
    c.cst = d.f64pool[so_f64_off + min(lane, 19)];
    c.blk = L.blk;
    const int B = c.B, p = c.p;
    // all seven visiting lists of the option sit back to back in memory: one coalesced 16 B/lane load brings the first
    // 64 entries; list offsets/lengths ride in one lane-indexed register (lane q: deletion list q, lane 6: insertion list)
    const int ins_off = so_head[5], ins_len = so_head[6];
    const hs_visit_t* ins_list = d.visits + ins_off;
    const int total = uni(c.so->del_off[HS_MAXREP-1]) + uni(c.so->del_len[HS_MAXREP-1]) - ins_off;
    const int shapes = (lane <= HS_MAXREP) ? c.so->shape[lane] : -1;
    // deletion sizes larger than the block have no list (shape -1) but are never evaluated
    const bool all_simple = __all((lane > HS_MAXREP) || (shapes >= 0) || (lane < HS_MAXREP && B - (lane+1)*p < 0));
    const bool all_closed = __all((lane > HS_MAXREP) || (shapes >= 0) || (shapes == HS_SHAPE_PIECEWISE) || (lane < HS_MAXREP && B - (lane+1)*p < 0));

    {
      const int* src = (const int*)(d.chars + so_seq_off);
      for (int i = lane; i < (B + 3)/4; i += 64) ((int*)L.blk)[i] = src[i];
    }
    if (lane < 20) L.cstl[lane] = c.cst;
    // tabulated closed form of the simple lists (when prep.cpp could build it): entry bases ride in a lane-indexed register
    const int tab_len = so_tab[1];
    const bool use_tab = (MODE == 0);          // prep.cpp put exactly the alleles with all_simple && tab_len > 0 into [0, n_tab)
    const int tbase = (lane <= HS_MAXREP) ? c.so->tab_base[lane] : 0;
    double tab_bmin = 0.0;
    if (use_tab){
      const double* src = d.f64pool + so_tab[0];
      for (int e = lane; e < tab_len; e += 64){ L.tab[e] = src[3*e]; L.tab[HS_TAB_CAP + e] = src[3*e + 1]; }
      tab_bmin = uni(src[3*tab_len]);            // min over the entries' Bnd
    }
    wave_lds_sync();

    // --- StutterAlignerClass::load_read (StutterAlignerClass.cpp:12-53): match_probs_ and del_probs_
    // (a block that ends with the previous allele's block only appends terms to that allele's sums)
    const int t0 = chained ? min(prev_B, n) : 0;
    prev_B = B;
    const int tmax = min(B, n);
    if (t0 < tmax)                      // a continued block that already covered the whole read side adds nothing
    for (int kk = 0; kk < ncyc; kk++){
      const int j = min(lane + 64*kk, n-1);
      double lp = (t0 > 0) ? L.Mt[j] : 0.0;
      const int ndp = c.nd * p;
      int t = t0;
      // the first nd*p steps also fill the deletion table: one step at a time, the "(t+1) is a multiple of p" test carried as a counter.
      // Read position j - t is not clamped: a step with t > j is masked (Dl sits in front of bq)
      if (t < min(tmax, ndp)){
        const uint8_t* prd = L.rd + (j - t); const double2* pbq = L.bq + (j - t);
        int left = j - t, ph = (t + 1) % p;
        double* dl = L.Dl + ((t + 1)/p - 1)*L.ld + j;          // row of the deletion table the next multiple of p goes to
        for (; t < min(tmax, ndp); t++){
          const double e = emit(*prd, blk_at(c, B-1-t), *pbq);
          if (left >= 0) lp += e;
          if (ph == 0){ if (left >= 0) *dl = lp; }
          ph++; if (ph == p){ ph = 0; dl += L.ld; }
          prd--; pbq--; left--;
        }
      }
      // the rest only extends match_probs_: groups of four steps sharing one address (index kept opaque to the optimiser, immediate
      // offsets), without a mask while t <= 64 kk, the smallest column of this chunk
      auto steps = [&](int tend, auto masked){
        int xr = j - t - 3, xb = B - 1 - t - 3;
        for (; t + 4 <= tend; t += 4){
          asm volatile("" : "+v"(xr));
          const uint8_t* prd = L.rd + xr; const double2* pbq = L.bq + xr;
#pragma unroll
          for (int k = 0; k < 4; k++){
            const double e = emit(prd[3-k], blk_at(c, xb + 3 - k), pbq[3-k]);
            if (!decltype(masked)::value || xr + 3 - k >= 0) lp += e;
          }
          xr -= 4; xb -= 4;
        }
        for (; t < tend; t++){
          const double e = emit(L.rd[xr + 3], blk_at(c, xb + 3), L.bq[xr + 3]);
          if (!decltype(masked)::value || xr + 3 >= 0) lp += e;
          xr--; xb--;
        }
      };
      if (t < tmax){
        steps(min(tmax, 64*kk + 1), std::false_type());
        steps(tmax, std::true_type());
      }
      L.Mt[j] = lp;
    }
    wave_lds_sync();

    // --- deletion start values of the columns whose segment reaches the read end (the `else` branch of
    // StutterAlignerClass.cpp:117-120: a sequential sum starting from the position prior).  There are only
    // min(|D|, n) such columns per deletion size, so (size, column) pairs are spread over the lanes instead of
    // looping over the block once per deletion size.
    {
      // one round: every lane sums its (size q, column j) pair and stores it at nd[dst]
      auto nd_round = [&](int q, int j, bool valid, int dst){
        const int aD = (q+1)*p;
        const int len = min(B - aD, j + 1);
        // pairs are numbered size-major, so the lengths within a round are close: the first lmin steps need no mask at all, and the
        // round stops at its own longest sum
        const int lmin = uni(wave_min_i(len)), lmax = uni(wave_max_i(len));
        double lp = L.cstl[14 + q];
        // step t pairs read base j - t with block base B-1-aD - t.  xr / xb: lowest read / block index of the current group of four
        // steps (opaque to the optimiser, so that the four accesses become one address plus immediate offsets); in the masked loops an
        // index may fall in front of its array (Dl leads the LDS carve; in front of blk sit the per-wave tables): read, never used
        int xr = j - 3, xb = (B - 1 - aD) - 3, t = 0;
        for (; t + 4 <= lmin; t += 4){
          asm volatile("" : "+v"(xr), "+v"(xb));
          const uint8_t* prd = L.rd + xr; const double2* pbq = L.bq + xr; const uint8_t* pbk = L.blk + xb;
#pragma unroll
          for (int k = 0; k < 4; k++) lp += emit(prd[3-k], pbk[3-k], pbq[3-k]);
          xr -= 4; xb -= 4;
        }
        for (; t + 4 <= lmax; t += 4){
          asm volatile("" : "+v"(xr), "+v"(xb));
          const uint8_t* prd = L.rd + xr; const double2* pbq = L.bq + xr; const uint8_t* pbk = L.blk + xb;
#pragma unroll
          for (int k = 0; k < 4; k++){
            const double e = emit(prd[3-k], pbk[3-k], pbq[3-k]);
            if (t + k < len) lp += e;
          }
          xr -= 4; xb -= 4;
        }
        for (; t < lmax; t++){
          const double e = emit(L.rd[xr + 3], L.blk[xb + 3], L.bq[xr + 3]);
          if (t < len) lp += e;
          xr--; xb--;
        }
        if (valid) L.nd[dst] = lp;
      };
      // A periodic block that extends the previous allele's block by exactly one repeat unit (str_order bit 29), on a side at least as
      // long as the largest deletion: the sum of (size q, column j) starts from the same prior -ln(B - |D| + 1) and adds the same
      // emissions in the same order as the previous allele's (size q-1, column j), so the rows move up one size and only the `period`
      // columns each size gains — and size 0 — are summed: nv p pairs instead of nv (nv + 1) p / 2 (nv = c.nd sizes fit the block).
      const int nv = c.nd;
      const bool nd_reuse = (MODE == 0) && chained && ((oe >> 29) & 1) && (n >= nv*p);
      if (nd_reuse){
        auto row_off = [&](int q){ return p*((q*(q+1)) >> 1); };           // size q holds (q+1)p columns, sizes back to back
        const int ncopy = row_off(nv - 1);                                // rows 0..nv-2 -> rows 1..nv-1, read completely before the first write
        double tmp[3]; int dst[3];
#pragma unroll
        for (int rnd = 0; rnd < 3; rnd++){
          const int e = rnd*64 + lane;
          int qn = 1;
#pragma unroll
          for (int k = 1; k <= 4; k++) qn += (e >= row_off(k)) ? 1 : 0;       // destination row: row_off(qn-1) <= e < row_off(qn)
          const int idx = e - row_off(qn - 1);
          dst[rnd] = row_off(qn) + p + idx;
          tmp[rnd] = (e < ncopy) ? L.nd[row_off(qn - 1) + idx] : 0.0;
        }
        wave_lds_sync();
#pragma unroll
        for (int rnd = 0; rnd < 3; rnd++) if (rnd*64 + lane < ncopy) L.nd[dst[rnd]] = tmp[rnd];
        // the new pairs: the p lowest columns of every size
        const int e = min(lane, nv*p - 1); const bool valid = lane < nv*p;
        int q = 0;
#pragma unroll
        for (int k = 1; k <= 5; k++) q += (e >= k*p) ? 1 : 0;
        const int off = e - q*p;
        nd_round(q, (n - (q+1)*p) + off, valid, row_off(q) + off);
      } else {
        // size q has min(|D|, n) such columns; the pairs of all sizes are numbered back to back and nd[] is indexed by that number
        int cnt[HS_MAXREP], npairs = 0;
#pragma unroll
        for (int q = 0; q < HS_MAXREP; q++){ cnt[q] = (B - (q+1)*p >= 0) ? min((q+1)*p, n) : 0; npairs += cnt[q]; }
        for (int base = 0; base < npairs; base += 64){
          // surplus lanes of the last round repeat its last pair (computed, not stored), so that every lane has a real length
          int off = min(base + lane, npairs - 1); const bool valid = base + lane < npairs;
          int q = 0;
#pragma unroll
          for (int qq = 0; qq < HS_MAXREP - 1; qq++) if (q == qq && off >= cnt[qq]){ off -= cnt[qq]; q = qq + 1; }
          nd_round(q, max(0, n - (q+1)*p) + off, valid, base + lane);
        }
      }
    }
    wave_lds_sync();

    for (int kk = 0; kk < ncyc; kk++){
      const int jraw = lane + 64*kk;
      const bool actj = jraw < n;
      const int j = min(jraw, n-1);
      const int jmax = min(n-1, 64*kk + 63);        // largest column of this chunk: bounds are monotone in j
      // The 13 artifact terms are produced by ONE runtime loop (no artifact, insertions +p..+6p, deletions -p..-6p) and
      // kept in a rotating register window; fast_log_sum_exp (mathops.cpp:97-106) does not depend on their order.
      double terms[HS_NART];
      auto finish_chunk = [&](){                     // fast_log_sum_exp over the 13 artifact terms (mathops.cpp:97-106)
        Lse acc;
        for (int pass = 0; pass < 2; pass++){
          acc.start(pass, terms[0]);
#pragma unroll
          for (int t = 0; t < HS_NART; t++) acc.push(pass, terms[t], d.log_thresh);
        }
        if (actj) mr_out[j] = acc.finish();
      };
      if constexpr (MODE == 0){
        // every list is simple and tabulated: S = (lp0 + A[e]) + G[e], e from the lane's bound; a lane whose |lp0| is not below
        // Bnd[e] sends the chunk through the long form below (rare: a float rounding boundary within reach of lp0's rounding error)
        double lp0_max = 0.0;                        // largest |lp0| of the lane's 12 evaluations, against the smallest Bnd of the table
        auto tab_eval = [&](double lp0, int lim, int k) -> double {
          const int e = rdlane(tbase, k) + min(lim, 1) + max(lim - rdlane(shapes, k), 0);
          const double A = L.tab[e], G = L.tab[HS_TAB_CAP + e];
          lp0_max = fmax(lp0_max, fabs(lp0));
          return (lp0 + A) + G;
        };
        {
          const int len = min(B, j + 1);
          const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
          terms[HS_MAXREP] = (rdlane(c.cst, HS_MAXREP) + L.Mt[j]) + pre;
        }
        // ins_probs_: the first nd_eq repeat units come from the tables (layout.h nd_eq), a loop continues from there.  The usual case —
        // all six units from the tables — gets its own copy of the six terms, free of the loop's pointers and merges
        auto ins_term = [&](int q, double li){
          const int D = (q+1)*p;
          const int len = min(B + D, j + 1);
          const double lp0 = (rdlane(c.cst, 13) + li) + ((len > D) ? L.Mt[max(j - D, 0)] : 0.0);
          const int lim = actj ? min(max(0, len - D), B) : 0;
          const double S = tab_eval(lp0, lim, HS_MAXREP);
          const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
          return (rdlane(c.cst, HS_MAXREP + 1 + q) + S) + pre;
        };
        if (nd_eq == HS_MAXREP){
#pragma unroll
          for (int q = 0; q < HS_MAXREP; q++)
            terms[HS_MAXREP + 1 + q] = ins_term(q, (j >= (q+1)*p - 1) ? L.Dl[q*L.ld + j] : L.Mt[j]);
        } else {
          double li = 0.0;
          const double2* pli_bq = L.bq + (j - nd_eq*p); const uint8_t* pli_rd = L.rd + (j - nd_eq*p); int li_left = j - nd_eq*p;
#pragma unroll
          for (int q = 0; q < HS_MAXREP; q++){
            if (q < nd_eq){
              li = (j >= (q+1)*p - 1) ? L.Dl[q*L.ld + j] : L.Mt[j];
            } else {
              for (int m = 0; m < p; m++){           // read position j - t: unclamped (a step with t > j is masked; Dl sits in front of bq)
                const double2 bq = *pli_bq;
                const double e = (m < B) ? emit(*pli_rd, blk_at(c, B-1-min(m, B-1)), bq) : bq.x;
                if (li_left >= 0) li += e;
                pli_bq--; pli_rd--; li_left--;
              }
            }
            terms[HS_MAXREP + 1 + q] = ins_term(q, li);
          }
        }
        int ndo = 0;                                   // number of the first (size q, column) pair: sizes 0..q-1 come first
#pragma unroll
        for (int q = 0; q < HS_MAXREP; q++){
          const int aD = (q+1)*p;
          terms[HS_MAXREP - 1 - q] = IMP;
          if (B - aD >= 0){
            const int cq = min(aD, n);
            const int len = min(B - aD, j + 1);
            const bool direct = (j + aD <= n - 1);
            // both candidates are read (clamped indices) and one is selected: cheaper than two exec-masked branches per size
            const int jd = min(j + aD, n-1);
            const double dsum = L.Mt[jd] - L.Dl[q*L.ld + jd];
            const double ndv = L.nd[ndo + min(max(j - (n - cq), 0), cq - 1)];
            const double lp0 = direct ? rdlane(c.cst, 14 + q) + dsum : ndv;
            const double S = tab_eval(lp0, actj ? len : 0, q);
            const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
            terms[HS_MAXREP - 1 - q] = (rdlane(c.cst, HS_MAXREP - 1 - q) + S) + pre;
            ndo += cq;
          }
        }
        bool bad = !(lp0_max < tab_bmin);
        if (d.debug_redo > 0) bad |= ((ai*31 + i*7 + kk) % d.debug_redo) == 0;       // tests: exercise the re-do path
        if (!__any(bad && actj)){ finish_chunk(); continue; }
        // leave the chunk to hs_str_kernel_generic
        if (actj) mr_out[j] = HS_REDO;
        if (lane == 0) d.redo[ai] = 1;
        continue;
      }
      // what only the long forms need is loaded where they run, so that it does not occupy registers across the tabulated path
      // (per chunk instead of per allele: a handful of L1-resident loads next to >1000 instructions of evaluation)
      int lofs = 0, llen = 0; hs_visit_t bundle; double pwA = 0.0, pwB = 0.0;
      bundle.meta = 0; bundle.logU = 0.0;
      if constexpr (MODE != 0){
        lofs = (lane < HS_MAXREP) ? c.so->del_off[lane] - ins_off : 0;
        llen = (lane < HS_MAXREP) ? c.so->del_len[lane] : 0;
        if (!all_closed) bundle = ins_list[min(lane, max(total, ins_len) - 1)];
        if (!all_simple && __any((lane <= HS_MAXREP) && (shapes == HS_SHAPE_PIECEWISE))){      // descriptor slots of the piecewise-simple lists (present only where a shape says so: prep.cpp)
          pwA = d.f64pool[so_f64_off + 20 + lane];
          pwB = d.f64pool[so_f64_off + 20 + 64 + min(lane, (HS_MAXREP + 1)*HS_PW_SLOTS - 65)];
        }
      }
      if (all_simple){
        // every visiting list of this STR option is "simple" (periodic block): closed-form evaluators, statically indexed terms
        {
          const int len = min(B, j + 1);
          const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
          terms[HS_MAXREP] = (rdlane(c.cst, HS_MAXREP) + L.Mt[j]) + pre;
        }
        double li = 0.0;
        const double2* pli_bq = L.bq + (j - nd_eq*p); const uint8_t* pli_rd = L.rd + (j - nd_eq*p); int li_left = j - nd_eq*p;
#pragma unroll
        for (int q = 0; q < HS_MAXREP; q++){
          const int D = (q+1)*p;
          if (q < nd_eq) li = (j >= D - 1) ? L.Dl[q*L.ld + j] : L.Mt[j];       // layout.h nd_eq
          else
          for (int m = 0; m < p; m++){               // read position j - t: unclamped (a step with t > j is masked; Dl sits in front of bq)
            const double2 bq = *pli_bq;
            const double e = (m < B) ? emit(*pli_rd, blk_at(c, B-1-min(m, B-1)), bq) : bq.x;
            if (li_left >= 0) li += e;
            pli_bq--; pli_rd--; li_left--;
          }
          const int len = min(B + D, j + 1);
          const double lp0 = (rdlane(c.cst, 13) + li) + ((len > D) ? L.Mt[max(j - D, 0)] : 0.0);
          const int lim = actj ? min(max(0, len - D), B) : 0;
          const double S = simple_eval(d, L, lp0, lim, rdlane(shapes, HS_MAXREP), B);
          const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
          terms[HS_MAXREP + 1 + q] = (rdlane(c.cst, HS_MAXREP + 1 + q) + S) + pre;
        }
        int ndo = 0;                                   // number of the first (size q, column) pair: sizes 0..q-1 come first
#pragma unroll
        for (int q = 0; q < HS_MAXREP; q++){
          const int aD = (q+1)*p;
          terms[HS_MAXREP - 1 - q] = IMP;
          if (B - aD >= 0){
            const int cq = min(aD, n);
            const int len = min(B - aD, j + 1);
            const bool direct = (j + aD <= n - 1);
            double lp0 = rdlane(c.cst, 14 + q);
            if (direct) lp0 += L.Mt[min(j + aD, n-1)] - L.Dl[q*L.ld + min(j + aD, n-1)];
            else        lp0 = L.nd[ndo + min(max(j - (n - cq), 0), cq - 1)];
            const double S = simple_eval(d, L, lp0, actj ? len : 0, rdlane(shapes, q), B - aD);
            const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
            terms[HS_MAXREP - 1 - q] = (rdlane(c.cst, HS_MAXREP - 1 - q) + S) + pre;
            ndo += cq;
          }
        }
      } else if (MODE != 0 && all_closed){
        // simple and piecewise-simple lists only (some allele of the locus has an interrupted repeat): the same unrolled evaluation,
        // each list taking the closed form its shape names
        {
          const int len = min(B, j + 1);
          const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
          terms[HS_MAXREP] = (rdlane(c.cst, HS_MAXREP) + L.Mt[j]) + pre;
        }
        double li = 0.0;
        const double2* pli_bq = L.bq + (j - nd_eq*p); const uint8_t* pli_rd = L.rd + (j - nd_eq*p); int li_left = j - nd_eq*p;
#pragma unroll
        for (int q = 0; q < HS_MAXREP; q++){
          const int D = (q+1)*p;
          if (q < nd_eq) li = (j >= D - 1) ? L.Dl[q*L.ld + j] : L.Mt[j];       // layout.h nd_eq
          else
          for (int m = 0; m < p; m++){               // read position j - t: unclamped (a step with t > j is masked; Dl sits in front of bq)
            const double2 bq = *pli_bq;
            const double e = (m < B) ? emit(*pli_rd, blk_at(c, B-1-min(m, B-1)), bq) : bq.x;
            if (li_left >= 0) li += e;
            pli_bq--; pli_rd--; li_left--;
          }
          const int len = min(B + D, j + 1);
          const double lp0 = (rdlane(c.cst, 13) + li) + ((len > D) ? L.Mt[max(j - D, 0)] : 0.0);
          const int lim = actj ? min(max(0, len - D), B) : 0;
          const int shp = rdlane(shapes, HS_MAXREP);
          const double S = (shp >= 0) ? simple_eval(d, L, lp0, lim, shp, B) : pw_eval(d, L, j, lp0, lim, pwA, pwB, HS_MAXREP, q+1, p, B);
          const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
          terms[HS_MAXREP + 1 + q] = (rdlane(c.cst, HS_MAXREP + 1 + q) + S) + pre;
        }
        int ndo = 0;                                   // number of the first (size q, column) pair: sizes 0..q-1 come first
#pragma unroll
        for (int q = 0; q < HS_MAXREP; q++){
          const int aD = (q+1)*p;
          terms[HS_MAXREP - 1 - q] = IMP;
          if (B - aD >= 0){
            const int cq = min(aD, n);
            const int len = min(B - aD, j + 1);
            const bool direct = (j + aD <= n - 1);
            double lp0 = rdlane(c.cst, 14 + q);
            if (direct) lp0 += L.Mt[min(j + aD, n-1)] - L.Dl[q*L.ld + min(j + aD, n-1)];
            else        lp0 = L.nd[ndo + min(max(j - (n - cq), 0), cq - 1)];
            const int shp = rdlane(shapes, q);
            const double S = (shp >= 0) ? simple_eval(d, L, lp0, actj ? len : 0, shp, B - aD) : pw_eval(d, L, j, lp0, actj ? len : 0, pwA, pwB, q, 1, 0, B - aD);
            const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
            terms[HS_MAXREP - 1 - q] = (rdlane(c.cst, HS_MAXREP - 1 - q) + S) + pre;
            ndo += cq;
          }
        }
      } else {
#pragma unroll
      for (int t = 0; t < HS_NART; t++) terms[t] = IMP;
      double li = 0.0;                               // running ins_probs_ sum (StutterAlignerClass.cpp:40-51)
      int ndo = 0;                                   // number of the first (size q, column) pair in nd[]
      for (int itn = 0; itn < HS_NART; itn++){
        double term = IMP;
        if (itn == 0){                               // no artifact (StutterAlignerClass.cpp:55-57)
          const int len = min(B, j + 1);
          const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
          term = (rdlane(c.cst, HS_MAXREP) + L.Mt[j]) + pre;
        } else if (itn <= HS_MAXREP){                // insertion of D = (q+1) p
          const int q = itn - 1, D = (q+1)*p;
          if (q < nd_eq) li = (j >= D - 1) ? L.Dl[q*L.ld + j] : L.Mt[j];       // layout.h nd_eq
          else
          for (int m = 0; m < p; m++){               // extend the insertion table by one repeat unit
            const int t = q*p + m;
            const int pos = max(j - t, 0);
            const double2 bq = L.bq[pos];
            const double e = (m < B) ? emit(L.rd[pos], blk_at(c, B-1-min(m, B-1)), bq) : bq.x;
            if (t <= j) li += e;
          }
          const int len = min(B + D, j + 1);
          const double lp0 = (rdlane(c.cst, 13) + li) + ((len > D) ? L.Mt[max(j - D, 0)] : 0.0);
          const int lim = actj ? min(max(0, len - D), B) : 0;
          const int limmax = min(max(0, min(B + D, jmax + 1) - D), B);
          const int shape = rdlane(shapes, HS_MAXREP);
          const double S = (shape >= 0) ? simple_eval(d, L, lp0, lim, shape, B)
                         : (shape == HS_SHAPE_PIECEWISE) ? pw_eval(d, L, j, lp0, lim, pwA, pwB, HS_MAXREP, q+1, p, B)
                                        : visit_eval(d, L, j, lp0, lim, limmax, bundle, 0, ins_list, ins_len, q+1, p, B);
          const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
          term = (rdlane(c.cst, HS_MAXREP + 1 + q) + S) + pre;
        } else {                                     // deletion of aD = (q+1) p bases
          const int q = itn - 1 - HS_MAXREP, aD = (q+1)*p;
          if (B - aD >= 0){
            const int cq = min(aD, n);
            const int len = min(B - aD, j + 1);
            const bool direct = (j + aD <= n - 1);
            double lp0 = rdlane(c.cst, 14 + q);
            if (direct) lp0 += L.Mt[min(j + aD, n-1)] - L.Dl[q*L.ld + min(j + aD, n-1)];
            else        lp0 = L.nd[ndo + min(max(j - (n - cq), 0), cq - 1)];
            const int lim = actj ? len : 0;
            const int limmax = min(B - aD, jmax + 1);
            const int rel = rdlane(lofs, q);
            const int shape = rdlane(shapes, q);
            const double S = (shape >= 0) ? simple_eval(d, L, lp0, lim, shape, B - aD)
                           : (shape == HS_SHAPE_PIECEWISE) ? pw_eval(d, L, j, lp0, lim, pwA, pwB, q, 1, 0, B - aD)
                                          : visit_eval(d, L, j, lp0, lim, limmax, bundle, rel, ins_list + rel, rdlane(llen, q), 1, 0, B - aD);
            const double pre = (j - len < 0) ? 0.0 : L.rowP[max(j - len, 0)];
            term = (rdlane(c.cst, HS_MAXREP - 1 - q) + S) + pre;
            ndo += cq;
          }
        }
#pragma unroll
        for (int t = 0; t + 1 < HS_NART; t++) terms[t] = terms[t+1];
        terms[HS_NART-1] = term;
      }
      }
      finish_chunk();
    }
  }
}

extern "C" __global__ void __launch_bounds__(128, HS_STR_WAVES)
hs_str_kernel(const hs_dev_t* __restrict__ dp, int active_begin, int only_long){ str_body<0>(*dp, active_begin, only_long); }

#ifndef HS_STRG_WAVES
#define HS_STRG_WAVES 3      // the long forms want registers more than wavefronts: 168 VGPRs without spills beat 128 with 160 B of them
#endif
extern "C" __global__ void __launch_bounds__(128, HS_STRG_WAVES)
hs_str_kernel_generic(const hs_dev_t* __restrict__ dp, int active_begin, int pw_grouped){ str_body<1>(*dp, active_begin, pw_grouped); }

// ------------------------------------------------------------------ the STR block of tabulated alleles, grouped form
// hs_str_kernel gives every read side its own wavefront: a 150-base read seeded in the middle has ~75 columns per side, two passes of
// 64 lanes with 11 live lanes in the second, and the read-end deletion sums of a chained allele fill 24 lanes (profiles/r02_notes.md:
// 29 % and 21 % of the phase).  Here a workgroup takes a GROUP of reads of one locus and side (prep.cpp packs them so that their
// columns fill HS_GRP_COLS) and lays their columns end to end over its lanes: lane x = column j of read g, the per-column tables of all
// reads back to back in LDS, the alleles still one after the other (they are the same for every read of the locus, in the same order).
// What was wave-uniform per read (side length, workspace rows) becomes a per-lane value; what a column reads from its neighbours
// (Mt[j - D], Dl[q][j + |D|], rowP[j - len]) stays inside its own read's stretch by the same clamps as before.  Same operations in the same
// order per column as str_body<0>: bit-identical.  Two workgroup barriers per allele separate the table phase from the evaluation;
// the allele's block, constants and closed-form table are double-buffered so that loading the next allele needs no third one.
#define HS_GRP_MAXREADS 16
#ifdef HS_GTIME       // timing experiment: cycles per section of the allele loop, printed by a few workgroups
#define HS_TICK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[k] += now_ - tprev; tprev = now_; } while (0)
#else
#define HS_TICK(k) do {} while (0)
#endif
struct GrpLds {
  double* rowP; double* Mt; double* Dl;
  double* E;            // [4][XC] emission log of every column against A, C, T, G (code = (char >> 1) & 3): one read instead of base + qualities + compare
  // two of each but nd, used alternately from allele to allele; addressed as base + parity * stride (a pointer picked from an array loses its
  // address space and every access through it becomes a flat_load)
  double* nd0; double* cstl0; double2* tab0;
  const double* ilog;      // KIND 1: [max_B + 9] int_log
  uint16_t* boff0;         // [blk_len] byte offset of the block base's plane of E (valid offsets, zeros, in front of the first one: masked steps may look there)
  int ld;
};
// nd_cap: doubles of the read-end deletion table of a group = the largest (reads x 36 period) of the batch's groups (prep.cpp)
// with_ilog: hs_str_group_kernel_pw keeps int_log(0 .. max_B + 8) in LDS as well (the tail terms of the piecewise lists)
extern "C" size_t hs_str_group_lds_bytes(int max_B, int nd_cap, int with_ilog){
  const size_t XC = HS_GRP_COLS;
  const size_t blk_len = ((size_t)max_B + 19) & ~(size_t)15;
  const size_t ilog_bytes = with_ilog ? (((size_t)max_B + 9) & ~(size_t)1)*8 : 0;
  // (+ HS_GRP_MAXREADS + 2 doubles: hs_str_group_kernel_p keeps a 0.0 in front of every read in match_probs_ as well)
  // (+ 3 HS_GRP_MAXREADS + 1 ints: that kernel keeps its per-read tables behind the carve instead of in static LDS)
  return XC*8*HS_MAXREP + 2*(XC + HS_GRP_MAXREADS + 2)*8 + XC*32 + (size_t)nd_cap*8 + 2*24*8 + 2*HS_TAB_CAP*16 + (3*blk_len + 64)*2 + (3*HS_GRP_MAXREADS + 1)*4 + 16 + ilog_bytes;
}

// KIND 0: tabulated alleles, positions [0, n_tab) (or [0, n_short)) of the side's order.  KIND 1 (hs_str_group_kernel_pw): the alleles with
// piecewise simple lists, positions [n_tab, n_pw) — interrupted repeats: the same table phase and read-end sums (their blocks are
// not periodic: nothing is inherited from allele to allele but the match / deletion tables of a block that ends with the previous one),
// every list evaluated by the closed form its shape names (table entry or pw_eval_lean).
template <int KIND>
__device__ __forceinline__ void str_group_body(const hs_dev_t& d, int item_begin, int short_only){
  constexpr int XC = HS_GRP_COLS, NT = HS_GRP_COLS;
  static_assert((NT & (NT - 1)) == 0 && NT >= 128 && 4*XC*8 < 65536, "the wavefronts' turns at the read-end sums assume a power-of-two workgroup; plane offsets are 16 bits");
  const int lane = threadIdx.x & 63, x = threadIdx.x;
  const hs_item_t* item = d.items + item_begin + blockIdx.x;
  const int side = uni(item->side), G = uni(item->slot), tp = uni(item->active);
  __shared__ int s_off[HS_GRP_MAXREADS + 1], s_n[HS_GRP_MAXREADS], s_ai[HS_GRP_MAXREADS];
  if (KIND >= 1){
    // most groups have no allele of this kind (a locus whose alleles are all periodic): gone before the tables, the barriers and the LDS fill
    const hs_locus_t* loc0 = d.loci + uni(d.reads[uni(d.active[uni(d.tpack[tp])])].locus);
    const int lo = KIND == 2 ? uni(loc0->n_pw[side]) : uni(loc0->n_tab[side]), hi = KIND == 2 ? uni(loc0->n_rp[side]) : uni(loc0->n_pw[side]);
    if (lo + (int)blockIdx.y * d.allele_chunk >= hi) return;
  }
  GrpLds L;
  {
    double* Dl = (double*)hs_lds_raw;                       // first: masked steps may touch up to B entries in front of E
    double* rowP = Dl + HS_MAXREP*XC;                       // read g's stretch is shifted by g + 1: a 0.0 sits in front of every read's first column
    double* Mt = rowP + (XC + HS_GRP_MAXREADS + 2);
    const int blk_len = (d.max_B + 19) & ~15;
    double* E = Mt + XC;
    double* ndb = E + 4*XC;
    double* cst = ndb + d.grp_nd_cap;
    const int ilog_len = KIND >= 1 ? ((d.max_B + 9) & ~1) : 0;
    double* ilogb = cst + 2*24;
    double2* tab = (double2*)(ilogb + ilog_len);
    for (int i = x; i < ilog_len; i += NT) ilogb[i] = d.int_log[i];
    L.ilog = ilogb;
    uint16_t* boffb = (uint16_t*)(tab + 2*HS_TAB_CAP) + blk_len + 64;        // zeros in front: the read-end chains fetch ahead of themselves
    for (int i = x; i < blk_len + 64; i += NT) boffb[i - blk_len - 64] = 0;
    L.rowP = rowP; L.Mt = Mt; L.Dl = Dl; L.E = E; L.ld = XC;
    L.nd0 = ndb; L.cstl0 = cst; L.tab0 = tab; L.boff0 = boffb;
  }
  if (x < G){
    const int ai = d.tpack[tp + x];
    const hs_read_t r = d.reads[d.active[ai]];
    s_ai[x] = ai; s_n[x] = side ? r.len - r.seed - 1 : r.seed;
  }
  __syncthreads();
  if (x == 0){ int o = 0; for (int g = 0; g < G; g++){ s_off[g] = o; o += s_n[g]; } s_off[G] = o; }
  __syncthreads();
  const int X = s_off[G];
  // this lane's column: read g, column j of n (lanes past the last column repeat it and write nothing)
  const bool actj = x < X;
  const bool wave_act = (x & ~63) < X;            // a wavefront without columns only helps with the read-end sums
  const int xx = min(x, X - 1);
  int g = 0;
  for (int k = 1; k < G; k++) g += (xx >= s_off[k]) ? 1 : 0;
  const int offg = s_off[g], n = s_n[g], j = xx - offg, ai = s_ai[g];
  const hs_read_t rdv = d.reads[d.active[ai]];
  const hs_locus_t* loc = d.loci + uni(rdv.locus);
  const hs_ws_t wsr = d.ws[ai];
  const int lenm1 = rdv.len - 1;
  double* const mr_base = d.ws_mr + wsr.mr + (side ? rdv.seed : 0) + j;               // + re_ord*(len-1): this column in the allele's MR row
  const int lead_stride = n + uni(loc->lead_flank[side]) + 1;
  const double* const lead_base = d.ws_lead + wsr.lead[side] + j;                     // + slot*lead_stride: rowP of this column
  {
    const int src = rdv.base_off + (side ? rdv.len - 1 - j : j);
    const uint8_t q = (uint8_t)d.quals[src];
    if (actj){
      const uint8_t r = (uint8_t)d.bases[src];
      const double qc = d.qual_correct[q], qe = d.qual_error[q];
      L.E[xx] = (r == 'A') ? qc : qe; L.E[XC + xx] = (r == 'C') ? qc : qe;
      L.E[2*XC + xx] = (r == 'T') ? qc : qe; L.E[3*XC + xx] = (r == 'G') ? qc : qe;
      if (j == 0) L.rowP[xx + g] = 0.0;
    }
  }
  const int blk_len = (d.max_B + 19) & ~15;
  const int xrp = xx + g + 1;                       // this column in rowP: rowP[xrp - len] is M of column j - len, or the 0.0 in front when len = j + 1
  const int n_tab = KIND == 2 ? uni(loc->n_rp[side]) : (KIND >= 1 ? uni(loc->n_pw[side]) : (short_only ? uni(loc->n_short[side]) : uni(loc->n_tab[side])));      // short_only: hs_str_group_kernel_p has the rest
  const int i0 = (KIND == 2 ? uni(loc->n_pw[side]) : (KIND >= 1 ? uni(loc->n_tab[side]) : 0)) + blockIdx.y * d.allele_chunk, i1 = min(n_tab, i0 + d.allele_chunk);
  if (i0 >= i1) return;                              // the same for every lane of the workgroup
  const int32_t* order = d.str_order + uni(loc->order_off[side]);
  const int jmaxw = uni(wave_max_i(j)), jminw = uni(wave_min_i(j));
  const int nh_l = s_n[min(lane, G - 1)];                       // lane h < G: columns of read h
  const int nmin_g = uni(wave_min_i(nh_l));                      // shortest side of the group
  // The alleles' records are fetched 64 at a time, one allele per lane (order entry -> allele -> STR option: three dependent loads, paid
  // once), and read lane by lane; what an allele needs beyond them (constants, block, closed-form table) is requested one allele ahead.
  // a_pk1 = lead slot (10 bits) | nd << 10 | period << 13 | nd_eq << 17 | B << 20 (B <= 1024: prep.cpp);  a_pk2 = re_ord (24 bits) | tab_len << 24
  int a_oe = 0, a_pk1 = 0, a_pk2 = 0, a_sopt = 0, a_seq = 0, a_f64 = 0, a_taboff = 0;
  auto fetch_alleles = [&](int first){
    const int k = min(first + lane, i1 - 1);
    a_oe = order[k];
    const hs_allele_t* al = d.alleles + uni(loc->hap_begin) + (a_oe & 0x1fffffff);
    a_sopt = al->str_opt[side];
    const hs_stropt_t* so = d.stropts + a_sopt;
    a_pk1 = (al->lead_slot[side] & 0x3ff) | (so->nd << 10) | (so->period << 13) | (so->nd_eq << 17) | (so->B << 20);
    a_pk2 = (al->re_ord & 0xffffff) | (so->tab_len << 24);
    a_seq = so->seq_off; a_f64 = so->f64_off; a_taboff = so->tab_off;
  };
  double nx_cst = 0.0, nx_bmin = 0.0, nx_tabA = 0.0, nx_tabG = 0.0; int nx_shapes = -1, nx_tbase = 0, nx_blkw = 0;
  auto request = [&](int k){                          // k: lane of the allele in the fetched batch
    const int f64o = rdlane(a_f64, k), sopt = rdlane(a_sopt, k), tl = (rdlane(a_pk2, k) >> 24) & 0xff, Bk = (rdlane(a_pk1, k) >> 20) & 0x7ff;
    const double* tsrc = d.f64pool + rdlane(a_taboff, k);
    nx_cst = d.f64pool[f64o + min(lane, 19)];        // lane t < 20: pmf[13] | prior_ins | prior_del[6]
    nx_shapes = (lane <= HS_MAXREP) ? d.stropts[sopt].shape[lane] : -1;
    nx_tbase = (lane <= HS_MAXREP) ? d.stropts[sopt].tab_base[lane] : 0;
    nx_bmin = tsrc[3*tl];
    nx_blkw = (x < (Bk + 3)/4) ? ((const int*)(d.chars + rdlane(a_seq, k)))[x] : 0;
    if (x < tl){ nx_tabA = tsrc[3*x]; nx_tabG = tsrc[3*x + 1]; }
  };
  if (i0 < i1){ fetch_alleles(i0); request(0); }
  int cur_slot = -1, prev_B = 0, nd_base = 0;
#ifdef HS_GTIME
  unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime(), nrounds = 0;
#endif
  for (int i = i0; i < i1; i++){
    const int par = (i - i0) & 1;
    const int k = (i - i0) & 63;
    const int oe = rdlane(a_oe, k);
    const bool chained = (i > i0) && ((oe >> 30) & 1);
    const int pk1 = rdlane(a_pk1, k), pk2 = rdlane(a_pk2, k);
    const int slot = pk1 & 0x3ff, re_ord = pk2 & 0xffffff;
    double* const mr_out = mr_base + (int64_t)re_ord*lenm1;
    const int B = (pk1 >> 20) & 0x7ff, nv = (pk1 >> 10) & 7, p = (pk1 >> 13) & 15, nd_eq = (pk1 >> 17) & 7, tab_len = (pk2 >> 24) & 0xff;
    const double* const pw_desc = d.f64pool + rdlane(a_f64, k) + 20;          // KIND 1: the option's 7 x HS_PW_SLOTS descriptor slots
    // Read-end deletion sums of one read: six row slots of 6p entries, entry = distance of the column from the read end.  Size q lives in
    // slot (nd_base + q) mod 6.  Where the block extends the previous allele's by one repeat unit, the sums of (size q, column) are the
    // previous allele's (size q-1, column): the base steps back by one, every row is the next size without moving, each gains the p
    // columns farthest from the end, and the slot of the old largest size becomes the new size 0 — nothing is copied.
    const int sixp = HS_MAXREP*p, nds = HS_MAXREP*sixp;
    const int ndb = g*nds;
    const double cst = nx_cst;
    const int shapes = nx_shapes, tbase = nx_tbase;
    const double tab_bmin = uni(nx_bmin);
    if (slot != cur_slot){          // M of the row before the STR block, from the leading-flank kernel: the previous allele's readers first
      __syncthreads();
      if (actj) L.rowP[xrp] = lead_base[(int64_t)slot*lead_stride];
      cur_slot = slot;
    }
    // this allele's block, constants and table go to the buffers the allele before the previous one used
    if (x < (B + 3)/4){
      int2 bo;                                      // A, C, T, G -> plane 0, 1, 2, 3 (prep.cpp tabulates only blocks made of these four), two 16-bit offsets per word
      bo.x = (((nx_blkw >> 1) & 3) * (XC*8)) | ((((nx_blkw >> 9) & 3) * (XC*8)) << 16);
      bo.y = (((nx_blkw >> 17) & 3) * (XC*8)) | ((((nx_blkw >> 25) & 3) * (XC*8)) << 16);
      ((int2*)(L.boff0 + par*blk_len))[x] = bo;
    }
    if (x < 20) (L.cstl0 + par*24)[x] = cst;
    if (x < tab_len) (L.tab0 + par*HS_TAB_CAP)[x] = make_double2(nx_tabA, nx_tabG);
    int pw_touch;
    if (i + 1 < i1){
      if (k == 63) fetch_alleles(i + 1);
      request((k + 1) & 63);
    }
    {
      if (KIND >= 1){
        // the next allele's descriptor slots (560 bytes, nine cache lines) on their way to the scalar cache: a miss at the point of use is a
        // trip to L2 per list with every wavefront of the workgroup waiting (one destination: the values are not used)
        const int kn = (i + 1 < i1) ? ((k + 1) & 63) : k;
        const uint64_t ta0 = (uint64_t)(uintptr_t)(d.f64pool + rdlane(a_f64, kn) + 20);
        const uint64_t ta = ((uint64_t)(uint32_t)uni((int)(ta0 >> 32)) << 32) | (uint32_t)uni((int)ta0);
        asm volatile("s_load_dword %0, %1, 0x0\n\ts_load_dword %0, %1, 0x40\n\ts_load_dword %0, %1, 0x80\n\ts_load_dword %0, %1, 0xc0\n\ts_load_dword %0, %1, 0x100\n\t"
                     "s_load_dword %0, %1, 0x140\n\ts_load_dword %0, %1, 0x180\n\ts_load_dword %0, %1, 0x1c0\n\ts_load_dword %0, %1, 0x200"
                     : "=&s"(pw_touch) : "s"(ta) : "memory");
      }
    }
    HS_TICK(0);   // phase 3 of the previous allele + setup
    __syncthreads();                // ... and every wavefront is done with the previous allele's Mt / Dl
    HS_TICK(1);   // barrier 1 wait
    const double* cstl = L.cstl0 + par*24;
    const double2* tab = L.tab0 + par*HS_TAB_CAP;
    const uint16_t* boff = L.boff0 + par*blk_len;
    auto Eat = [&](int col, int bo) -> double { return *(const double*)((const char*)L.E + col*8 + bo); };   // column col against the block base with plane offset bo
    double* nd = L.nd0;

    // --- StutterAlignerClass::load_read (StutterAlignerClass.cpp:12-53): match_probs_ and del_probs_ of this lane's column
    const int t0 = chained ? prev_B : 0;
    prev_B = B;
    const int tmax = min(B, jmaxw + 1);          // a step t > j is masked: no lane of this wavefront goes past its largest column
    if (wave_act && t0 < tmax){
      double lp = (t0 > 0) ? L.Mt[xx] : 0.0;
      const int ndp = nv * p;
      int t = t0;
      if (KIND >= 1 && p <= 6 && t % p == 0 && t + p <= min(tmax, ndp)){
        // whole repeat units first: a unit's emissions are requested together and added in order, the deletion table's row written once per
        // unit (an interrupted block rarely continues the previous allele's tables: all its B steps are taken here, one LDS round trip per unit
        // instead of one per step)
        int col = xx - t, left = j - t;
        double* dl = L.Dl + (t/p)*L.ld + xx;
        for (; t + p <= min(tmax, ndp); t += p){
          double e[6];
#pragma unroll
          for (int r = 0; r < 6; r++) if (r < p) e[r] = Eat(col - r, boff[B-1-t-r]);
#pragma unroll
          for (int r = 0; r < 6; r++) if (r < p){ if (left - r >= 0) lp += e[r]; }
          if (left - (p - 1) >= 0 && actj) *dl = lp;
          dl += L.ld; col -= p; left -= p;
        }
      }
      if (t < min(tmax, ndp)){
        int col = xx - t;
        int left = j - t, ph = (t + 1) % p;
        double* dl = L.Dl + ((t + 1)/p - 1)*L.ld + xx;
        for (; t < min(tmax, ndp); t++){
          const double e = Eat(col, boff[B-1-t]);
          if (left >= 0) lp += e;
          if (ph == 0){ if (left >= 0 && actj) *dl = lp; }
          ph++; if (ph == p){ ph = 0; dl += L.ld; }
          col--; left--;
        }
      }
      auto steps = [&](int tend, auto masked){
        int xr = xx - t - 3, xb = B - 1 - t - 3;
        for (; t + 4 <= tend; t += 4){
          asm volatile("" : "+v"(xr));
#pragma unroll
          for (int k = 0; k < 4; k++){
            const double e = Eat(xr + 3 - k, boff[xb + 3 - k]);
            if (!decltype(masked)::value || xr + 3 - k >= offg) lp += e;
          }
          xr -= 4; xb -= 4;
        }
        for (; t < tend; t++){
          const double e = Eat(xr + 3, boff[xb + 3]);
          if (!decltype(masked)::value || xr + 3 >= offg) lp += e;
          xr--; xb--;
        }
      };
      if (t < tmax){
        steps(min(tmax, jminw + 1), std::false_type());
        steps(tmax, std::true_type());
      }
      if (actj) L.Mt[xx] = lp;
    }

    HS_TICK(2);   // phase 1
    // --- deletion start values of the columns whose segment reaches the read end (the `else` branch of StutterAlignerClass.cpp:117-120),
    // (size, column) pairs of all reads of the group spread over the workgroup's lanes; layout per read as in str_body
    {
      asm volatile("" ::: "memory");
      auto nd_sum = [&](int q, int xcol, int jcol, bool valid, int dst){
        const int aD = (q+1)*p;
        const int len = min(B - aD, jcol + 1);
        const int lmin = uni(wave_min_i(len)), lmax = uni(wave_max_i(len));
        double lp = cstl[14 + q];
        // step t pairs read column xcol - t with block base B-1-aD - t, four steps per group.  The chain of additions is the critical
        // path of the allele, so its operands run ahead of it: plane offsets two groups ahead, emissions one group ahead.
        constexpr int GS = 4;                             // steps per group
        int xr = xcol - (GS - 1), xb = (B - 1 - aD) - (GS - 1), t = 0;
        int bo[GS]; double ev[GS];
#pragma unroll
        for (int k = 0; k < GS; k++) bo[k] = boff[xb + GS - 1 - k];
#pragma unroll
        for (int k = 0; k < GS; k++) ev[k] = Eat(xr + GS - 1 - k, bo[k]);
        xr -= GS; xb -= GS;
#pragma unroll
        for (int k = 0; k < GS; k++) bo[k] = boff[xb + GS - 1 - k];    // in front of the table: zeros (a valid plane), never used
        auto advance = [&](){
          asm volatile("" : "+v"(xr), "+v"(xb));
#pragma unroll
          for (int k = 0; k < GS; k++) ev[k] = Eat(xr + GS - 1 - k, bo[k]);
          xr -= GS; xb -= GS;
#pragma unroll
          for (int k = 0; k < GS; k++) bo[k] = boff[xb + GS - 1 - k];
        };
        // the chain is the allele's critical path (the other wavefronts of the workgroup wait for it at the barrier) and a handful of
        // instructions per step: it goes first whenever it is ready
#ifdef HS_GTIME
        const unsigned long long tc0 = __builtin_amdgcn_s_memtime();
#endif
        __builtin_amdgcn_s_setprio(3);
        for (; t + GS <= lmin; t += GS){
          double e[GS];
#pragma unroll
          for (int k = 0; k < GS; k++) e[k] = ev[k];
          advance();
#pragma unroll
          for (int k = 0; k < GS; k++) lp += e[k];
        }
        for (; t < lmax; t += GS){
          double e[GS];
#pragma unroll
          for (int k = 0; k < GS; k++) e[k] = ev[k];
          advance();
#pragma unroll
          for (int k = 0; k < GS; k++) if (t + k < len) lp += e[k];
        }
        __builtin_amdgcn_s_setprio(0);
#ifdef HS_GTIME
        tacc[5] += __builtin_amdgcn_s_memtime() - tc0; tacc[6] += lmax; nrounds++;
#endif
        if (valid) nd[dst] = lp;
      };
      auto row_off = [&](int q){ return p*((q*(q+1)) >> 1); };
      const bool reuse_al = chained && ((oe >> 29) & 1);
      if (reuse_al) nd_base = (nd_base + HS_MAXREP - 1) % HS_MAXREP;
      auto slot_of = [&](int q){ int sl = nd_base + q; sl -= (sl >= HS_MAXREP) ? HS_MAXREP : 0; return sl*sixp; };     // row of size q
      // Work of the group for this allele.  Usual case (every side of the group holds all six deletion sizes, so all reads count
      // alike): closed-form numbering, size-major so that the sums of a wavefront have (nearly) the same length.  Otherwise lane
      // h < G counts read h's work, a prefix sum over those lanes numbers it back to back and every lane searches the prefix.
      const bool all_long = nmin_g >= HS_MAXREP*p;
      auto udiv = [&](int e, int c, float rc) -> int {            // e / c for small non-negative e, rc ~ 1/c
        int h = (int)((float)e * rc);
        h -= (h*c > e) ? 1 : 0; h += ((h + 1)*c <= e) ? 1 : 0;
        return h;
      };
      const int xw = (x + (NT/2)*(i - i0)) & (NT - 1);               // the sums rarely fill the workgroup: the wavefronts take turns at them (a wavefront's SIMD is fixed)
      if (all_long){
        const int Gp = G*p;
        const float rc_p = __builtin_amdgcn_rcpf((float)p);        // approximate: udiv corrects by one either way
        if (reuse_al){
          const int n_sums = nv*Gp;                                // e = (q G + h) p + off
          const float rc_gp = __builtin_amdgcn_rcpf((float)Gp);
          for (int base = 0; base < n_sums; base += NT){
            if (base + (xw & ~63) >= n_sums) continue;             // whole wavefront past the end (wave-uniform)
            const int e = min(base + xw, n_sums - 1);
            const int q = udiv(e, Gp, rc_gp), r = e - q*Gp, hh = udiv(r, p, rc_p), off = r - hh*p;
            const int jcol = (s_n[hh] - (q+1)*p) + off;
            nd_sum(q, s_off[hh] + jcol, jcol, base + xw < n_sums, hh*nds + slot_of(q) + ((q+1)*p - 1 - off));
          }
        } else {
          const int n_sums = G*row_off(nv);                        // size q: G reads x (q+1)p columns, sizes back to back from G row_off(q)
          for (int base = 0; base < n_sums; base += NT){
            if (base + (xw & ~63) >= n_sums) continue;
            const int e = min(base + xw, n_sums - 1);
            int q = 0;
#pragma unroll
            for (int k = 1; k <= 5; k++) q += (e >= G*row_off(k)) ? 1 : 0;
            const int r = e - G*row_off(q), w = (q+1)*p;
            const int hh = udiv(r, w, __builtin_amdgcn_rcpf((float)w)), off = r - hh*w;
            const int jcol = (s_n[hh] - w) + off;
            nd_sum(q, s_off[hh] + jcol, jcol, base + xw < n_sums, hh*nds + slot_of(q) + (w - 1 - off));
          }
        }
      } else {
      int c_l = 0;
      {
        const bool ru = reuse_al && (nh_l >= nv*p);
        int np_l = 0;
#pragma unroll
        for (int q = 0; q < HS_MAXREP; q++) np_l += (B - (q+1)*p >= 0) ? min((q+1)*p, nh_l) : 0;
        if (lane < G) c_l = ru ? nv*p : np_l;
      }
      int pc = c_l;                                           // inclusive prefix sums over lanes 0..15
#pragma unroll
      for (int dd = 1; dd < HS_GRP_MAXREADS; dd <<= 1){
        const int t1 = __shfl_up(pc, dd);
        if (lane >= dd) pc += t1;
      }
      const int n_sums = rdlane(pc, HS_GRP_MAXREADS - 1);
      auto find = [&](int pref, int e, int& hh, int& loc_e){   // read hh holds item e: prefix(hh-1) <= e < prefix(hh)
        hh = 0; loc_e = e;
        for (int h = 0; h + 1 < G; h++){
          const int ph = rdlane(pref, h);
          if (e >= ph){ hh = h + 1; loc_e = e - ph; }
        }
      };
      for (int base = 0; base < n_sums; base += NT){
        const int wbase = base + (xw & ~63);
        if (wbase >= n_sums) continue;                        // whole wavefront past the end (wave-uniform)
        const int e = min(base + xw, n_sums - 1);
        const bool valid = base + xw < n_sums;
        int hh, loc_e; find(pc, e, hh, loc_e);
        const int nh = s_n[hh], offh = s_off[hh];
        const bool ruh = reuse_al && (nh >= nv*p);
        int q = 0, off = loc_e, jcol, dst;
        if (ruh){
#pragma unroll
          for (int k = 1; k <= 5; k++) q += (loc_e >= k*p) ? 1 : 0;
          off = loc_e - q*p;
          jcol = (nh - (q+1)*p) + off;
        } else {
          int cnt[HS_MAXREP];
#pragma unroll
          for (int qq = 0; qq < HS_MAXREP; qq++) cnt[qq] = (B - (qq+1)*p >= 0) ? min((qq+1)*p, nh) : 0;
#pragma unroll
          for (int qq = 0; qq < HS_MAXREP - 1; qq++) if (q == qq && off >= cnt[qq]){ off -= cnt[qq]; q = qq + 1; }
          jcol = max(0, nh - (q+1)*p) + off;
        }
        dst = slot_of(q) + (nh - 1 - jcol);
        nd_sum(q, offh + jcol, jcol, valid, hh*nds + dst);
      }
      }
    }
    HS_TICK(3);   // nd section
    __syncthreads();

    HS_TICK(4);   // barrier 2 wait
    if (KIND >= 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(pw_touch) :: "memory");      // (long since there; the register is free again)
    // --- the 13 artifact terms of this lane's column (HapAligner.cpp:62-109) and their fast_log_sum_exp
    if (wave_act){
      asm volatile("" ::: "memory");
      double terms[HS_NART];
      double lp0_max = 0.0;
      const int Eb = (int)(uintptr_t)(__attribute__((address_space(3))) char*)L.E;
      {
      const int k_allele = k;                                 // (the allele's lane in the fetched batch: the lambdas below use k for the list)
      auto load_pw = [&](int k) -> PwSlots {                  // the ten descriptor slots of list k into scalar registers
          PwSlots S;
          typedef int hs_i2w __attribute__((ext_vector_type(2)));
          hs_i2w q0, q1, q2, q3, q4, q5, q6, q7, q8, q9;
          const uint64_t pa = (uint64_t)(uintptr_t)(pw_desc + k*HS_PW_SLOTS);
          asm volatile("s_load_dwordx2 %0, %10, 0x0\n\ts_load_dwordx2 %1, %10, 0x8\n\ts_load_dwordx2 %2, %10, 0x10\n\ts_load_dwordx2 %3, %10, 0x18\n\t"
                       "s_load_dwordx2 %4, %10, 0x20\n\ts_load_dwordx2 %5, %10, 0x28\n\ts_load_dwordx2 %6, %10, 0x30\n\ts_load_dwordx2 %7, %10, 0x38\n\t"
                       "s_load_dwordx2 %8, %10, 0x40\n\ts_load_dwordx2 %9, %10, 0x48\n\ts_waitcnt lgkmcnt(0)"
                       : "=&s"(q0), "=&s"(q1), "=&s"(q2), "=&s"(q3), "=&s"(q4), "=&s"(q5), "=&s"(q6), "=&s"(q7), "=&s"(q8), "=&s"(q9) : "s"(pa) : "memory");
          S.v[0] = q0.x; S.v[1] = q0.y; S.v[2] = q1.x; S.v[3] = q1.y; S.v[4] = q2.x; S.v[5] = q2.y; S.v[6] = q3.x; S.v[7] = q3.y; S.v[8] = q4.x; S.v[9] = q4.y;
          S.v[10] = q5.x; S.v[11] = q5.y; S.v[12] = q6.x; S.v[13] = q6.y; S.v[14] = q7.x; S.v[15] = q7.y; S.v[16] = q8.x; S.v[17] = q8.y; S.v[18] = q9.x; S.v[19] = q9.y;
          return S;
      };
      auto tab_eval = [&](double lp0, int lim, int k, int nsub, int stride, int tail, const PwSlots* pre) -> double {
        const int shp = rdlane(shapes, k);
        if (KIND >= 1 && shp == HS_SHAPE_PIECEWISE){          // (the same for every lane)
          if (pre) return pw_eval_lean<XC>(*pre, L.ilog, d.log_thresh, Eb + 8*xx, lp0, lim, nsub, 8*stride, tail);
          const PwSlots S = load_pw(k);
          return pw_eval_lean<XC>(S, L.ilog, d.log_thresh, Eb + 8*xx, lp0, lim, nsub, 8*stride, tail);
        }
        if (KIND == 2 && shp == HS_SHAPE_PWK){                // three to six breaks: the K-level closed form (the same for every lane)
          return pwk_eval_lean<XC>(pw_desc + (HS_MAXREP + 1)*HS_PW_SLOTS + k*HS_PWK_SLOTS, L.ilog, d.log_thresh, Eb + 8*xx, lp0, lim, nsub, 8*stride, tail);
        }
        if (KIND == 2 && shp == -1){                          // more: the list itself, replayed (the same for every lane)
          const hs_stropt_t* so = d.stropts + rdlane(a_sopt, k_allele);
          const int loff = uni(k == HS_MAXREP ? so->ins_off : so->del_off[min(k, HS_MAXREP - 1)]);
          const int llen = uni(k == HS_MAXREP ? so->ins_len : so->del_len[min(k, HS_MAXREP - 1)]);
          return visit_eval_grp<XC>((const hs_visit_t*)d.visits + loff, llen, L.ilog, d.log_thresh, Eb, xx, lp0, lim, tail, nsub, stride, tail);
        }
        const int e = rdlane(tbase, k) + min(lim, 1) + max(lim - shp, 0);
        const double2 ag = tab[e];
        lp0_max = fmax(lp0_max, fabs(lp0));
        return (lp0 + ag.x) + ag.y;
      };
      {
        const int len = min(B, j + 1);
        const double pre = L.rowP[xrp - len];
        terms[HS_MAXREP] = (rdlane(cst, HS_MAXREP) + L.Mt[xx]) + pre;
      }
      PwSlots Sins;                                           // the insertion list serves all six sizes: its slots are fetched once
      if (KIND >= 1) Sins = load_pw(HS_MAXREP);                 // (a kind-2 option has the slots of all seven lists: "not piecewise" where the list is simple)
      else { for (int t = 0; t < 2*HS_PW_SLOTS; t++) Sins.v[t] = 0; }
      auto ins_term = [&](int q, double li){
        const int D = (q+1)*p;
        const int len = min(B + D, j + 1);
        const double lp0 = (rdlane(cst, 13) + li) + ((len > D) ? L.Mt[xx - min(D, j)] : 0.0);
        const int lim = min(max(0, len - D), B);            // a lane past the group's last column repeats it: its bound is a real one
        const double S = tab_eval(lp0, lim, HS_MAXREP, q + 1, p, B, &Sins);
        const double pre = L.rowP[xrp - len];
        return (rdlane(cst, HS_MAXREP + 1 + q) + S) + pre;
      };
      if (nd_eq == HS_MAXREP){
#pragma unroll
        for (int q = 0; q < HS_MAXREP; q++)
          terms[HS_MAXREP + 1 + q] = ins_term(q, (j >= (q+1)*p - 1) ? L.Dl[q*L.ld + xx] : L.Mt[xx]);
      } else {
        double li = 0.0;
        int li_col = xx - nd_eq*p, li_left = j - nd_eq*p;
#pragma unroll
        for (int q = 0; q < HS_MAXREP; q++){
          if (q < nd_eq){
            li = (j >= (q+1)*p - 1) ? L.Dl[q*L.ld + xx] : L.Mt[xx];
          } else {
            for (int m = 0; m < p; m++){           // m < period <= B for a tabulated block (prep.cpp)
              const double e = Eat(li_col, boff[B-1-m]);
              if (li_left >= 0) li += e;
              li_col--; li_left--;
            }
          }
          terms[HS_MAXREP + 1 + q] = ins_term(q, li);
        }
      }
#pragma unroll
      for (int q = 0; q < HS_MAXREP; q++){
        const int aD = (q+1)*p;
        terms[HS_MAXREP - 1 - q] = IMP;
        if (B - aD >= 0){
          const int cq = min(aD, n);
          const int len = min(B - aD, j + 1);
          const bool direct = (j + aD <= n - 1);
          const int xd = xx + min(aD, n - 1 - j);
          const double dsum = L.Mt[xd] - L.Dl[q*L.ld + xd];
          int slq = nd_base + q; slq -= (slq >= HS_MAXREP) ? HS_MAXREP : 0;
          const double ndv = nd[ndb + slq*sixp + min(n - 1 - j, cq - 1)];
          const double lp0 = direct ? rdlane(cst, 14 + q) + dsum : ndv;
          const double S = tab_eval(lp0, len, q, 1, 0, B - aD, (const PwSlots*)0);
          const double pre = L.rowP[xrp - len];
          terms[HS_MAXREP - 1 - q] = (rdlane(cst, HS_MAXREP - 1 - q) + S) + pre;
        }
      }
      }
      bool bad = !(lp0_max < tab_bmin);
      if (d.debug_redo > 0) bad |= ((ai*31 + i*7 + (j >> 6)) % d.debug_redo) == 0;       // tests: exercise the re-do path
      if (!__any(bad && actj)){
        // fast_log_sum_exp of the 13 terms (mathops.cpp:97-106): two terms per step, the float exponential's bits as in LeanAcc
        double mx13 = terms[0];
#pragma unroll
        for (int t = 1; t < HS_NART; t++) mx13 = fmax(mx13, terms[t]);
        LeanAcc acc; acc.log_thresh = d.log_thresh; acc.tot = 0.0; acc.mx = mx13;
#pragma unroll
        for (int t = 0; t + 1 < HS_NART; t += 2) acc.pair(terms[t], true, terms[t + 1], 1.0, false);
        acc.one(terms[HS_NART - 1]);
        if (actj) mr_out[0] = mx13 + (double)f_fasterlog((float)acc.tot);
      } else if (actj){                              // leave these columns to hs_str_kernel_generic
        mr_out[0] = HS_REDO;
        d.redo[ai] = 1;
      }
    }
  }
#ifdef HS_GTIME
  { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[0] += now_ - tprev; }
  if ((blockIdx.x % 20000) == 7 && lane == 0)
    printf("grp %d wave %d G %d X %d alleles %d: ph3+setup %llu  wait1 %llu  ph1 %llu  nd %llu  wait2 %llu | chains %llu in %llu rounds, %llu steps\n", (int)blockIdx.x, (int)(x >> 6), G, X, i1 - i0,
           tacc[0], tacc[1], tacc[2], tacc[3], tacc[4], tacc[5], nrounds, tacc[6]);
#endif
}

#ifndef HS_GRP_OCC
#define HS_GRP_OCC HS_STR_WAVES      // wavefronts per SIMD the register allocation aims at
#endif
extern "C" __global__ void __launch_bounds__(HS_GRP_COLS, HS_GRP_OCC)
hs_str_group_kernel(const hs_dev_t* __restrict__ dp, int item_begin, int short_only){ str_group_body<0>(*dp, item_begin, short_only); }
#ifndef HS_GRP_PW_OCC
#define HS_GRP_PW_OCC 4      // wavefronts per SIMD the register allocation aims at (measured: 3 with 141 registers is 17 % slower than 4 with 128 and four spilled)
#endif
extern "C" __global__ void __launch_bounds__(HS_GRP_COLS, HS_GRP_PW_OCC)
hs_str_group_kernel_pw(const hs_dev_t* __restrict__ dp, int item_begin){ str_group_body<1>(*dp, item_begin, 0); }
// ... and the alleles with a list that has to be replayed (three and more interruptions: hs_stropt_t::kind 3), positions [n_pw, n_rp): the same
// body with visit_eval_grp for those lists; registers before wavefronts (the replay loop sits inside the 13-term evaluation)
#ifndef HS_GRP_RP_OCC
#define HS_GRP_RP_OCC 4      // (measured with the K-level closed form: 4 with 128 registers and 128 B of scratch is 16 % faster than 3 with 158)
#endif
extern "C" __global__ void __launch_bounds__(HS_GRP_COLS, HS_GRP_RP_OCC)
hs_str_group_kernel_rp(const hs_dev_t* __restrict__ dp, int item_begin){ str_group_body<2>(*dp, item_begin, 0); }

// ------------------------------------------------------------------ the STR block of tabulated alleles, grouped form, period known at compile time
// hs_str_group_kernel_p<P> takes the tabulated alleles whose blocks hold at least six repeat units of period P (positions
// [n_short, n_tab) of a side's order: nearly all of them) — the same lane layout, LDS tables, allele loop and operations per column as
// str_group_body, with what a compile-time period and a periodic block allow:
//   * the 13 artifact terms are evaluated branch-free with every allele-independent quantity of a lane — its LDS addresses into the match /
//     deletion tables for the six insertion and six deletion sizes, its distance from the read end — computed once before the allele loop,
//     the table offsets of the sizes as immediates, and the per-allele constants as scalars: 10-11 integer operations per term instead of ~25
//     plus six v_readlane;
//   * the insertion table is the deletion table: with six repeat units in the block (nd_eq = 6) ins_probs_[q] of a column is del_probs_[q]
//     or, for a column closer than (q+1)P to the read start, the truncated sum — which the table phase now stores in that row as well;
//   * a read-end deletion sum pairs read column xcol - t with block base B-1-|D| - t, and in a periodic block that base is base (t mod P)
//     from the right end whatever the allele and the size: the plane of the emission table is a scalar per step (sp[t mod P], the loop
//     unrolled over a period), the addresses of a chain are arithmetic, and its emissions are requested three groups ahead of the additions
//     instead of waiting for a plane-offset lookup per group (the chain is the allele's critical path: the other wavefronts wait for it).
// Same values added in the same order: bit-identical to str_group_body (tools/fuzz_align.py).
// LDS of one hs_str_group_kernel_p workgroup: deletion table | rowP | match_probs_ | emission table | closed-form table x 2 | per-read ints
#define HS_GRP_P_LDS_BYTES ((HS_MAXREP*HS_GRP_COLS + 2*(HS_GRP_COLS + HS_GRP_MAXREADS + 2) + 4*HS_GRP_COLS + 4*HS_TAB_CAP)*8 + (3*HS_GRP_MAXREADS + 1)*4 + 12)
extern "C" size_t hs_str_group_p_lds_bytes(){ return HS_GRP_P_LDS_BYTES; }

// ------------------------------------------------------------------ read-end deletion sums of the tabulated alleles, all of a locus side at once
// A deletion whose segment reaches past the read end cannot take its start value from the match / deletion tables: it is the position
// prior plus the emissions of the read's last bases against the block remainder, a strictly sequential sum (the `else` branch of
// StutterAlignerClass.cpp:117-120).  It depends on the remainder's length B - |D| and the column only, so an allele whose block is the
// previous one's plus a repeat unit inherits all but one row of sums (hs_ndrow_t, layout.h).  Inside the allele loop of the group kernel
// these sums were its critical path — 50 to 150 dependent additions on a quarter of a workgroup's lanes while the other wavefronts waited
// at a barrier.  Here every (row, column) pair of a read side is a lane of its own, rows of near-equal length next to each other, and
// the latency is hidden by the other workgroups; the group kernel fetches its six values per column (hs_ws_t::nd).
// The entries of a read side's read-end rows, period P: entry e = (row, distance of the column from the read end).  Step t of an entry's sum
// pairs column j - t with block base (t mod P) from the right end, so the P LDS addresses of a repeat unit move together: one address
// per base of the unit, stepped once per group of units, the group's emissions requested together and added in the reference's order
// (the additions are the same chain as before: bit-identical; the loop was one LDS round trip per step).
template <int P>
__device__ __forceinline__ void nd_entries(const hs_dev_t& d, const hs_ndrow_t* __restrict__ rows, double* __restrict__ out, int total, int sixp, int n, int tid, int e_lds){
  constexpr int M = (P == 1) ? 6 : (P == 2) ? 3 : (P == 3) ? 2 : 1;       // repeat units per group: four to six steps
  constexpr int GS = M*P;
  auto ldb = [](int byte_addr) -> double { return *(const __attribute__((address_space(3))) double*)(uintptr_t)(uint32_t)byte_addr; };
  for (int e = tid; e < total; e += 256){
    const int r = e / sixp, off = e - r*sixp;
    const hs_ndrow_t rw = rows[r];
    const int j = n - 1 - off;
    if (rw.len < 0 || j < 0) continue;                  // no allele has this size / the read side has no such column
    const int len = min(rw.len, j + 1);
    double lp = -d.int_log[rw.len + 1];                 // the position prior (StutterAlignerClass.cpp:112)
    int a[P];                                           // byte address of column j - t - k against the block base k from the right end
#pragma unroll
    for (int k = 0; k < P; k++) a[k] = e_lds + (((rw.tail_codes >> (2*k)) & 3)*HS_MAX_SIDE_LEN + j - k)*8;
    int t = 0;
    for (; t + GS <= len; t += GS){
      double ev[GS];
      // the group's lowest column is >= 0, so a[k] - (GS-P)*8 is an LDS address: masked to say so, the units' distances become the
      // instruction's (unsigned) offset field instead of an addition each
#pragma unroll
      for (int k = 0; k < P; k++){
        const int lo = (a[k] - (GS - P)*8) & 0x3ffff;
#pragma unroll
        for (int m = 0; m < M; m++) ev[m*P + k] = ldb(lo + (GS - P - m*P)*8);
      }
#pragma unroll
      for (int k = 0; k < P; k++) a[k] -= GS*8;
#pragma unroll
      for (int q = 0; q < GS; q++) lp += ev[q];
    }
    for (int k = 0; t < len; t++, k = (k + 1 == P) ? 0 : k + 1){      // the last, partial group
#pragma unroll
      for (int kk = 0; kk < P; kk++) if (kk == k){ lp += ldb(a[kk]); a[kk] -= P*8; }
    }
    out[e] = lp;
  }
}

extern "C" __global__ void __launch_bounds__(256)
hs_nd_kernel(const hs_dev_t* __restrict__ dp, int active_begin){
  const hs_dev_t& d = *dp;
  const int ai = active_begin + blockIdx.x, side = blockIdx.y, tid = threadIdx.x;
  const hs_read_t rd = d.reads[d.active[ai]];
  const hs_locus_t* loc = d.loci + uni(rd.locus);
  const int nrows = uni(loc->n_ndrows[side]);
  if (nrows == 0) return;
  const int p = uni(loc->period), sixp = HS_MAXREP*p;
  const int n = uni(side ? rd.len - rd.seed - 1 : rd.seed);
  if (n > HS_MAX_SIDE_LEN) return;                      // not in a group: hs_str_kernel sums for itself
  __shared__ double E[4][HS_MAX_SIDE_LEN];              // emission log of every column against A, C, T, G (code = (char >> 1) & 3)
  for (int c = tid; c < n; c += 256){
    const int src = rd.base_off + (side ? rd.len - 1 - c : c);
    const uint8_t q = (uint8_t)d.quals[src], r = (uint8_t)d.bases[src];
    const double qc = d.qual_correct[q], qe = d.qual_error[q];
    E[0][c] = (r == 'A') ? qc : qe; E[1][c] = (r == 'C') ? qc : qe; E[2][c] = (r == 'T') ? qc : qe; E[3][c] = (r == 'G') ? qc : qe;
  }
  __syncthreads();
  const hs_ndrow_t* rows = d.nd_rows + uni(loc->ndrow_off[side]);
  double* out = d.ws_nd + uni(side ? d.ws[ai].nd[1] : d.ws[ai].nd[0]);
  const int total = nrows*sixp;
  const int e_lds = (int)(uintptr_t)(__attribute__((address_space(3))) char*)&E[0][0];
  switch (p){
    case 1: nd_entries<1>(d, rows, out, total, sixp, n, tid, e_lds); break;
    case 2: nd_entries<2>(d, rows, out, total, sixp, n, tid, e_lds); break;
    case 3: nd_entries<3>(d, rows, out, total, sixp, n, tid, e_lds); break;
    case 4: nd_entries<4>(d, rows, out, total, sixp, n, tid, e_lds); break;
    case 5: nd_entries<5>(d, rows, out, total, sixp, n, tid, e_lds); break;
    case 6: nd_entries<6>(d, rows, out, total, sixp, n, tid, e_lds); break;
    default: break;                                  // HS_MAXREP periods only (prep.cpp: longer periods have no read-end rows)
  }
}

#ifndef HS_GTIME_EVERY
#define HS_GTIME_EVERY 20000
#endif
template <int P>
__device__ __forceinline__ void str_group_body_p(const hs_dev_t& d, int item_begin){
  constexpr int XC = HS_GRP_COLS, NT = HS_GRP_COLS;
  constexpr int SIXP = HS_MAXREP*P, NDS = HS_MAXREP*SIXP;            // a row slot / the six row slots of one read's read-end sums
  const int lane = threadIdx.x & 63, x = threadIdx.x;
#ifdef HS_GTIME
  const unsigned long long t_entry = __builtin_amdgcn_s_memtime();
#endif
  const hs_item_t* item = d.items + item_begin + blockIdx.x;
  const int side = uni(item->side), G = uni(item->slot), tp = uni(item->active);
  // LDS carve of str_group_body (same size function), addressed as offsets in doubles from the start.  No static LDS in this kernel: the
  // dynamic block starts at address 0 and the offsets below are the addresses (the compiler adds the base to every access otherwise)
  // (match_probs_ is indexed like rowP here: a 0.0 in front of every read's first column)
  constexpr int oDl = 0, oRowP = HS_MAXREP*XC, oMt = oRowP + (XC + HS_GRP_MAXREADS + 2), oE = oMt + (XC + HS_GRP_MAXREADS + 2);
  constexpr int oTab = oE + 4*XC, oInts = oTab + 4*HS_TAB_CAP;        // tab: per parity A[HS_TAB_CAP] | G[HS_TAB_CAP]
  static_assert(oInts*8 + (3*HS_GRP_MAXREADS + 1)*4 <= HS_GRP_P_LDS_BYTES, "LDS carve of hs_str_group_kernel_p");
  double* const lds = (double*)hs_lds_raw;
  int* const s_off = (int*)(lds + oInts);                             // [HS_GRP_MAXREADS + 1] first column of every read | [..] columns | [..] active-read index
  int* const s_n = s_off + (HS_GRP_MAXREADS + 1);
  int* const s_ai = s_n + HS_GRP_MAXREADS;
  // LDS reads by byte ADDRESS: lds0 (the address of the carve: 0, there is no static LDS in this kernel — but it is not assumed) is part of
  // every per-lane and per-allele base below, so that an access is one ds_read with an immediate offset and no addition of the base
  const int lds0 = (int)(uintptr_t)(__attribute__((address_space(3))) char*)hs_lds_raw;
  auto ldb = [&](int byte_addr) -> double { return *(const __attribute__((address_space(3))) double*)(uintptr_t)(uint32_t)byte_addr; };
  if (x < G){
    const int ai = d.tpack[tp + x];
    const hs_read_t r = d.reads[d.active[ai]];
    s_ai[x] = ai; s_n[x] = side ? r.len - r.seed - 1 : r.seed;
  }
  __syncthreads();
  if (x == 0){ int o = 0; for (int g = 0; g < G; g++){ s_off[g] = o; o += s_n[g]; } s_off[G] = o; }
  __syncthreads();
  const int X = s_off[G];
  const bool actj = x < X;
  const bool wave_act = (x & ~63) < X;
  const int xx = min(x, X - 1);
  int g = 0;
  for (int k = 1; k < G; k++) g += (xx >= s_off[k]) ? 1 : 0;
  const int offg = s_off[g], n = s_n[g], j = xx - offg, ai = s_ai[g];
  const hs_read_t rdv = d.reads[d.active[ai]];
  const hs_locus_t* loc = d.loci + uni(rdv.locus);
  const int n_tab = uni(loc->n_tab[side]);
  const int i0 = uni(uni(loc->n_short[side]) + (int)blockIdx.y * d.allele_chunk), i1 = uni(min(n_tab, i0 + d.allele_chunk));
  if (i0 >= i1) return;                              // the same for every lane of the workgroup
  const int lenm1 = rdv.len - 1;
  double* const mr_base = d.ws_mr + d.ws[ai].mr + (side ? rdv.seed : 0) + j;          // (field by field: a local copy of the record, indexed by `side`, would live in scratch memory)
  const int lead_stride = n + uni(loc->lead_flank[side]) + 1;
  const double* const lead_base = d.ws_lead + (side ? d.ws[ai].lead[1] : d.ws[ai].lead[0]) + j;
  {
    const int src = rdv.base_off + (side ? rdv.len - 1 - j : j);
    const uint8_t q = (uint8_t)d.quals[src];
    if (actj){
      const uint8_t r = (uint8_t)d.bases[src];
      const double qc = d.qual_correct[q], qe = d.qual_error[q];
      lds[oE + xx] = (r == 'A') ? qc : qe; lds[oE + XC + xx] = (r == 'C') ? qc : qe;
      lds[oE + 2*XC + xx] = (r == 'T') ? qc : qe; lds[oE + 3*XC + xx] = (r == 'G') ? qc : qe;
      if (j == 0){ lds[oRowP + xx + g] = 0.0; lds[oMt + xx + g] = 0.0; }
    }
  }
  const int xrp = xx + g + 1;
  const int32_t* order = d.str_order + uni(loc->order_off[side]);
  const int jmaxw = uni(wave_max_i(j)), jminw = uni(wave_min_i(j));
  const int nh_l = s_n[min(lane, G - 1)];
  const int nmin_g = uni(wave_min_i(nh_l));
  // ---- what a lane needs of its column for the 13 terms, whatever the allele (byte offsets into the LDS carve)
  int rj = n - 1 - j;                                  // distance from the read end
  int j8p8 = 8*(j + 1);
  int aM = lds0 + 8*xrp;                               // + 8 oRowP - 8 len: M of column j - len before the block, or the 0.0 in front of the read;  + 8 oMt: match_probs_ of this column
  int aCol = lds0 + 8*xx;                              // + 8 oDl + q XC 8: del_probs_[q] of this column
  int aZ = lds0 + 8*(offg + g);                        // the 0.0 in front of this lane's read (rowP and match_probs_ alike)
  // this column's read-end deletion sums (hs_nd_kernel): row r of the side's block + the column's distance from the read end
  const double* const ndp = d.ws_nd + (side ? d.ws[ai].nd[1] : d.ws[ai].nd[0]) + min(rj, SIXP - 1);
  const bool nd_lane = rj < SIXP;                      // (a column farther than the largest deletion from the read end takes every start value from the tables)
  // ---- what an allele needs that is the same for every lane comes from its record (layout.h HS_GRP_REC_DWORDS) by scalar loads: the header of
  // the NEXT allele while this one is evaluated (its table and block are requested one allele ahead), the constants when they are needed
  typedef int hs_i16v __attribute__((ext_vector_type(16)));
  typedef int hs_i8v_ __attribute__((ext_vector_type(8)));
  typedef int hs_i4v_ __attribute__((ext_vector_type(4)));
  typedef int hs_i2v_ __attribute__((ext_vector_type(2)));
  const int32_t* const recs = d.grp_recs + (int64_t)uni(loc->rec_off[side])*HS_GRP_REC_DWORDS;
  auto rec_addr = [&](int i){ return (uint64_t)(uintptr_t)(recs + (int64_t)i*HS_GRP_REC_DWORDS); };
  // (the scalar loads below are inline assembly, several per block: their destinations are early-clobber operands, or the first load's
  // destination may be given the registers that hold the address the following loads still need)
  // The header (four dwords) of the allele after the next is requested while this one is worked on, together with one dword of each
  // of the record's other three cache lines: by the time the constants are wanted they sit in the scalar cache (a miss costs a trip to
  // L2 at each of the three places that wait for them, with all four wavefronts of the workgroup waiting at the same time).
  auto load_header = [&](int i) -> hs_i4v_ {
    hs_i4v_ hd; const uint64_t ra = rec_addr(i);
    asm volatile("s_load_dwordx4 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)" : "=&s"(hd) : "s"(ra) : "memory");
    return hd;
  };
  double nx_tab = 0.0;                                // lanes 0.. of wavefront 0 the table's A, of wavefront 1 its G
  auto request = [&](const hs_i4v_& hd){
    const int tl = (hd[0] >> 10) & 0xff;
    const double* tsrc = d.f64pool + hd[3];
    if (lane < tl && x < 128) nx_tab = tsrc[3*lane + (x >> 6)];
  };
  auto load_row = [&](int i) -> int {
    int r; const uint64_t ra = rec_addr(i);
    asm volatile("s_load_dword %0, %1, 0x10\n\ts_waitcnt lgkmcnt(0)" : "=&s"(r) : "s"(ra) : "memory");
    return r;
  };
  hs_i4v_ hd_cur = load_header(i0), hd_nx1 = load_header(min(i0 + 1, i1 - 1));
  int row_cur = load_row(i0), row_nx1 = load_row(min(i0 + 1, i1 - 1));
  request(hd_cur);
  int cur_slot = -1, prev_B = 0;
#ifdef HS_GTIME
  unsigned long long tacc[7] = {0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime(); const unsigned long long t_loop0 = tprev;
#endif
  for (int i = i0; i < i1; i++){
    const int par = (i - i0) & 1;
    const hs_i4v_ hd = hd_cur;
    const bool chained = (i > i0) && ((hd[0] >> 30) & 1);
    const int slot = hd[0] & 0x3ff, re_ord = hd[1];
    double* const mr_out = mr_base + (int64_t)re_ord*lenm1;
    const int B = hd[2] >> 12, tab_len = (hd[0] >> 10) & 0xff;
    const int tail = hd[2] & 0xfff;
    if (slot != cur_slot){
      __syncthreads();
      if (actj) lds[oRowP + xrp] = lead_base[(int64_t)slot*lead_stride];
      cur_slot = slot;
    }
    if (lane < tab_len && x < 128) lds[oTab + par*2*HS_TAB_CAP + (x >> 6)*HS_TAB_CAP + lane] = nx_tab;
    if (i + 1 < i1) request(hd_nx1);
    hs_i4v_ hd_nx2; int row_nx2, touch1, touch2, touch3;
    { const uint64_t ra2 = rec_addr(min(i + 2, i1 - 1));
      asm volatile("s_load_dwordx4 %0, %5, 0x0\n\ts_load_dword %1, %5, 0x10\n\ts_load_dword %2, %5, 0x40\n\ts_load_dword %3, %5, 0x80\n\ts_load_dword %4, %5, 0xc0"
                   : "=&s"(hd_nx2), "=&s"(row_nx2), "=&s"(touch1), "=&s"(touch2), "=&s"(touch3) : "s"(ra2) : "memory"); }
    HS_TICK(0);   // evaluation of the previous allele + setup
    __syncthreads();
    HS_TICK(1);   // barrier 1 wait
    // plane (byte offset into the emission table) of the block base r = t mod P from the right end — the block is periodic, so that is
    // base t from the right end: a scalar, extracted from the record's tail codes where it is used
    auto spk = [&](int r) -> int { return ((tail >> (2*r)) & 3) * (XC*8); };

    // --- StutterAlignerClass::load_read (StutterAlignerClass.cpp:12-53): match_probs_ and del_probs_ of this lane's column.  Row q of the
    // deletion table also takes the truncated sum of a column closer than (q+1)P to the read start (the running value no longer changes
    // there): that is ins_probs_[q] of the column (StutterAlignerClass.cpp:40-51), and the deletions only read rows of columns >= (q+1)P
    // Where a chain of alleles starts (t0 = 0) the six rows are built in 6P steps whatever the block length, the block base cycling through
    // the block's last P bases (that is how ins_probs_ continues past a block of fewer than six units, and in a periodic block it is the
    // base itself), match_probs_ being the running value after min(B, j + 1) steps; a block that continues the previous one only adds steps.
    const int t0 = chained ? prev_B : 0;
    prev_B = B;
    const int nv = min(HS_MAXREP, B / P);                   // deletion sizes the block holds (num_deletions_, StutterAlignerClass.h:64-69)
    const int tmax = min(B, jmaxw + 1);
    if (wave_act && (t0 == 0 || t0 < tmax)){
      double lp = (t0 > 0) ? lds[oMt + xrp] : 0.0;
      int t = t0;
      if (t0 == 0){
        double lpB = 0.0;
        int cb = lds0 + 8*(oE + xx - (P - 1)), left = j, dl = oDl + xx;
#pragma unroll 1
        for (int u = 0; u < HS_MAXREP; u++){
          double e[P];
#pragma unroll
          for (int r = 0; r < P; r++) e[r] = ldb(cb + spk(r) + 8*(P - 1 - r));
#pragma unroll
          for (int r = 0; r < P; r++){
            if (left - r >= 0) lp += e[r];
            if (u*P + r + 1 == B) lpB = lp;                 // (wave-uniform) the block ends here: match_probs_
          }
          cb -= 8*P; left -= P;
          if (actj) lds[dl] = lp;
          dl += XC;
        }
        if (B <= SIXP) lp = lpB;
        t = (B > SIXP) ? SIXP : B;                          // B > 6P: the sum goes on from step 6P;  else it is complete
      }
      if (t < tmax){
        // steps t .. tmax-1 in units of P: step t + r pairs with base (t + r) mod P from the right end = code r of the tail rotated by t mod P
        const int rem = t % P;
        const int t2 = tail & ((1 << (2*P)) - 1);
        const int tailR = (t2 | (t2 << (2*P))) >> (2*rem);
        auto spR = [&](int r) -> int { return ((tailR >> (2*r)) & 3) * (XC*8); };
        int cb = lds0 + 8*(oE + xx - t - (P - 1)), left = j - t;
        const int t_all = min(tmax, jminw + 1);             // up to here every lane of the wavefront has the column
        for (; t + P <= t_all; t += P){
          double e[P];
#pragma unroll
          for (int r = 0; r < P; r++) e[r] = ldb(cb + spR(r) + 8*(P - 1 - r));
#pragma unroll
          for (int r = 0; r < P; r++) lp += e[r];
          cb -= 8*P; left -= P;
        }
        for (; t < tmax; t += P){
          double e[P];
#pragma unroll
          for (int r = 0; r < P; r++) e[r] = ldb(cb + spR(r) + 8*(P - 1 - r));
#pragma unroll
          for (int r = 0; r < P; r++) if (left - r >= 0 && t + r < tmax) lp += e[r];
          cb -= 8*P; left -= P;
        }
      }
      if (actj) lds[oMt + xrp] = lp;
    }

    HS_TICK(2);   // table phase
    __syncthreads();
    HS_TICK(4);   // barrier 2 wait

    // --- the 13 artifact terms of this lane's column (HapAligner.cpp:62-109) and their fast_log_sum_exp
    if (wave_act){
      double terms[HS_NART];
      double lp0_max = 0.0;
      const int B8 = 8*B;
      const int aTab = uni(lds0 + 8*(oTab + par*2*HS_TAB_CAP));
      // the column's read-end deletion sums for the six sizes, requested now (from L2: hs_nd_kernel wrote them) and used after the insertions
      double ndv[HS_MAXREP];
#pragma unroll
      for (int q = 0; q < HS_MAXREP; q++) ndv[q] = 0.0;
      if (nd_lane){
        const double* pr = ndp + (int64_t)row_cur*SIXP;
#pragma unroll
        for (int q = 0; q < HS_MAXREP; q++) ndv[q] = pr[-q*SIXP];       // size q: row (size-0 row) - q
      }
      // the lane's column constants, opaque from here on: otherwise every address below that does not depend on the allele is computed once
      // in front of the allele loop and kept — in more registers than there are (they went to scratch memory)
      asm volatile("" : "+v"(aM), "+v"(aCol), "+v"(aZ), "+v"(j8p8), "+v"(rj));
      // (lp0 + A[e]) + G[e], e = tab_base + [bound > 0] + max(bound - U0, 0), everything in bytes
      // the allele's constants, in two stages of scalar loads (the scalar registers do not hold all of them next to everything else): the
      // lists' table bases and shapes (record dwords 8..15), the table's smallest Bnd (56..57), pmf[6..12] | prior_ins (28..43) for the
      // unchanged size and the insertions; then pmf[0..5] (16..27) | prior_del[6] (44..55) for the deletions
      hs_i16v cA; hs_i8v_ cS; hs_i2v_ cD;
      const uint64_t ra = rec_addr(i);
      asm volatile("s_load_dwordx16 %0, %3, 0x70\n\ts_load_dwordx8 %1, %3, 0x20\n\ts_load_dwordx2 %2, %3, 0xe0\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(cA), "=&s"(cS), "=&s"(cD) : "s"(ra) : "memory");
      auto pmf_hi = [&](int t) -> double { return __hiloint2double(cA[2*(t - 6) + 1], cA[2*(t - 6)]); };      // t = 6..13, a constant once unrolled
      const double tab_bmin = __hiloint2double(cD[1], cD[0]);
      auto tab_eval = [&](double lp0, int lim8, int k) -> double {
        const int stk = cS[k];
        const int tb8 = aTab + 8*(stk >> 16), u8 = 8*(stk & 0xffff);
        const int e8 = (min(lim8, 8) + max(lim8 - u8, 0)) + tb8;
        const double A = ldb(e8), Gv = ldb(e8 + 8*HS_TAB_CAP);
        lp0_max = fmax(lp0_max, fabs(lp0));
        return (lp0 + A) + Gv;
      };
      {
        const int len8 = min(B8, j8p8);
        terms[HS_MAXREP] = (pmf_hi(HS_MAXREP) + ldb(8*oMt + aM)) + ldb(8*oRowP + aM - len8);
      }
      const double prior_ins = pmf_hi(13);
#pragma unroll
      for (int q = 0; q < HS_MAXREP; q++){                 // insertion of D = (q+1) P (StutterAlignerClass.cpp:59-104)
        const int D8 = 8*(q+1)*P;
        const int len8 = min(B8 + D8, j8p8);
        const double li = ldb(8*oDl + 8*q*XC + aCol);      // ins_probs_[q] of this column (table phase above)
        // match_probs_ of column j - D if the segment is longer than the insertion (StutterAlignerClass.cpp:66), else the 0.0 in front of the read
        const double lp0 = (prior_ins + li) + ldb(8*oMt + max(aM - D8, aZ));
        const int lim8 = min(max(j8p8 - D8, 0), B8);       // min(max(0, len - D), B)
        const double S = tab_eval(lp0, lim8, HS_MAXREP);
        terms[HS_MAXREP + 1 + q] = (pmf_hi(HS_MAXREP + 1 + q) + S) + ldb(8*oRowP + aM - len8);
        __builtin_amdgcn_sched_barrier(0);                // one term at a time: the scheduler otherwise requests every table value of the 13 terms up front, in more registers than there are
      }
      hs_i8v_ cE, cP; hs_i4v_ cF, cQ;
      asm volatile("s_load_dwordx8 %0, %4, 0x40\n\ts_load_dwordx4 %1, %4, 0x60\n\ts_load_dwordx8 %2, %4, 0xb0\n\ts_load_dwordx4 %3, %4, 0xd0\n\ts_waitcnt lgkmcnt(0)"
                   : "=&s"(cP), "=&s"(cQ), "=&s"(cE), "=&s"(cF) : "s"(ra) : "memory");
      auto pmf_lo = [&](int t) -> double { return t < 4 ? __hiloint2double(cP[2*t + 1], cP[2*t]) : __hiloint2double(cQ[2*(t - 4) + 1], cQ[2*(t - 4)]); };   // t = 0..5
      auto pd_at = [&](int q) -> double { return q < 4 ? __hiloint2double(cE[2*q + 1], cE[2*q]) : __hiloint2double(cF[2*(q - 4) + 1], cF[2*(q - 4)]); };
#pragma unroll
      for (int q = 0; q < HS_MAXREP; q++){                 // deletion of aD = (q+1) P (StutterAlignerClass.cpp:106-150)
        const int aD = (q+1)*P;
        if (q >= nv){ terms[HS_MAXREP - 1 - q] = IMP; continue; }          // (wave-uniform) the block is shorter than this deletion (HapAligner.cpp:75-77)
        const int len8 = min(B8 - 8*aD, j8p8);
        // column j + |D|: past the read end the values are another read's (or nothing's) and not used
        const double dsum = ldb(8*oMt + 8*aD + aM) - ldb(8*oDl + 8*q*XC + 8*aD + aCol);
        const double dv = pd_at(q) + dsum;
        const double lp0 = (rj >= aD) ? dv : ndv[q];       // the segment ends inside the read: from the tables; else the read-end sum
        const double S = tab_eval(lp0, len8, q);
        terms[HS_MAXREP - 1 - q] = (pmf_lo(HS_MAXREP - 1 - q) + S) + ldb(8*oRowP + aM - len8);
        __builtin_amdgcn_sched_barrier(0);
      }
      bool bad = !(lp0_max < tab_bmin);
      if (d.debug_redo > 0) bad |= ((ai*31 + i*7 + (j >> 6)) % d.debug_redo) == 0;
      if (!__any(bad && actj)){
        double mx = terms[0];
#pragma unroll
        for (int t = 1; t < HS_NART; t++) mx = fmax(mx, terms[t]);
        // fast_log_sum_exp (mathops.cpp:97-106), branch-free.  A term that passes the threshold has 1.44 dd > -10: fasterexp's clamp at -126
        // (fastonebigheader.h:210) cannot act on it and is left out; a term that does not contributes 0.0, whatever its bits would have been
        // (two terms per float operation: packed multiplies and adds, each rounded on its own like the scalar ones)
        LeanAcc acc; acc.log_thresh = d.log_thresh; acc.tot = 0.0; acc.mx = mx;
#pragma unroll
        for (int t = 0; t + 1 < HS_NART; t += 2) acc.pair(terms[t], true, terms[t + 1], 1.0, false);
        acc.one(terms[HS_NART - 1]);
        const double tot = acc.tot;
        if (actj) mr_out[0] = mx + (double)f_fasterlog((float)tot);
      } else if (actj){
        mr_out[0] = HS_REDO;
        d.redo[ai] = 1;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(hd_nx2), "+s"(row_nx2), "+s"(touch1), "+s"(touch2), "+s"(touch3) :: "memory");      // (long since there)
    hd_cur = hd_nx1; hd_nx1 = hd_nx2; row_cur = row_nx1; row_nx1 = row_nx2;
  }
#ifdef HS_GTIME
  { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tacc[0] += now_ - tprev; }
  if ((blockIdx.x % HS_GTIME_EVERY) == 7 && lane == 0)
    printf("grp_p<%d> %d wave %d G %d X %d alleles %d: prologue %llu  eval+setup %llu  wait1 %llu  table %llu  nd %llu  wait2 %llu\n", P, (int)blockIdx.x, (int)(x >> 6), G, X, i1 - i0,
           t_loop0 - t_entry, tacc[0], tacc[1], tacc[2], tacc[3], tacc[4]);
#endif
}

extern "C" __global__ void __launch_bounds__(HS_GRP_COLS, HS_GRP_OCC)
hs_str_group_kernel_p(const hs_dev_t* __restrict__ dp, int item_begin){
  const hs_dev_t& d = *dp;
  // the period of the group's locus (every allele of a locus has the same): read through the first tabulated allele of the side
  const hs_item_t* item = d.items + item_begin + blockIdx.x;
  const int side = uni(item->side);
  const hs_locus_t* loc = d.loci + uni(d.reads[d.active[d.tpack[uni(item->active)]]].locus);
  if (uni(loc->n_short[side]) >= uni(loc->n_tab[side])) return;
  const int oe = uni(d.str_order[uni(loc->order_off[side]) + uni(loc->n_short[side])]);
  const int p = uni(d.stropts[d.alleles[uni(loc->hap_begin) + (oe & 0x1fffffff)].str_opt[side]].period);
  switch (p){
    case 1: str_group_body_p<1>(d, item_begin); break;
    case 2: str_group_body_p<2>(d, item_begin); break;
    case 3: str_group_body_p<3>(d, item_begin); break;
    case 4: str_group_body_p<4>(d, item_begin); break;
    case 5: str_group_body_p<5>(d, item_begin); break;
    case 6: str_group_body_p<6>(d, item_begin); break;
    default: break;                                  // prep.cpp: n_short == n_tab for longer periods
  }
}


// compute_aln_logprob (HapAligner.cpp:163-231): log-sum-exp over the haplotype positions the seed base can sit on.
//
// A term of the sum is ((prior + e(y)) + a) + b (HapAligner.cpp:182-229) where exactly one of a, b comes from the allele's own hand-off
// rows (the STR block's row in MR, the trailing rows in LT) and the other one from the read's leading-flank record, which the alleles of a
// locus share.  What does not depend on the allele — prior, emission of the seed base, the shared addend, the ADDRESS of the allele's
// value for workspace row 0 and its stride per row — is therefore set up once per (read, flank configuration) in registers, lane = seed
// position (up to HS_CMB_ROUNDS x 64 positions); an allele then costs one address, one load and two additions per position:
//     term = (X + v) + Y      X = prior + e, Y = shared addend        where the allele's value is the first addend
//                             X = (prior + e) + shared, Y = -0.0      where it is the second (x + -0.0 == x for every x)
// Four alleles go through the maximum and the sum together: two lane swaps (v_permlane32_swap, v_permlane16_swap) leave every row of 16
// lanes with one allele's partial results, so that the row-wise DPP steps reduce four alleles at once and lanes 15/31/47/63 finish and
// store one allele each.  Haplotypes with an empty flank or more than HS_CMB_ROUNDS x 64 flank bases take the per-allele form below.
#ifndef HS_CMB_WAVES
#define HS_CMB_WAVES 1        // wavefronts per active read: a wavefront sets its registers up once and takes every HS_CMB_WAVES-th quad of alleles
                              // (NS: 1 -> 3.55 ms, 2 -> 3.76, 4 -> 4.86; the kernel reads 15 GB of LT per pass: ~4 TB/s)
#endif
#ifndef HS_CMB_ROUNDS
#define HS_CMB_ROUNDS 4
#endif

// swap of 32-lane halves / of 16-lane rows between two registers, for doubles (gfx950: v_permlane32_swap, v_permlane16_swap)
//   halves: a' = [a.lo32 | b.lo32], b' = [a.hi32 | b.hi32]        rows: a' = [a.r0 b.r0 a.r2 b.r2], b' = [a.r1 b.r1 a.r3 b.r3]
__device__ __forceinline__ void swap_halves(double& a, double& b){
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap_rows(double& a, double& b){
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  a = __hiloint2double((int)hi[0], (int)lo[0]); b = __hiloint2double((int)hi[1], (int)lo[1]);
}
// four per-lane values, one per allele of a quad -> one register whose 16-lane rows hold the alleles [0, 2, 1, 3], reduced row-wise:
// lane 15 of a row ends with the row's maximum / sum
__device__ __forceinline__ double quad_max(double a0, double a1, double a2, double a3){
  swap_halves(a0, a1); swap_halves(a2, a3);
  double m01 = fmax(a0, a1), m23 = fmax(a2, a3);          // halves: [allele 0 | allele 1], [allele 2 | allele 3]
  swap_rows(m01, m23);
  double v = fmax(m01, m23);                              // rows: alleles 0, 2, 1, 3
  v = fmax(v, dpp_d<0x111, 0xf>(v, v)); v = fmax(v, dpp_d<0x112, 0xf>(v, v)); v = fmax(v, dpp_d<0x114, 0xf>(v, v)); v = fmax(v, dpp_d<0x118, 0xf>(v, v));
  return v;
}
__device__ __forceinline__ double quad_sum(double a0, double a1, double a2, double a3){
  swap_halves(a0, a1); swap_halves(a2, a3);
  double m01 = a0 + a1, m23 = a2 + a3;
  swap_rows(m01, m23);
  double v = m01 + m23;
  v += dpp_d<0x111, 0xf>(v, 0.0); v += dpp_d<0x112, 0xf>(v, 0.0); v += dpp_d<0x114, 0xf>(v, 0.0); v += dpp_d<0x118, 0xf>(v, 0.0);
  return v;
}

// the per-position registers of one flank configuration
template <int NR> struct CmbRegs { double X[NR], Y[NR]; const char* P[NR]; int S[NR]; };

// one quad of alleles (workspace rows ord[0..3]; a slot that is not wanted repeats a wanted one's row and is not stored)
template <int NR>
__device__ __forceinline__ void combine_quad(const hs_dev_t& d, const CmbRegs<NR>& g, const int (&ord)[4], int want, double* out_row, int k0, int lane){
  double t[4][NR], m[4];
#pragma unroll
  for (int q = 0; q < 4; q++){
#pragma unroll
    for (int k = 0; k < NR; k++){
      const double v = *(const double*)(g.P[k] + (uint64_t)(uint32_t)ord[q]*(uint64_t)(uint32_t)g.S[k]);     // 64-bit product: 2^24 haplotypes x a row of KBs passes 4 GiB
      t[q][k] = (g.X[k] + v) + g.Y[k];
    }
    m[q] = t[q][0];
#pragma unroll
    for (int k = 1; k < NR; k++) m[q] = fmax(m[q], t[q][k]);
    m[q] = fmax(m[q], -1.0e300);                          // fast_log_sum_exp starts its maximum there (mathops.cpp:97-106 through Lse)
  }
  const double mrow = quad_max(m[0], m[1], m[2], m[3]);   // lane 15 of row r: the maximum of allele [0, 2, 1, 3][r]
  const double mx[4] = { rdlane(mrow, 15), rdlane(mrow, 47), rdlane(mrow, 31), rdlane(mrow, 63) };
  double s[4];
#pragma unroll
  for (int q = 0; q < 4; q++){
    s[q] = 0.0;
#pragma unroll
    for (int k = 0; k < NR; k++){
      const double df = t[q][k] - mx[q];
      const double e = (double)f_fasterexp((float)df);
      s[q] += (df > d.log_thresh) ? e : 0.0;
    }
  }
  const double tot = quad_sum(s[0], s[1], s[2], s[3]);
  const double res = mrow + (double)f_fasterlog((float)tot);
  const int row = lane >> 4, slot = ((row & 1) << 1) | (row >> 1);
  if ((lane & 15) == 15 && ((want >> slot) & 1)) out_row[k0 + slot] = res;
}

// the per-allele form: one wavefront, lanes = seed positions, every operand selected per position (any flank lengths)
__device__ __forceinline__ void combine_one(const hs_dev_t& d, const SideView& vL, const SideView& vR, int lane, uint8_t seed_c, double seed_lc, double seed_lw,
                                            int N, int ord, int lead_off, int F0, int trail_off, int slotL, int slotR, double* out){
  const int F2 = N - F0;
  const double* recL = lead_record(d, vL, slotL);
  const double* recR = lead_record(d, vR, slotR);
  const double* mr = d.ws_mr + vL.ws_mr + (int64_t)ord*(vL.len-1);
  const double* lt = d.ws_lt + vL.ws_lt + (int64_t)ord*uni(vL.loc->lt_stride);
  // last-column value of compact row u of a side: leading rows 0..Flead-1, the STR block's row at Flead, trailing rows after it
  auto lcL = [&](int u){ const double* a = (u < F0) ? recL + (vL.n + u) : ((u == F0) ? mr + (vL.nL - 1) : lt + (u - F0 - 1)); return *a; };
  auto lcR = [&](int u){ const double* a = (u < F2) ? recR + (vR.n + u) : ((u == F2) ? mr + (vL.len - 2) : lt + (F2 + (u - F2 - 1))); return *a; };
  const double sideL = recL[vL.n + uni(vL.loc->lead_flank[0])], sideR = recR[vR.n + uni(vL.loc->lead_flank[1])];
  const double prior = -d.int_log[N];
  auto term = [&](int y){
    const uint8_t hc = (uint8_t)((y < F0 ? d.rows[lead_off + y] : d.rows[trail_off + y - F0]) & 0xff);
    const double e = (seed_c == hc) ? seed_lc : seed_lw;
    // y == 0: the whole left side hangs off the haplotype; y == N-1: the right side does (HapAligner.cpp:182-189)
    const int uL = (y == N-1) ? N-1 : ((y < F0) ? max(y-1, 0) : y);
    const int uR = (y == 0) ? N-1 : ((y < F0) ? N-1-y : max(N-2-y, 0));
    const double vl = lcL(uL), vr = lcR(uR);
    const double a = (y == 0) ? sideL : ((y == N-1) ? sideR : vl);
    const double b = (y == N-1) ? vl : vr;
    return ((prior + e) + a) + b;
  };
  Lse acc;
  for (int pass = 0; pass < 2; pass++){
    if (pass == 0) acc.mx = -1.0e300; else acc.tot = 0.0;
    for (int y = lane; y < N; y += 64) acc.push(pass, term(y), d.log_thresh);
    if (pass == 0) acc.mx = wave_max_d(acc.mx); else acc.tot = wave_sum_d(acc.tot);
  }
  if (lane == 0) *out = acc.finish();
}

// registers of a configuration: position y = lane + 64 k.  Requires F0 >= 1, F2 >= 1 (so that exactly one addend is the allele's).
template <int NR>
__device__ __forceinline__ void combine_setup(const hs_dev_t& d, const SideView& vL, const SideView& vR, int lane, uint8_t seed_c, double seed_lc, double seed_lw,
                                              int N, int lead_off, int F0, int trail_off, int slotL, int slotR, CmbRegs<NR>& g){
  const int F2 = N - F0;
  const double* recL = lead_record(d, vL, slotL);
  const double* recR = lead_record(d, vR, slotR);
  const char* mr0 = (const char*)(d.ws_mr + vL.ws_mr);
  const char* lt0 = (const char*)(d.ws_lt + vL.ws_lt);
  const int mr_stride = (vL.len - 1)*8, lt_stride = uni(vL.loc->lt_stride)*8;
  const double sideL = recL[vL.n + uni(vL.loc->lead_flank[0])], sideR = recR[vR.n + uni(vL.loc->lead_flank[1])];
  const double prior = -d.int_log[N];
#pragma unroll
  for (int k = 0; k < NR; k++){
    const int y = lane + 64*k;
    const bool live = y < N;
    const int yy = live ? y : 0;                                        // a dead lane computes position 0 and is overwritten below
    const uint8_t hc = (uint8_t)((yy < F0 ? d.rows[lead_off + yy] : d.rows[trail_off + yy - F0]) & 0xff);
    const double pe = prior + ((seed_c == hc) ? seed_lc : seed_lw);
    // which compact row of which side the allele's value is: the right side's row uR for positions in the left flank (and the first
    // position), the left side's row uL for positions in the right flank (and the last one)
    const bool left_part = yy < F0 && yy != N-1;
    const int u = left_part ? N-1-yy : yy;                              // right side's row : left side's row
    const int fl = left_part ? F2 : F0;                                 // the side's leading-flank length: row fl is the STR block's
    const int mr_col = left_part ? vL.len - 2 : vL.nL - 1;
    const int lt_col = left_part ? u - 1 : u - F0 - 1;                  // lcR: lt[F2 + (u-F2-1)], lcL: lt[u-F0-1]
    const bool in_mr = (u == fl);
    g.P[k] = (in_mr ? mr0 : lt0) + (int64_t)(in_mr ? mr_col : lt_col)*8;
    g.S[k] = in_mr ? mr_stride : lt_stride;
    // the shared addend and its place
    const double sh = (yy == 0) ? sideL : ((yy == N-1) ? sideR : (left_part ? recL[vL.n + yy - 1] : recR[vR.n + (N-2-yy)]));
    const bool shared_first = left_part || yy == N-1;                   // a is shared: ((prior+e) + shared) + allele's
    g.X[k] = live ? (shared_first ? pe + sh : pe) : -1.0e300;
    g.Y[k] = (live && !shared_first) ? sh : -0.0;
  }
}

template <int NR>
__device__ __forceinline__ void combine_config(const hs_dev_t& d, const SideView& vL, const SideView& vR, int lane, uint8_t seed_c, double seed_lc, double seed_lw,
                                               int N, int lead_off, int F0, int trail_off, int slotL, int slotR,
                                               const hs_allele_t& alv, uint64_t members, int kb, int n_alleles, double* out_row){
  CmbRegs<NR> g;
  combine_setup<NR>(d, vL, vR, lane, seed_c, seed_lc, seed_lw, N, lead_off, F0, trail_off, slotL, slotR, g);
  const int wave = threadIdx.x >> 6;
  // the wavefronts of the workgroup share the quads that have a member
  int seen = 0;
  for (int q = 0; q < 16 && kb + 4*q < n_alleles; q++){
    const int want = (int)((members >> (4*q)) & 0xf);
    if (!want) continue;
    if ((seen++ % HS_CMB_WAVES) != wave) continue;
    const int first = __builtin_ctz(want);
    int ord[4];
#pragma unroll
    for (int j = 0; j < 4; j++) ord[j] = rdlane(alv.re_ord, 4*q + (((want >> j) & 1) ? j : first));
    combine_quad<NR>(d, g, ord, want, out_row, kb + 4*q, lane);
  }
}

// One workgroup (HS_CMB_WAVES wavefronts) per active read.
extern "C" __global__ void __launch_bounds__(64*HS_CMB_WAVES)
hs_combine_kernel(const hs_dev_t* __restrict__ dp, int active_begin){
  const hs_dev_t& d = *dp;
  const int lane = threadIdx.x & 63;
  const int ai = active_begin + blockIdx.x;
  const SideView vL = side_view(d, ai, 0);
  const SideView vR = side_view(d, ai, 1);
  const uint8_t seed_c = (uint8_t)d.bases[vL.base_off + vL.nL];
  const uint8_t seed_q = (uint8_t)d.quals[vL.base_off + vL.nL];
  const double seed_lc = d.qual_correct[seed_q], seed_lw = d.qual_error[seed_q];
  const int n_alleles = uni(vL.loc->n_alleles);
  const int hap_begin = uni(vL.loc->hap_begin);
  double* out_row = d.aln_probs + uni(vL.loc->out_off) + (int64_t)(vL.r - uni(vL.loc->read_begin))*n_alleles;
  // allele records and their rowsets are fetched 64 alleles at a time, one allele per lane
  for (int kb = 0; kb < n_alleles; kb += 64){
    const int kl = min(kb + lane, n_alleles - 1);
    const hs_allele_t alv = d.alleles[hap_begin + kl];
    const hs_rowset_t rsl = d.rowsets[alv.lead_rows[0]], rst = d.rowsets[alv.trail_rows[0]];
    uint64_t todo = __ballot(alv.realign != 0 && kb + lane < n_alleles);
    while (todo){
      // the flank configuration of the first allele left, and everybody who shares it
      const int f = __builtin_ctzll(todo);
      const int N = rdlane(alv.n_flank, f), lr = rdlane(alv.lead_rows[0], f), tr = rdlane(alv.trail_rows[0], f);
      const int slotL = rdlane(alv.lead_slot[0], f), slotR = rdlane(alv.lead_slot[1], f);
      const uint64_t members = todo & __ballot(alv.n_flank == N && alv.lead_rows[0] == lr && alv.trail_rows[0] == tr && alv.lead_slot[0] == slotL && alv.lead_slot[1] == slotR);
      todo &= ~members;
      const int lead_off = rdlane(rsl.off, f), F0 = rdlane(rsl.len, f), trail_off = rdlane(rst.off, f);
      if (F0 < 1 || N - F0 < 1 || N > 64*HS_CMB_ROUNDS){
        int seen = 0;
        for (uint64_t mm = members; mm; mm &= mm - 1){
          const int kk = __builtin_ctzll(mm);
          if ((seen++ % HS_CMB_WAVES) != (int)(threadIdx.x >> 6)) continue;
          combine_one(d, vL, vR, lane, seed_c, seed_lc, seed_lw, N, rdlane(alv.re_ord, kk), lead_off, F0, trail_off, slotL, slotR, out_row + kb + kk);
        }
        continue;
      }
      if (N <= 64)       combine_config<1>(d, vL, vR, lane, seed_c, seed_lc, seed_lw, N, lead_off, F0, trail_off, slotL, slotR, alv, members, kb, n_alleles, out_row);
      else if (N <= 128) combine_config<2>(d, vL, vR, lane, seed_c, seed_lc, seed_lw, N, lead_off, F0, trail_off, slotL, slotR, alv, members, kb, n_alleles, out_row);
#if HS_CMB_ROUNDS >= 4
      else if (N <= 192) combine_config<3>(d, vL, vR, lane, seed_c, seed_lc, seed_lw, N, lead_off, F0, trail_off, slotL, slotR, alv, members, kb, n_alleles, out_row);
      else               combine_config<4>(d, vL, vR, lane, seed_c, seed_lc, seed_lw, N, lead_off, F0, trail_off, slotL, slotR, alv, members, kb, n_alleles, out_row);
#endif
    }
  }
}

// ------------------------------------------------------------------ host-side launch helpers (called from api.hip)
// leading flanks of the reads [active_begin, active_begin + n_active) of a chunk: column tables first, then the reads-as-lanes sweep
#ifndef HS_LAT_WAVES
#define HS_LAT_WAVES 8
#endif
#ifndef HS_LAT_ROWS
#define HS_LAT_ROWS 8
#endif
#ifndef HS_LAT_ITEMS
#define HS_LAT_ITEMS 128u        // launches with at most this many flank items take the latency shape
#endif
extern "C" int hs_flank_waves_per_group(){ return HS_COOP_WAVES; }
extern "C" int hs_combine_waves(){ return HS_CMB_WAVES; }
// HIPSTR_FLANK_SYSTOLIC: 1 (default) = launches of at most HS_LAT_ITEMS flank items whose read sides fit HS_SYS_MAXCOLS columns take the
// systolic kernels (a wavefront per alignment); 0 = never; 2 = every launch that fits (tests: the whole suite through this form)
static int systolic_mode(){ const char* e = getenv("HIPSTR_FLANK_SYSTOLIC"); return e ? atoi(e) : 1; }      // (read per launch: tests switch it)
static bool use_systolic(int item_begin, int item_end, int max_cols){
  const int m = systolic_mode();
  // (a wavefront per alignment: up to 64 per item; beyond ~6000 of them the sweeps that share rows across lanes are faster again)
  return m != 0 && max_cols <= HS_SYS_MAXCOLS && (m == 2 || (unsigned)(item_end - item_begin) <= HS_SYS_ITEMS);
}
extern "C" void hs_launch_lead2(unsigned n_active, unsigned n_wavefronts, hipStream_t st, const hs_dev_t* dp, int active_begin, int item_begin, int item_end, int chunk, int max_cols, int n_clear){
  hipLaunchKernelGGL(hs_col_kernel, dim3(n_active), dim3(64), 0, st, dp, active_begin, n_clear);
  if (item_end <= item_begin) return;
  if (use_systolic(item_begin, item_end, max_cols)){
    hipLaunchKernelGGL((hs_flank_systolic<true>), dim3((unsigned)(item_end - item_begin), 64), dim3(64), 0, st, dp, item_begin);
    return;
  }
  // few items (a locus or two per call): the chip is far from full and what counts is the serial length of a sweep, so the bands are
  // half as tall and twice as many (HS_LAT_WAVES x HS_LAT_ROWS: a step is shorter, the pipeline four steps longer): -10 % per sweep
  if ((unsigned)(item_end - item_begin) <= HS_LAT_ITEMS)
    hipLaunchKernelGGL((hs_lead_kernel_coop<HS_LAT_ROWS, HS_LAT_WAVES, 2>), dim3(std::max(1u, std::min(n_wavefronts, 256u))), dim3(64*HS_LAT_WAVES), 0, st, dp, item_begin, item_end, chunk);
  else
    hipLaunchKernelGGL((hs_lead_kernel_coop<HS_COOP_ROWS, HS_COOP_WAVES, HS_COOP_OCC>), dim3(std::max(1u, std::min(n_wavefronts, 256u*HS_COOP_OCC*4/HS_COOP_WAVES))), dim3(64*HS_COOP_WAVES), 0, st, dp, item_begin, item_end, chunk);
}
extern "C" void hs_launch_trail(unsigned n_wavefronts, hipStream_t st, const hs_dev_t* dp, int item_begin, int item_end, int chunk, int max_cols, int max_rows){
  if (item_end > item_begin && use_systolic(item_begin, item_end, max_cols)){
    hipLaunchKernelGGL((hs_flank_systolic<false>), dim3((unsigned)(item_end - item_begin), 64), dim3(64), 0, st, dp, item_begin);
    return;
  }
  const bool lat = (unsigned)(item_end - item_begin) <= HS_LAT_ITEMS;
  // short flanks (production panels: <= 35 bp): three bands of up to 12 rows fill their wavefronts better than four of 9 (p30 trailing flank
  // 4.25 -> 4.02 ms, profiles/r05_notes.md; at 60 rows the 4 x 15 shape is the best by 25 %)
  const bool shrt = max_rows - 1 <= 36;
  if (lat)       hipLaunchKernelGGL((hs_trail_kernel_coop<HS_LAT_ROWS, HS_LAT_WAVES, 2>), dim3(std::max(1u, std::min(n_wavefronts, 256u))), dim3(64*HS_LAT_WAVES), 0, st, dp, item_begin, item_end, chunk);
  else if (shrt) hipLaunchKernelGGL((hs_trail_kernel_coop<12, 3, 3>), dim3(std::max(1u, std::min(n_wavefronts, 256u*3*4/3))), dim3(64*3), 0, st, dp, item_begin, item_end, chunk);
  else hipLaunchKernelGGL((hs_trail_kernel_coop<HS_COOP_ROWS, HS_COOP_WAVES, HS_COOP_OCC>), dim3(std::max(1u, std::min(n_wavefronts, 256u*HS_COOP_OCC*4/HS_COOP_WAVES))), dim3(64*HS_COOP_WAVES), 0, st, dp, item_begin, item_end, chunk);
}
