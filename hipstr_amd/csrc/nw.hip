// nw.hip — Needleman-Wunsch with affine gaps on gfx950: NeedlemanWunsch::Align (NeedlemanWunsch.cpp:370-420) for a batch of
// (reference, read) pairs.  It is the DP in front of the HMM: realign() runs it for every unique read against its reference
// window (AlignmentOps.cpp:14-26) and Haplotype::aln_haps_to_ref for every haplotype against the reference haplotype
// (Haplotype.cpp:58-86, with the end penalty).
//
//   hs_nw_fill<C>   one wavefront per pair, the same systolic sweep as the traceback fill (trace.hip): lane k owns C consecutive read
//                   positions (rows), reference columns enter at lane 0 and move down one lane per step together with the
//                   three scores of the lane's last row at the current and the previous column (v_mov_b32_dpp wave_shr:1).
//                   Scores live in registers only; what goes to HBM is one traceback byte per cell (three 2-bit choices) and
//                   the last row.  Scores are sums of 2, -2, -5, -0.125: exact in float, so no ordering concerns; ties follow
//                   bestIndex (NeedlemanWunsch.cpp:120-140).
//   hs_nw_walk      one thread per pair: findOptimalStop / findOptimalStopEndPenalty (:142-193) on the last row and the pointer
//                   walk of traceAlignment (:247-324); emits the raw operation string.
// The host turns operations into the two gapped strings and the run-length CIGAR.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hipstr_hmm.h"
#include "layout.h"
#include "device_common.h"
#include "api_internal.h"
#include "prep.h"

#define HS_NW_MAX_REF 4095
#define HS_NW_MAX_READ 1536          // rows per lane come from {1,2,3,4,6,8,12,16,24}

struct hs_nw_pair_t {
  int32_t ref_off, L1, read_off, L2;
  int64_t trace_off;       // bytes: (i-1)*L1 + (j-1), i = 1..L2, j = 1..L1
  int64_t last_off;        // floats: M | Iref | Iread of row L2, columns 1..L1 (L1 each)
  int64_t ops_off;         // bytes
};

struct hs_nw_dev_t {
  const hs_nw_pair_t* pairs;
  const int32_t* items;    // pair indices grouped by rows-per-lane class
  const char* refs;
  const char* reads;
  uint8_t* trace;
  float*   last;
  char*    ops;
  float*   score;          // [n]
  int32_t* best_col;       // [n] column where the alignment ends
  int32_t* lead_col;       // [n] reference columns left of the alignment
  int32_t* n_ops;          // [n]
  int32_t  end_penalty;
};

namespace {

constexpr float NW_MATCH = 2.0f, NW_MISMATCH = -2.0f, NW_GAPOPEN = 5.0f, NW_GAPEXTEND = 0.125f, NW_LARGE = 1000000.0f;

__device__ __forceinline__ int nw_base(uint8_t c){              // base_to_int (NeedlemanWunsch.cpp:98-118)
  c &= 0xdf;                                                    // toupper for letters
  return c == 'A' ? 0 : (c == 'C' ? 1 : (c == 'G' ? 2 : (c == 'T' ? 3 : 4)));
}
__device__ __forceinline__ float nw_best(float s1, float s2, float s3, int& c){      // bestIndex (:120-140)
  if (s2 > s1){ if (s2 > s3){ c = 1; return s2; } c = 2; return s3; }
  if (s3 > s1){ c = 2; return s3; }
  c = 0; return s1;
}
__device__ __forceinline__ float shrf(float old, float src){
  return __int_as_float(shr1(__float_as_int(old), __float_as_int(src)));
}

template <int C>
__global__ void __launch_bounds__(64) hs_nw_fill(const hs_nw_dev_t* __restrict__ dp, int item_begin){
  __shared__ uint8_t s_ref[HS_NW_MAX_REF + 1];
  const hs_nw_dev_t& d = *dp;
  const int lane = threadIdx.x;
  const hs_nw_pair_t P = d.pairs[uni(d.items[item_begin + blockIdx.x])];
  const int L1 = uni(P.L1), L2 = uni(P.L2);
  for (int x = lane; x < L1; x += 64) s_ref[x] = (uint8_t)nw_base((uint8_t)d.refs[P.ref_off + x]);
  __syncthreads();
  const int nl = (L2 + C - 1) / C;
  const int last_lane = (L2 - 1) / C, last_r = (L2 - 1) % C;
  uint8_t* tr = d.trace + P.trace_off;
  float* lastM = d.last + P.last_off; float* lastR = lastM + L1; float* lastD = lastR + L1;

  int rb[C]; float pM[C], pR[C], pD[C];
#pragma unroll
  for (int r = 0; r < C; r++){
    const int i = lane*C + r + 1;                               // 1-based read position
    rb[r] = nw_base((uint8_t)d.reads[P.read_off + min(i, L2) - 1]);
    pM[r] = -NW_LARGE; pR[r] = -NW_LARGE; pD[r] = -NW_GAPOPEN - (float)(i-1)*NW_GAPEXTEND;     // column 0 (initMatrices :350-365)
  }
  // what the next lane needs of this lane's last row: at the column just done (cur) and the one before (prev)
  float cM = -NW_LARGE, cR = -NW_LARGE, cD = pD[C-1], qM = 0, qR = 0, qD = 0;
  const int steps = L1 + nl - 1;
  for (int t = 0; t < steps; t++){
    float upM = shrf(0.f, cM), upR = shrf(0.f, cR), upD = shrf(0.f, cD);          // row above, this column
    float dgM = shrf(0.f, qM), dgR = shrf(0.f, qR), dgD = shrf(0.f, qD);          // row above, previous column
    const int j = t - lane + 1;                                 // 1-based reference column of this lane at this step
    const bool on = (j >= 1) && (j <= L1) && (lane < nl);
    if (lane == 0){                                             // row 0 (initMatrices :331-347)
      upM = -NW_LARGE; upD = -NW_LARGE; upR = d.end_penalty ? -NW_GAPOPEN - (float)(j-1)*NW_GAPEXTEND : 0.0f;
      if (j == 1){ dgM = 0.0f; dgR = -NW_LARGE; dgD = -NW_LARGE; }
      else { dgM = -NW_LARGE; dgD = -NW_LARGE; dgR = d.end_penalty ? -NW_GAPOPEN - (float)(j-2)*NW_GAPEXTEND : 0.0f; }
    }
    if (on){
      const int fb = s_ref[j-1];
      qM = cM; qR = cR; qD = cD;                                // becomes "previous column" for the lane below
#pragma unroll
      for (int r = 0; r < C; r++){
        const float lM = pM[r], lR = pR[r], lD = pD[r];         // same row, previous column
        const float sc = (fb == 4 || rb[r] == 4 || fb == rb[r]) ? NW_MATCH : NW_MISMATCH;
        int c0, c1, c2;
        const float nM = nw_best(dgM, dgR, dgD, c0) + sc;
        const float nR = nw_best(lM - NW_GAPOPEN, lR - NW_GAPEXTEND, lD - NW_GAPOPEN, c1);
        const float nD = nw_best(upM - NW_GAPOPEN, upR - NW_GAPOPEN, upD - NW_GAPEXTEND, c2);
        const int i = lane*C + r + 1;
        if (i <= L2) tr[(int64_t)(i-1)*L1 + (j-1)] = (uint8_t)(c0 | (c1 << 2) | (c2 << 4));
        if (lane == last_lane && r == last_r){ lastM[j-1] = nM; lastR[j-1] = nR; lastD[j-1] = nD; }
        dgM = lM; dgR = lR; dgD = lD;                           // this row at j-1 is the diagonal of the next row
        upM = nM; upR = nR; upD = nD;
        pM[r] = nM; pR[r] = nR; pD[r] = nD;
      }
      cM = pM[C-1]; cR = pR[C-1]; cD = pD[C-1];
    }
  }
}

__global__ void __launch_bounds__(64) hs_nw_walk(const hs_nw_dev_t* __restrict__ dp, int n){
  const hs_nw_dev_t& d = *dp;
  const int p = blockIdx.x*64 + threadIdx.x;
  if (p >= n) return;
  const hs_nw_pair_t P = d.pairs[p];
  const int L1 = P.L1, L2 = P.L2;
  const float* lastM = d.last + P.last_off; const float* lastR = lastM + L1; const float* lastD = lastR + L1;
  const uint8_t* tr = d.trace + P.trace_off;
  float best_val; int best_col, best_type;
  const float d0 = -NW_GAPOPEN - (float)(L2-1)*NW_GAPEXTEND;    // row L2, column 0: M, Iref impossible
  if (d.end_penalty){                                           // findOptimalStopEndPenalty (:173-193)
    best_col = L1; best_val = lastM[L1-1]; best_type = 0;
    if (lastR[L1-1] > best_val){ best_val = lastR[L1-1]; best_type = 1; }
    if (lastD[L1-1] > best_val){ best_val = lastD[L1-1]; best_type = 2; }
  } else {                                                      // findOptimalStop (:142-171)
    best_val = -NW_LARGE; best_col = -1; best_type = -1;
    for (int j = 0; j <= L1; j++){
      const float m = j ? lastM[j-1] : -NW_LARGE, r = j ? lastR[j-1] : -NW_LARGE, dd = j ? lastD[j-1] : d0;
      if (m >= best_val){ best_val = m; best_col = j; best_type = 0; }
      if (r > best_val){ best_val = r; best_col = j; best_type = 1; }
      if (dd > best_val){ best_val = dd; best_col = j; best_type = 2; }
    }
  }
  d.score[p] = best_val; d.best_col[p] = best_col;
  // traceAlignment (:262-300): follow the pointers from (L2, best_col)
  char* ops = d.ops + P.ops_off;
  int row = L2, col = best_col, type = best_type, k = 0;
  while (row > 0 && type >= 0){
    if (col == 0 && type != 2){ type = -1; break; }            // the reference would index refseq.at(-1) here; cannot happen
    const int byte = (col >= 1) ? tr[(int64_t)(row-1)*L1 + (col-1)] : (2 << 4);      // column 0: Iread continues (:353-354)
    if (type == 0){
      ops[k++] = (nw_base((uint8_t)d.refs[P.ref_off + col-1]) == nw_base((uint8_t)d.reads[P.read_off + row-1])) ? '=' : 'X';
      type = byte & 3; row--; col--;
    } else if (type == 1){ ops[k++] = 'D'; type = (byte >> 2) & 3; col--; }
    else { ops[k++] = 'I'; type = (byte >> 4) & 3; row--; }
    if (type == 3) type = -1;
  }
  d.n_ops[p] = (row == 0) ? k : -1;
  d.lead_col[p] = col;
}

#define NW_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess){ \
  hipstr::api_fail(std::string(#call) + ": " + hipGetErrorString(e_)); return 1; } } while (0)

struct NwBufs {
  std::vector<void*> p;
  hipstr::Ctx* ctx = NULL;          // blocks come from (and return to) the context's cache: no hipMalloc / hipFree per call
  // the cache serves other host threads: nothing this call queued may still be running on the blocks when they go back (error paths leave early)
  hipStream_t streams[3] = {NULL, NULL, NULL}; int n_streams = 0;
  void runs_on(hipStream_t st){ for (int i = 0; i < n_streams; i++) if (streams[i] == st) return; if (n_streams < 3) streams[n_streams++] = st; }
  ~NwBufs(){ if (ctx){ for (int i = 0; i < n_streams; i++) hipStreamSynchronize(streams[i]); for (void* x : p) hipstr::dev_free(ctx, x); } }
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
    if (count) NW_HIP(hipMemcpy(*out, src, count*sizeof(T), hipMemcpyHostToDevice));
    return 0;
  }
};

}  // namespace

extern "C" int hipstr_nw_align(const hipstr_nw_batch_t* nb, hipstr_nw_out_t* o){
  hipstr::ApiTimer prof_t(hipstr::PB_NW_ALIGN);
  using hipstr::api_fail;
  if (!nb || !o || nb->n_pairs < 0) return api_fail("null argument");
  const int n = nb->n_pairs;
  if (!o->aln_off || !o->cigar_off) return api_fail("null output array");
  o->aln_off[0] = 0; o->cigar_off[0] = 0;
  if (n == 0) return 0;
  if (!o->score || !o->ok || !o->ref_al || !o->read_al || !o->cigar_op || !o->cigar_len) return api_fail("null output array");
  hipstr::ApiTables T;
  if (hipstr::api_device_tables(&T)) return 1;
  std::vector<hs_nw_pair_t> pairs(n);
  for (int i = 0; i < n; i++){
    hs_nw_pair_t& P = pairs[i];
    P.ref_off = nb->ref_off[i]; P.L1 = nb->ref_off[i+1] - nb->ref_off[i];
    P.read_off = nb->read_off[i]; P.L2 = nb->read_off[i+1] - nb->read_off[i];
    if (P.L1 < 1 || P.L2 < 1) return api_fail("Needleman-Wunsch needs non-empty sequences");
    if (P.L2 > HS_NW_MAX_READ) return api_fail("second sequence longer than 1536 bases is not supported");
    if (P.L1 > HS_NW_MAX_REF) return api_fail("reference longer than 4095 bases is not supported");
  }
  hs_nw_dev_t h; memset(&h, 0, sizeof h);
  hipstr::HostArena seqs;                                     // the two sequence pools: one pinned block, one copy
  {
    const size_t o_refs = seqs.add(nb->ref_seqs, (size_t)nb->ref_off[n]), o_reads = seqs.add(nb->read_seqs, (size_t)nb->read_off[n]);
    if (seqs.reserve(T.ctx)) return api_fail("out of device or pinned host memory");
    if (seqs.send(T.stream)) return 1;
    h.refs = seqs.at<char>(o_refs); h.reads = seqs.at<char>(o_reads);
  }
  h.end_penalty = nb->use_ref_end_penalty ? 1 : 0;
  // traceback bytes per chunk; the device is only asked how much it has free when the call could need more than 256 MiB
  int64_t budget = (int64_t)256 << 20;
  {
    int64_t all = 0; for (int i = 0; i < n; i++) all += (int64_t)pairs[i].L1*pairs[i].L2;
    if (all > budget){
      size_t free_b = 0, total_b = 0;
      NW_HIP(hipMemGetInfo(&free_b, &total_b));
      budget = std::min<int64_t>((int64_t)8 << 30, (int64_t)(free_b / 4));
    }
  }
  if (const char* e = getenv("HIPSTR_NW_WS_MIB")) budget = std::max<int64_t>(1, atoll(e)) << 20;
  for (int p0 = 0; p0 < n; ){
    int p1 = p0; int64_t tb = 0, lf = 0, ob = 0;
    while (p1 < n){
      const int64_t need = (int64_t)pairs[p1].L1*pairs[p1].L2;
      if (p1 > p0 && tb + need > budget) break;
      pairs[p1].trace_off = tb; tb += need;
      pairs[p1].last_off = lf; lf += 3*(int64_t)pairs[p1].L1;
      pairs[p1].ops_off = ob; ob += pairs[p1].L1 + pairs[p1].L2;
      p1++;
    }
    const int np = p1 - p0;
    static const int kRows[9] = { 1, 2, 3, 4, 6, 8, 12, 16, 24 };       // rows per lane of the fill kernel instantiations
    std::vector<int32_t> items; int cls_begin[10];
    for (int cl = 0; cl < 9; cl++){
      cls_begin[cl] = items.size();
      for (int i = p0; i < p1; i++){
        const int need = (pairs[i].L2 + 63)/64;
        if (need <= kRows[cl] && (cl == 0 || need > kRows[cl-1])) items.push_back(i - p0);
      }
    }
    cls_begin[9] = items.size();
    NwBufs ws;
    ws.runs_on(T.stream);
    hs_nw_dev_t hc = h;
    if (ws.alloc(&hc.trace, tb) || ws.alloc(&hc.last, lf)) return 1;
    // what comes back — score, stop column, leading columns, operation count per pair and the operation strings — is one device block
    const size_t r_score = 0, r_bcol = r_score + (size_t)np*4, r_lcol = r_bcol + (size_t)np*4, r_nops = r_lcol + (size_t)np*4, r_ops = r_nops + (size_t)np*4,
                 r_end = r_ops + (size_t)(ob ? ob : 1);
    char* d_res;
    if (ws.alloc(&d_res, r_end)) return 1;
    hc.score = (float*)(d_res + r_score); hc.best_col = (int32_t*)(d_res + r_bcol); hc.lead_col = (int32_t*)(d_res + r_lcol);
    hc.n_ops = (int32_t*)(d_res + r_nops); hc.ops = d_res + r_ops;
    hipstr::HostArena ch;                                     // this chunk's pairs, launch order and argument block
    const size_t o_pairs = ch.add(pairs.data() + p0, (size_t)np*sizeof(hs_nw_pair_t)), o_items = ch.add(items.data(), items.size()*sizeof(int32_t)),
                 o_args = ch.add(&hc, sizeof hc);
    if (ch.reserve(T.ctx)) return api_fail("out of device or pinned host memory");
    hc.pairs = ch.at<hs_nw_pair_t>(o_pairs); hc.items = ch.at<int32_t>(o_items);
    if (ch.send(T.stream)) return 1;
    const hs_nw_dev_t* d_args = ch.at<hs_nw_dev_t>(o_args);
    for (int cl = 0; cl < 9; cl++){
      const int cnt = cls_begin[cl+1] - cls_begin[cl];
      if (cnt == 0) continue;
#define NW_LAUNCH(C_) hipLaunchKernelGGL(hs_nw_fill<C_>, dim3(cnt), dim3(64), 0, T.stream, d_args, cls_begin[cl])
      switch (kRows[cl]){
        case 1: NW_LAUNCH(1); break;   case 2: NW_LAUNCH(2); break;   case 3: NW_LAUNCH(3); break;
        case 4: NW_LAUNCH(4); break;   case 6: NW_LAUNCH(6); break;   case 8: NW_LAUNCH(8); break;
        case 12: NW_LAUNCH(12); break; case 16: NW_LAUNCH(16); break; default: NW_LAUNCH(24); break;
      }
#undef NW_LAUNCH
    }
    hipLaunchKernelGGL(hs_nw_walk, dim3((np + 63)/64), dim3(64), 0, T.stream, d_args, np);
    NW_HIP(hipGetLastError());
    char* hostblk = (char*)hipstr::pin_alloc(T.ctx, r_end);
    if (!hostblk) return api_fail("out of pinned host memory");
    struct PinGuard { hipstr::Ctx* c; void* p; ~PinGuard(){ hipstr::pin_free(c, p); } } pin_guard{T.ctx, hostblk};
    NW_HIP(hipMemcpyAsync(hostblk, d_res, r_end, hipMemcpyDeviceToHost, T.stream));
    NW_HIP(hipstr::wait_stream(T.stream));
    const float* score = (const float*)(hostblk + r_score); const int32_t* bcol = (const int32_t*)(hostblk + r_bcol);
    const int32_t* lcol = (const int32_t*)(hostblk + r_lcol); const int32_t* nops = (const int32_t*)(hostblk + r_nops);
    struct { const char* p; const char* data() const { return p; } } ops{hostblk + r_ops};
    // gapped strings and run-length CIGAR (traceAlignment :252-323)
    for (int i = p0; i < p1; i++){
      const hs_nw_pair_t& P = pairs[i];
      const int k = nops[i-p0];
      if (k < 0) return api_fail("Invalid matrix type in Needleman-Wunsch alignment");
      const char* ref = nb->ref_seqs + P.ref_off; const char* rd = nb->read_seqs + P.read_off;
      const int lead = lcol[i-p0], stop = bcol[i-p0];
      const int64_t len = (int64_t)lead + k + (P.L1 - stop);
      int64_t a = o->aln_off[i];
      if (a + len > o->cap_aln) return api_fail("hipstr_nw_out_t alignment pools are too small (cap_aln)");
      for (int j = 0; j < lead; j++){ o->ref_al[a] = ref[j]; o->read_al[a++] = '-'; }
      int ri = lead, qi = 0;
      int64_t co = o->cigar_off[i]; char cur = 0; int run = 0;
      const char* po = ops.data() + P.ops_off;
      for (int x = k-1; x >= 0; x--){
        const char op = po[x];
        if (op == 'D'){ o->ref_al[a] = ref[ri++]; o->read_al[a++] = '-'; }
        else if (op == 'I'){ o->ref_al[a] = '-'; o->read_al[a++] = rd[qi++]; }
        else { o->ref_al[a] = ref[ri++]; o->read_al[a++] = rd[qi++]; }
        if (op == cur) run++;
        else {
          if (run){ if (co >= o->cap_cigar) return api_fail("hipstr_nw_out_t CIGAR pools are too small (cap_cigar)"); o->cigar_op[co] = cur; o->cigar_len[co++] = run; }
          cur = op; run = 1;
        }
      }
      if (run){ if (co >= o->cap_cigar) return api_fail("hipstr_nw_out_t CIGAR pools are too small (cap_cigar)"); o->cigar_op[co] = cur; o->cigar_len[co++] = run; }
      for (int j = stop; j < P.L1; j++){ o->ref_al[a] = ref[j]; o->read_al[a++] = '-'; }
      o->aln_off[i+1] = a; o->cigar_off[i+1] = co;
      o->score[i] = score[i-p0];
      o->ok[i] = 1;                   // Align only fails on a CIGAR that starts or ends with 'S', which traceAlignment never writes (:413-415)
    }
    p0 = p1;
  }
  return 0;
}

namespace {
// Haplotype::adjust_indels (Haplotype.cpp:8-56): move indels of the leading flank to the right, into / up to the repeat block
void adjust_indels(std::string& ref_al, std::string& alt_al, int32_t first_start, int32_t str_start){
  int32_t ref_pos = first_start;
  size_t aln_index = 0;
  while (aln_index < alt_al.size()){
    if (alt_al[aln_index] == '-' && ref_pos < str_start){
      size_t index = aln_index;
      while (index < alt_al.size() && alt_al[index] == '-') index++;
      int32_t pos = ref_pos; size_t del_index = aln_index; const int32_t del_size = (int32_t)(index - aln_index);
      while (index < alt_al.size() && pos < str_start && ref_al[del_index] == ref_al[index]){
        alt_al[del_index] = alt_al[index]; alt_al[index] = '-';
        index++; del_index++; pos++;
      }
      aln_index = index; ref_pos = pos + del_size;
    } else if (ref_al[aln_index] == '-' && ref_pos < str_start){
      size_t index = aln_index;
      while (index < ref_al.size() && ref_al[index] == '-') index++;
      int32_t pos = ref_pos; size_t ins_index = aln_index;
      while (index < ref_al.size() && pos < str_start && alt_al[ins_index] == alt_al[index]){
        ref_al[ins_index] = ref_al[index]; ref_al[index] = '-';
        index++; ins_index++; pos++;
      }
      aln_index = index; ref_pos = pos;
    } else {
      if (ref_al[aln_index] != '-') ref_pos++;
      aln_index++;
    }
  }
}
}  // namespace

extern "C" int hipstr_hap_aln_info(const hipstr_batch_t* b, char* out, int64_t out_cap, int64_t* offs){
  using hipstr::api_fail;
  if (!b || !out || !offs || b->n_loci < 0) return api_fail("null argument");
  // every haplotype sequence (Haplotype::get_seq in Haplotype::next order) against its locus' reference haplotype
  std::string refs, alts; std::vector<int32_t> ref_off(1, 0), alt_off(1, 0), locus_of;
  int opt_cursor = 0;
  for (int l = 0; l < b->n_loci; l++){
    const int32_t* nopts = b->blk_nopts + 3*l;
    std::vector<std::string> opt[3];
    for (int k = 0; k < 3; k++){
      if (nopts[k] < 1) return api_fail("haplotype block without options");
      for (int x = 0; x < nopts[k]; x++, opt_cursor++) opt[k].push_back(std::string(b->seq + b->opt_off[opt_cursor], b->opt_off[opt_cursor+1] - b->opt_off[opt_cursor]));
    }
    const int A = nopts[0]*nopts[1]*nopts[2];
    if (A != b->hap_off[l+1] - b->hap_off[l]) return api_fail("hap_off does not match the product of block options");
    const std::string ref_hap = opt[0][0] + opt[1][0] + opt[2][0];
    for (int k = 0; k < A; k++){
      int32_t oi[3];
      hipstr::allele_options(nopts, k, oi);
      refs += ref_hap; ref_off.push_back((int32_t)refs.size());
      alts += opt[0][oi[0]] + opt[1][oi[1]] + opt[2][oi[2]]; alt_off.push_back((int32_t)alts.size());
      locus_of.push_back(l);
    }
  }
  const int n = (int)locus_of.size();
  offs[0] = 0;
  if (n == 0) return 0;
  hipstr_nw_batch_t nb; nb.n_pairs = n; nb.ref_off = ref_off.data(); nb.ref_seqs = refs.data(); nb.read_off = alt_off.data(); nb.read_seqs = alts.data();
  nb.use_ref_end_penalty = 1;
  const int64_t cap = (int64_t)refs.size() + (int64_t)alts.size() + 16;
  std::vector<float> score(n); std::vector<uint8_t> ok(n); std::vector<int64_t> aoff(n+1), coff(n+1);
  std::vector<char> ra(cap), qa(cap), cop(cap); std::vector<int32_t> clen(cap);
  hipstr_nw_out_t o; o.score = score.data(); o.ok = ok.data(); o.aln_off = aoff.data(); o.ref_al = ra.data(); o.read_al = qa.data();
  o.cigar_off = coff.data(); o.cigar_op = cop.data(); o.cigar_len = clen.data(); o.cap_aln = cap; o.cap_cigar = cap;
  if (hipstr_nw_align(&nb, &o)) return 1;
  int64_t pos = 0;
  for (int i = 0; i < n; i++){
    const int l = locus_of[i];
    std::string r(ra.data() + aoff[i], aoff[i+1] - aoff[i]), a(qa.data() + aoff[i], aoff[i+1] - aoff[i]);
    adjust_indels(r, a, b->blk_start[3*l], b->blk_start[3*l+1]);
    if (pos + (int64_t)r.size() + 1 > out_cap) return api_fail("hap_aln_info output is too small (out_cap)");
    offs[i] = pos;
    for (size_t x = 0; x < r.size(); x++) out[pos++] = r[x] == '-' ? 'I' : (a[x] == '-' ? 'D' : 'M');     // Haplotype.cpp:73-81
    out[pos++] = 0;
  }
  offs[n] = pos;
  return 0;
}
