// prep.cpp — flatten a batch of loci into the device layout (layout.h).
//
// This is the host half of what HapAligner's constructor and the RepeatBlock /
// StutterAlignerClass constructors do in the reference (HapAligner.h:56-69,
// RepeatBlock.h:29-43, StutterAlignerClass.h:52-82): build the reversed haplotype,
// the homopolymer-indexed rows of every flank block, the stutter pmf and the
// periodicity structure of every STR allele — but emitted as flat pools the kernels
// stream, instead of objects the CPU loop walks.  O(A·H) work per locus; the
// O(P·A·H·L) dynamic programme runs on the device.
#include "prep.h"

#include <algorithm>
#include <atomic>
#include <thread>
#include <pthread.h>
#include <chrono>
#include <condition_variable>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <unordered_map>

namespace hipstr {

static std::atomic<int> g_thread_override(0);
void set_host_threads(int n){ g_thread_override = n > 0 ? n : 0; }

static thread_local int tl_thread_budget = 0;
void set_thread_budget(int n){ tl_thread_budget = n > 0 ? n : 0; }

int host_threads(){
  if (tl_thread_budget) return tl_thread_budget;
  if (const int o = g_thread_override.load()) return o;
  static const int n = [](){
    if (const char* e = getenv("HIPSTR_HOST_THREADS")){ const int v = atoi(e); if (v >= 1) return v; }
    unsigned hw = std::thread::hardware_concurrency();
    if (hw == 0) hw = 1;
    // a container may be allowed fewer CPUs than it sees: cgroup v2 cpu.max = "<quota> <period>" (or "max")
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")){
      long long quota = 0, period = 0;
      if (fscanf(f, "%lld %lld", &quota, &period) == 2 && quota > 0 && period > 0) hw = std::min<unsigned>(hw, (unsigned)((quota + period - 1) / period));
      fclose(f);
    }
    return (int)std::max(1u, std::min(hw, 32u));
  }();
  return n;
}

// Work sharing on a persistent pool: the library's host work comes in bursts of a few milliseconds (fragments of a batch, its launch
// plan, the staging copy, the scatter of results), several per batch — starting and joining 15 threads for each of them cost as much
// as some of the bursts.  The pool's threads (host_threads() - 1, started on first use) sleep on a condition variable; a caller posts
// a job — a loop body and a counter that hands out its indices —, works on it itself and returns when every index is done.  Several
// callers may post at once (the stream's worker thread, a collector scattering results): jobs are served oldest first.
namespace {
struct PoolJob {
  const std::function<void(int)>* fn; int n, max_workers;
  std::atomic<int> next{0}, done{0}, workers{0};
};
struct Pool {
  std::mutex m;
  std::condition_variable cv;
  std::vector<PoolJob*> jobs;
  std::vector<std::thread> threads;
  bool stop = false;
  ~Pool(){
    { std::lock_guard<std::mutex> g(m); stop = true; }
    cv.notify_all();
    for (std::thread& t : threads) t.join();
  }
  void ensure(int n_workers){
    std::lock_guard<std::mutex> g(m);
    while ((int)threads.size() < n_workers) threads.emplace_back([this]{ pthread_setname_np(pthread_self(), "hipstr-pool"); serve(); });      // (named: bench.py attributes CPU time by thread)
  }
  static void run(PoolJob* j){
    for (int i = j->next.fetch_add(1); i < j->n; i = j->next.fetch_add(1)){ (*j->fn)(i); j->done.fetch_add(1, std::memory_order_release); }
  }
  void serve(){
    std::unique_lock<std::mutex> g(m);
    for (;;){
      PoolJob* pick = NULL;
      for (PoolJob* j : jobs)
        if (j->next.load(std::memory_order_relaxed) < j->n && j->workers.load(std::memory_order_relaxed) < j->max_workers){ pick = j; break; }
      if (!pick){ if (stop) return; cv.wait(g); continue; }
      pick->workers.fetch_add(1);
      g.unlock();
      run(pick);
      g.lock();
      pick->workers.fetch_sub(1);
      cv.notify_all();                 // the poster may be waiting for the last helper to leave its job
    }
  }
};
Pool& pool(){ static Pool p; return p; }
}  // namespace

void parallel_for(int n, int max_threads, const std::function<void(int)>& fn){
  const int nt = std::max(1, std::min(n, max_threads));
  if (nt == 1){ for (int i = 0; i < n; i++) fn(i); return; }
  constexpr bool no_pool = false;
  if (no_pool){
    std::atomic<int> next(0);
    auto work = [&](){ for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) fn(i); };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; t++) th.emplace_back(work);
    work();
    for (std::thread& t : th) t.join();
    return;
  }
  Pool& P = pool();
  P.ensure(std::max(host_threads(), nt) - 1);
  PoolJob job; job.fn = &fn; job.n = n; job.max_workers = nt - 1;
  { std::lock_guard<std::mutex> g(P.m); P.jobs.push_back(&job); }
  P.cv.notify_all();
  Pool::run(&job);
  std::unique_lock<std::mutex> g(P.m);       // indices are handed out; wait for the helpers still inside the body, then retire the job
  P.cv.wait(g, [&]{ return job.done.load(std::memory_order_acquire) == n && job.workers.load() == 0; });
  P.jobs.erase(std::find(P.jobs.begin(), P.jobs.end(), &job));
}


static const int MIN_SEED_DIST = 5;          // HapAligner.cpp:17
#define HIPSTR_MAX_PERIOD_TAB 9               // longest period (stutter_model.h:38): a periodic block's insertion list has period + 2 table entries
static const double LARGE_NEGATIVE = -10e6;  // RepeatStutterInfo.h:12

const HostTables& host_tables(){
  static HostTables t;
  static std::once_flag once;
  std::call_once(once, [](){
    t.int_log.resize(10000);
    t.int_log[0] = -1000;
    for (int i = 1; i < 10000; i++) t.int_log[i] = log((double)i);
    static const double dindel[10] = {2.9e-5, 2.9e-5, 2.9e-5, 2.9e-5, 4.3e-5, 1.1e-4, 2.4e-4, 5.7e-4, 1.0e-3, 1.4e-3};
    t.m2m[0] = t.m2i[0] = 0;
    for (unsigned int i = 1; i <= 15; i++){
      t.m2i[i] = (i <= 10 ? log(dindel[i-1]) : log(dindel[9]+(4.3e-4)*(i-10)));
      t.m2m[i] = log(1.0 - exp(t.m2i[i]) - exp(t.m2i[i]));
    }
    const int maxq = 'J'-'!';
    std::vector<double> lc(maxq+1), le(maxq+1);
    lc[0] = -100000; le[0] = -log(3);
    for (int i = 1; i <= maxq; i++){
      lc[i] = log(1.0 - pow(10.0, i/(-10.0)));
      le[i] = log(pow(10.0, i/(-10.0))/3.0);
    }
    t.qual_correct.resize(256); t.qual_error.resize(256);
    for (int c = 0; c < 256; c++){
      char q = (char)c;                       // `char` is signed in the reference build as well
      int idx = q < '!' ? 0 : (q > 'J' ? maxq : q-'!');
      t.qual_correct[c] = lc[idx]; t.qual_error[c] = le[idx];
    }
    t.log_thresh = log(0.001);
    t.log_half   = log(0.5);
  });
  return t;
}

double log_stutter_pmf(const double* sp, int period, int sample_bps, int read_bps){
  const double in_step = log(1-sp[0]), in_nostep = log(sp[0]), in_up = log(sp[1]), in_down = log(sp[2]);
  const double out_step = log(1-sp[3]), out_nostep = log(sp[3]), out_up = log(sp[4]), out_down = log(sp[5]);
  const int diff = read_bps - sample_bps;
  if (diff % period != 0){
    const int eff = diff - diff/period;
    return eff < 0 ? out_down + out_nostep + out_step*(-eff-1) : out_up + out_nostep + out_step*(eff-1);
  }
  const int rep = diff/period;
  if (rep == 0) return log(1-sp[1]-sp[2]-sp[4]-sp[5]);
  return rep < 0 ? in_down + in_nostep + in_step*(-rep-1) : in_up + in_nostep + in_step*(rep-1);
}

void allele_options(const int32_t nopts[3], int k, int32_t opts[3]){
  // reflected mixed-radix Gray code, block 0 the fastest digit
  int f = 1;
  for (int i = 0; i < 3; i++){
    const int n = nopts[i];
    const int digit = (k / f) % n;
    const bool reflected = ((k / (f*n)) & 1) != 0;
    opts[i] = reflected ? n-1-digit : digit;
    f *= n;
  }
}

// block whose option changes when stepping from allele k-1 to k (k >= 1)
static int changed_block(const int32_t nopts[3], int k){
  const int f1 = nopts[0], f2 = nopts[0]*nopts[1];
  if (k % f2 == 0) return 2;
  if (k % f1 == 0) return 1;
  return 0;
}

int calc_seed_base(const hipstr_batch_t* b, int l, int r){
  const int32_t win_lo = b->blk_start[3*l], win_hi = b->blk_end[3*l+2]-1;
  const int32_t rep_lo = b->blk_start[3*l+1], rep_hi = b->blk_end[3*l+1];   // [rep_lo, rep_hi)
  int32_t pos = b->read_start[r];
  int best = -1, consumed = 0, best_dist = MIN_SEED_DIST;
  for (int c = b->cigar_off[r]; c < b->cigar_off[r+1]; c++){
    const int num = b->cigar_len[c];
    const char op = b->cigar_op[c];
    if (op == '='){
      const int32_t lo = std::max(pos, win_lo), hi = std::min(pos+num-1, win_hi);
      if (lo <= hi){
        // candidate sub-intervals of [lo,hi] free of the repeat: the part left of it, then the part right of it;
        // ties go to the later one (the reference uses >= throughout)
        int32_t d = -1, p = -1;
        int32_t cur = lo;
        if (cur < rep_lo){
          const int32_t e = std::min(hi, rep_lo-1);
          d = 1 + (e-cur)/2; p = cur + d - 1;
          cur = rep_hi;
        } else if (cur < rep_hi) cur = rep_hi;
        if (cur <= hi){
          const int32_t d2 = 1 + (hi-cur)/2;
          if (d2 >= d){ d = d2; p = cur + d2 - 1; }
        }
        if (d >= best_dist){ best_dist = d; best = consumed + (p - pos); }
      }
      pos += num; consumed += num;
    }
    else if (op == 'X'){ pos += num; consumed += num; }
    else if (op == 'I') consumed += num;
    else if (op == 'D') pos += num;
    else return -2;    // "Unrecognized CIGAR char in calc_seed_base()"
  }
  const int len = b->base_off[r+1]-b->base_off[r];
  if (best < -1 || best == 0 || best >= len-1) return -2;   // "Invalid alignment seed"
  return best;
}

namespace {

struct SideSeqs {                 // the three block sequences of one allele in one orientation
  std::string s[3];
  std::vector<int> lrun[3], rrun[3];
  // Run-length tables of HapBlock::calc_homopolymer_lengths (HapBlock.cpp:7-30).  The reference
  // does not reset its counter between the forward and the backward pass, so the backward
  // ("right") lengths near the end of a block are inflated by the block's trailing run; the
  // transition tables are indexed with those values, so the quirk is reproduced on purpose.
  void index(){
    for (int b = 0; b < 3; b++){
      const std::string& q = s[b];
      const int n = q.size();
      lrun[b].assign(n, 0); rrun[b].assign(n, 0);
      if (n == 0) continue;
      int count = 0;
      for (int j = 1; j < n; j++){ count = (q[j-1] == q[j]) ? count+1 : 0; lrun[b][j] = count; }
      for (int j = n-2; j >= 0; j--){ count = (q[j+1] == q[j]) ? count+1 : 0; rrun[b][j] = count; }
    }
  }
};

// Haplotype::homopolymer_length with its neighbour-block extensions (Haplotype.cpp:239-287).
int homopolymer_len(const SideSeqs& h, int bi, int pos){
  const std::string& q = h.s[bi];
  const char c = q[pos];
  int l = h.lrun[bi][pos], r = h.rrun[bi][pos];
  if (pos - l == 0){
    for (int nb = bi-1; nb >= 0; nb--){
      const int n = h.s[nb].size();
      if (n == 0) continue;
      if (h.s[nb][n-1] != c) break;
      const int ll = h.lrun[nb][n-1];
      l += 1 + ll;
      if (ll != n) break;
    }
  }
  if (pos + r == (int)q.size()-1){
    for (int nb = bi+1; nb < 3; nb++){
      const int n = h.s[nb].size();
      if (n == 0) continue;
      if (h.s[nb][0] != c) break;
      const int rl = h.rrun[nb][0];
      r += 1 + rl;
      if (rl != n) break;
    }
  }
  return l + r + 1;
}

// rows of flank block `bi` (0 = lead, 2 = trail) under context h; u0 = compact index of its first row
std::vector<hs_row_t> flank_rows(const SideSeqs& h, int bi, int u0){
  const std::string& q = h.s[bi];
  std::vector<hs_row_t> rows(q.size());
  for (int i = 0; i < (int)q.size(); i++){
    const int hl = std::min(HIPSTR_MAX_HOMOP_LEN, std::max(homopolymer_len(h, bi, i), homopolymer_len(h, bi, std::max(0, i-1))));
    rows[i] = HS_ROW_VALID | ((uint32_t)(u0+i) << 12) | ((uint32_t)hl << 8) | (uint8_t)q[i];
  }
  return rows;
}

inline hs_visit_t visit(int ni, int U, char ca, char cb, bool plain, double logU){
  hs_visit_t v;
  v.meta = (uint64_t)(uint16_t)ni | ((uint64_t)(uint16_t)U << 16) | ((uint64_t)(uint8_t)ca << 32) | ((uint64_t)(uint8_t)cb << 40) | ((uint64_t)(plain ? 1 : 0) << 48);
  v.logU = logU;
  return v;
}

// Scratch of the host preparation, one per host thread: the per-locus path allocates nothing once these have grown.
struct Scratch {
  std::vector<int> up;                  // StutterAlignerClass's upstream-match run table of one shift (h:35-42)
  std::vector<hs_visit_t> vis;          // the visiting lists of the STR option being built
  std::vector<char> rev;                // reversed block sequences of the locus being prepared
  std::vector<int> lrun, rrun;          // HapBlock run lengths of the three blocks of one haplotype orientation
  std::vector<hs_row_t> rows;           // rows of one flank block before they are interned
};
static Scratch& scratch(){ thread_local Scratch s; return s; }

// StutterAlignerClass.h:35-42 for one shift, into the thread's scratch
static const int* upstream_runs(const char* s, int n, int shift){
  std::vector<int>& m = scratch().up;
  if ((int)m.size() < n) m.resize(n);
  int* p = m.data();
  for (int i = 0; i < shift && i < n; i++) p[i] = 0;
  for (int i = shift; i < n; i++) p[i] = (s[i-shift] != s[i]) ? 0 : 1 + p[i-1];
  return p;
}

// One STR option in one orientation -> hs_stropt_t (+ its pools)
static std::atomic<bool> g_host_tables(false);     // HIPSTR_HOST_TABLES=1 (tests, comparison runs; re-read at the start of every prepare_batch): constants and tables written by the host, as up to round 3
static std::atomic<double> g_bnd_scale(1.0);      // tests only (HIPSTR_DEBUG_BND_SCALE): re-read from the environment at the start of every prepare_batch; concurrent calls store the same value

// ---- host copies of the float bit tricks (fastonebigheader.h:206-218, 348-358), used to tabulate the closed form below
static float h_fasterexp(float p){
  const float y = 1.442695040f * p;
  const float c = (y < -126.0f) ? -126.0f : y;
  const float z = c + 126.94269504f;
  const uint32_t u = (uint32_t)(8388608.0f * z);
  float r; memcpy(&r, &u, 4); return r;
}
static float h_fasterlog(float x){
  uint32_t u; memcpy(&u, &x, 4);
  float y = (float)u;
  y = y * 8.2629582881927490e-8f;
  return y - 87.989971088f;
}

// One entry {A, G, Bnd} of the tabulated closed form of a simple visiting list (hmm_kernels.hip simple_eval is the long form):
// the pushed values are lp0 (1 + nplain times), ln(U0) + lp0 and ln(tail - stop) + lp0, so with A = the largest of the added
// constants the maximum is fl(lp0 + A) (rounding is monotone) and the float arguments of the exponentials are the differences
// of the constants, off by at most the rounding errors of the sums: |err| <= 2^-52 (|lp0| + 3A).  If every such difference
// keeps its float rounding and its side of LOG_THRESH under an error of 2^-50 (|lp0| + A + 1), the sum of exponentials and
// hence G = fasterlog(sum) do not depend on lp0.  Bnd is the largest |lp0| for which that is guaranteed.
static void simple_table_entry_compute(int lim, int U0, int tail, double ent[3]){
  const HostTables& T = host_tables();
  const bool skip = (U0 > 0) && (lim > 0);
  const int nplain = std::max(0, lim - U0);
  const int stop = (lim <= 0) ? 0 : ((U0 > 0 && lim <= U0) ? U0 : lim);
  const bool has_tail = stop < tail;
  const double a[3] = {0.0, T.int_log[U0], T.int_log[std::max(tail - stop, 0)]};
  const bool on[3] = {true, skip, has_tail};
  const double w[3] = {(double)(1 + nplain), 1.0, 1.0};
  double amax = 0.0;
  for (int i = 1; i < 3; i++) if (on[i] && a[i] > amax) amax = a[i];
  double tot = 0.0, delta = 1e300;
  for (int i = 0; i < 3; i++){
    if (!on[i]) continue;
    if (a[i] == amax){ tot += w[i] * (double)h_fasterexp(0.0f); continue; }        // the device difference is exactly zero
    const double x = a[i] - amax;
    const float f = (float)x;
    const double m_lo = 0.5*((double)f + (double)nextafterf(f, -INFINITY)), m_hi = 0.5*((double)f + (double)nextafterf(f, INFINITY));
    delta = std::min(delta, std::min(x - m_lo, m_hi - x));
    delta = std::min(delta, fabs(x - T.log_thresh));
    if (x > T.log_thresh) tot += w[i] * (double)h_fasterexp(f);
  }
  ent[0] = amax;
  ent[1] = (double)h_fasterlog((float)tot);
  ent[2] = (delta >= 1e300) ? 1e300 : std::max(0.0, delta * 1125899906842624.0 /* 2^50 */ - amax - 1.0);
  if (ent[2] < 1e300) ent[2] *= g_bnd_scale.load();      // tests: shrink the guarantee (HIPSTR_DEBUG_BND_SCALE)
}
// The entries are functions of three small integers and the same ones come back for every allele of every locus (a periodic block's lists
// have U0 = tail or tail - period): a direct-mapped memo per host thread turns the ~50 float steps of an entry into one probe.
static void simple_table_entry(int lim, int U0, int tail, double ent[3]){
  if ((unsigned)lim >= 4096u || (unsigned)U0 >= 4096u || (unsigned)tail >= 4096u || g_bnd_scale.load(std::memory_order_relaxed) != 1.0){
    simple_table_entry_compute(lim, U0, tail, ent); return;
  }
  struct Slot { uint64_t key; double v[3]; };
  thread_local std::vector<Slot> memo(8192, Slot{0, {0, 0, 0}});
  const uint64_t key = (uint64_t)lim | ((uint64_t)U0 << 12) | ((uint64_t)tail << 24) | ((uint64_t)1 << 40);
  Slot& s = memo[(size_t)((key * 0x9E3779B97F4A7C15ull) >> 51)];      // 13 bits
  if (s.key != key){ simple_table_entry_compute(lim, U0, tail, s.v); s.key = key; }
  ent[0] = s.v[0]; ent[1] = s.v[1]; ent[2] = s.v[2];
}

// The 13 values of StutterModel::log_stutter_pmf (stutter_model.cpp:29-53) an STR option needs — artifact sizes of -6..+6 repeat units, all
// in frame — depend on the stutter model alone, not on the block: computed once per locus (eight logarithms per value otherwise).
static void stutter_pmf13(const double* sp, int period, double pmf[HS_NART]){
  (void)period;                                  // (sizes of whole repeat units: the in-frame branch of log_stutter_pmf, same expressions)
  const double in_step = log(1-sp[0]), in_nostep = log(sp[0]), in_up = log(sp[1]), in_down = log(sp[2]);
  for (int t = 0; t < HS_NART; t++){
    const int rep = t - HS_MAXREP;
    pmf[t] = rep == 0 ? log(1-sp[1]-sp[2]-sp[4]-sp[5]) : (rep < 0 ? in_down + in_nostep + in_step*(-rep-1) : in_up + in_nostep + in_step*(rep-1));
  }
}

static std::atomic<uint64_t> g_so_cycles[6];
static const bool g_so_on = getenv("HIPSTR_PREP_PROFILE") != NULL;
#define HS_SOLAP(k) do { if (g_so_on){ const uint64_t now_ = __builtin_ia32_rdtsc(); g_so_cycles[k] += now_ - so_t; so_t = now_; } } while (0)
// pmf_off >= 0: the option may leave its constants and table to the device (hs_stropt_t::gen; the locus' pmf values sit at out.pmf13[pmf_off..])
// twin >= 0: the option at out.stropts[twin] is this block in the other orientation: what does not depend on the orientation (is the
// whole block periodic? only A/C/G/T?) is taken from it
void emit_stropt(const char* blk, const int B, int period, const double pmf13[HS_NART], Prepared& out, bool forward_only = false, int pmf_off = -1, int twin = -1){
  uint64_t so_t = g_so_on ? __builtin_ia32_rdtsc() : 0;
  const HostTables& T = host_tables();
  hs_stropt_t so; memset(&so, 0, sizeof so);
  so.seq_off = out.chars.size();
  {
    const size_t padded = ((size_t)B + 3) & ~(size_t)3;
    out.chars.resize((size_t)so.seq_off + padded);           // (value-initialised: the padding is zero)
    memcpy(out.chars.data() + so.seq_off, blk, (size_t)B);
  }
  so.B = B; so.period = period;
  int nd = HIPSTR_MAX_STUTTER_REPS;
  while (nd*period > B) nd--;
  so.nd = nd;
  // leading run (from the right end) on which the block repeats with the period: base k from the end equals base (k mod period) from
  // the end — which, while it has held for every earlier k, is the same as "equals the base one period to its right"
  const hs_stropt_t* tw = twin >= 0 ? &out.stropts[twin] : NULL;
  const bool twin_periodic = tw && tw->B == B && tw->period == period && tw->nd_eq == tw->nd && tw->shape[HS_MAXREP] == (period < B ? B - period : 0) && tw->kind == 1;
  int per_len;
  if (twin_periodic) per_len = B;          // (a tabulated twin whose insertion list is one run over B - period positions: periodic as a whole, and so is its mirror image)
  else {
    per_len = std::min(period, B);
    while (per_len < B && blk[B-1-per_len] == blk[B-1-per_len+period]) per_len++;
  }
  so.nd_eq = std::min(nd, per_len / period);
  for (int r = 0; r < 16 && r < B; r++) so.tail_codes |= (int32_t)(((uint32_t)(uint8_t)blk[B-1-r] >> 1) & 3u) << (2*r);
  const bool gen = per_len == B && pmf_off >= 0 && !g_host_tables.load(std::memory_order_relaxed) && g_bnd_scale.load(std::memory_order_relaxed) == 1.0;
  so.gen = gen ? 1 : 0; so.pmf_off = gen ? pmf_off : 0;
  if (gen){ so.f64_off = (int32_t)out.gen_f64; out.gen_f64 += HS_NART + 1 + HS_MAXREP; }
  else {
    so.f64_off = out.f64pool.size();
    out.f64pool.resize(so.f64_off + HS_NART + 1 + HS_MAXREP);
    double* c = out.f64pool.data() + so.f64_off;
    for (int t = 0; t < HS_NART; t++) c[t] = (B + (t - HS_MAXREP)*period < 0) ? LARGE_NEGATIVE : pmf13[t];
    c[HS_NART] = -T.int_log[B+1];                                             // StutterAlignerClass.cpp:64
    for (int q = 0; q < HS_MAXREP; q++){
      const int D = -(q+1)*period;
      c[HS_NART + 1 + q] = B+D >= 0 ? -T.int_log[B+D+1] : 0.0;                // StutterAlignerClass.cpp:112
    }
  }
  HS_SOLAP(0);
  const size_t visits_begin = out.visits.size();
  if (per_len == B){
    // The whole block repeats with the period (nearly every candidate allele): the run tables are up_s[i] = i - s + 1 for every shift s
    // that is a multiple of the period, so the loops of StutterAlignerClass.cpp:74-96 / 123-142 visit
    //   insertions:  one run of B - period positions, then the period plain ones, then the end   -> simple list, U0 = B - period
    //                (a block of at most one unit: plain positions only, U0 = 0)
    //   deletions of (q+1) units: one run over the whole remainder, then the end                 -> simple list, U0 = remainder
    // — what the general code below finds by building the seven tables and lists; here it is written down (tests/test_prep.py and the
    // digests of tests/golden/prep_digests.json hold the two to the same bytes).  No list is kept: every shape has a closed form.
    so.ins_off = (int32_t)visits_begin; so.ins_len = 0;
    so.shape[HS_MAXREP] = period < B ? B - period : 0;
    for (int q = 0; q < HS_MAXREP; q++){
      const int tail = B - (q+1)*period;
      so.del_off[q] = (int32_t)visits_begin; so.del_len[q] = 0;
      so.shape[q] = tail >= 0 ? tail : -1;
    }
    HS_SOLAP(1);
    so.tab_off = gen ? (int32_t)out.gen_f64 : (int32_t)out.f64pool.size(); so.tab_len = 0;
    bool ok = (B >= period);
    if (period > HS_GRP_MAXP) ok &= (B <= HS_GRP_MAX_BLOCK);          // (no hs_str_group_kernel_p for this period: hs_str_group_kernel fetches the block, at most HS_GRP_MAX_BLOCK bases)
    if (!twin_periodic) for (int i = 0; i < B; i++){ const char c = blk[i]; ok &= (c == 'A' || c == 'C' || c == 'G' || c == 'T'); }      // (a twin of kind 1 has passed this)
    if (ok && gen){
      // the device writes the entries (expand_kernels.hip): here only their number and where each list's begin
      int total = 0;
      for (int k = 0; k <= HS_MAXREP; k++){
        const int tail = (k == HS_MAXREP) ? B : B - (k+1)*period;
        so.tab_base[k] = total;
        if (tail >= 0) total += 2 + (tail - so.shape[k]);
      }
      so.kind = 1; so.tab_len = total;
      out.gen_f64 += 3*(int64_t)total + 1;
    } else if (ok){
      // the table of such a block is a function of (B, period) alone — 14 + period + [period < B ? 0 : ...] entries — and the same
      // lengths come back for both orientations, for the alleles of neighbouring loci ...: a direct-mapped memo per host thread
      struct Tab { int32_t key, total; int32_t tab_base[HS_MAXREP + 1]; double ent[3*(2*HS_MAXREP + 2 + HIPSTR_MAX_PERIOD_TAB) + 1]; };
      thread_local std::vector<Tab> memo(512);
      const bool use_memo = g_bnd_scale.load(std::memory_order_relaxed) == 1.0;
      const int32_t key = (B << 4) | period;                     // (never 0: B >= 1)
      Tab local; Tab& T2 = use_memo ? memo[((uint32_t)key * 2654435761u) >> 23] : local;
      if (!use_memo || T2.key != key){
        int total = 0;
        for (int k = 0; k <= HS_MAXREP; k++){
          const int tail = (k == HS_MAXREP) ? B : B - (k+1)*period;
          T2.tab_base[k] = total;
          if (tail >= 0) total += 2 + (tail - so.shape[k]);
        }
        double* ent = T2.ent;
        double bmin = 1e300;                              // what the kernel compares |lp0| with: the weakest guarantee of the table
        for (int k = 0; k <= HS_MAXREP; k++){
          const int tail = (k == HS_MAXREP) ? B : B - (k+1)*period;
          if (tail < 0) continue;
          const int U0 = so.shape[k], n = 2 + (tail - U0);
          for (int e = 0; e < n; e++, ent += 3){
            simple_table_entry((e == 0) ? 0 : (e == 1 ? 1 : U0 + e - 1), U0, tail, ent);
            bmin = std::min(bmin, ent[2]);
          }
        }
        *ent = bmin;
        T2.total = total; T2.key = key;
      }
      so.kind = 1;                      // (total <= 12 + 2 + 9 entries: always within HS_TAB_CAP)
      memcpy(so.tab_base, T2.tab_base, sizeof so.tab_base);
      out.f64pool.resize((size_t)so.tab_off + 3*(size_t)T2.total + 1);
      memcpy(out.f64pool.data() + so.tab_off, T2.ent, sizeof(double)*(3*(size_t)T2.total + 1));
      so.tab_len = T2.total;
    }
    out.stropts.push_back(so);
    HS_SOLAP(4);
    return;
  }
  std::vector<hs_visit_t>& V = scratch().vis;             // the option's lists; they go to the visit pool only if the kernels will read them
  V.clear();
  // insertion visiting list (StutterAlignerClass.cpp:74-96); uses the shift-`period` run table
  {
    const int* up = upstream_runs(blk, B, period);
    so.ins_off = (int32_t)(visits_begin + V.size());
    int i = 0;
    for (;;){
      const int ni = -i;
      if (ni >= B){ V.push_back(visit(ni, 0, 0, 0, true, 0)); break; }
      if (-i + period < B){
        const int U = up[B-1+i];
        if (U == 0){ V.push_back(visit(ni, 0, blk[B-1+i], blk[B-1+i-period], false, 0)); i -= 1; }
        else       { V.push_back(visit(ni, U, 0, 0, false, T.int_log[U])); i -= U; }
      } else { V.push_back(visit(ni, 0, 0, 0, true, 0)); i -= 1; }
    }
    so.ins_len = (int32_t)(visits_begin + V.size()) - so.ins_off;
  }
  auto classify = [&](int off, int len, int limmax) -> int {
    // simple: [optional entry (ni=0, U>0)] then plain entries at consecutive offsets up to the terminal at limmax
    int v = 0, next = 0, U0 = 0;
    const hs_visit_t* L = V.data() + (off - (int)visits_begin);
    auto ni_of = [](const hs_visit_t& e){ return (int)(e.meta & 0xffff); };
    auto U_of  = [](const hs_visit_t& e){ return (int)((e.meta >> 16) & 0xffff); };
    auto plain = [](const hs_visit_t& e){ return ((e.meta >> 48) & 1) != 0; };
    if (len >= 1 && ni_of(L[0]) == 0 && !plain(L[0]) && U_of(L[0]) > 0 && limmax > 0){ U0 = U_of(L[0]); next = U0; v = 1; }
    for (; v < len; v++){
      if (ni_of(L[v]) != next) return -1;
      if (next >= limmax) return (v == len-1) ? U0 : -1;      // terminal entry
      if (!plain(L[v])) return -1;
      next++;
    }
    return -1;
  };
  // "piecewise simple" lists (a periodic block with one or two interruptions): [run?] break [run?] break [run?] plain... terminal.
  // Between two emission-update ("break") entries the running likelihood is constant, so such a list has a closed form too.
  // Ten 64-bit slots per list go to the f64 pool right after the 20 constants (bit patterns: two int32 or one double per slot):
  //   0: nseg (-1 = not piecewise) | terminal ni    1,3,5: run ni | run U of segment 0,1,2    2,4,6: ln(run U)
  //   7,8: break ni | ca + 256 cb of break 0,1      9: first plain ni | one past the last plain ni
  auto piecewise = [&](int off, int len, int limmax, uint64_t slot[HS_PW_SLOTS]){
    auto pack = [](int lo, int hi){ return (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32); };
    auto bits = [](double v){ uint64_t u; memcpy(&u, &v, 8); return u; };
    for (int i = 0; i < HS_PW_SLOTS; i++) slot[i] = 0;
    slot[0] = pack(-1, 0);
    const hs_visit_t* E = V.data() + (off - (int)visits_begin);
    auto ni_of = [](const hs_visit_t& e){ return (int)(e.meta & 0xffff); };
    auto U_of  = [](const hs_visit_t& e){ return (int)((e.meta >> 16) & 0xffff); };
    auto plain = [](const hs_visit_t& e){ return ((e.meta >> 48) & 1) != 0; };
    int idx = 0, k = 0;
    for (;;){
      if (idx < len && !plain(E[idx]) && U_of(E[idx]) > 0 && ni_of(E[idx]) < limmax){
        slot[1 + 2*k] = pack(ni_of(E[idx]), U_of(E[idx])); slot[2 + 2*k] = bits(E[idx].logU); idx++;
      }
      if (idx < len && !plain(E[idx]) && U_of(E[idx]) == 0 && ni_of(E[idx]) < limmax){
        if (k == 2) return false;
        slot[7 + k] = pack(ni_of(E[idx]), (int)((E[idx].meta >> 32) & 0xffff)); idx++; k++;
        continue;
      }
      break;
    }
    if (k == 0) return false;                         // no interruption: the simple classifier's business (slot 0 still says "none")
    int pa = 0, pb = 0;
    if (idx < len && plain(E[idx]) && ni_of(E[idx]) < limmax){
      pa = ni_of(E[idx]); pb = pa;
      while (idx < len && plain(E[idx]) && ni_of(E[idx]) < limmax){ if (ni_of(E[idx]) != pb) return false; pb++; idx++; }
    }
    if (idx != len-1 || !plain(E[idx]) || ni_of(E[idx]) < limmax) return false;       // exactly the terminal entry must remain
    slot[9] = pack(pa, pb);
    slot[0] = pack(k, ni_of(E[idx]));
    return true;
  };
  // The same closed form with more levels (round 4): three to HS_PWK_MAX breaks — two or three interruptions of the repeat, what a
  // panel's imperfect loci carry in every candidate allele.  Slots (layout.h HS_PWK_SLOTS): 0 nseg | terminal ni, 1 first plain ni | one
  // past the last, then per segment s = 0..nseg: run ni | run U, ln(run U), break ni | ca + 256 cb (no break behind the last segment).
  // Only hs_str_group_kernel_rp reads them; every other consumer of such a list replays it (its entries are kept).
  auto piecewise_k = [&](int off, int len, int limmax, uint64_t slot[HS_PWK_SLOTS]){
    auto pack = [](int lo, int hi){ return (uint64_t)(uint32_t)lo | ((uint64_t)(uint32_t)hi << 32); };
    auto bits = [](double v){ uint64_t u; memcpy(&u, &v, 8); return u; };
    for (int i = 0; i < HS_PWK_SLOTS; i++) slot[i] = 0;
    slot[0] = pack(-1, 0);
    const hs_visit_t* E = V.data() + (off - (int)visits_begin);
    auto ni_of = [](const hs_visit_t& e){ return (int)(e.meta & 0xffff); };
    auto U_of  = [](const hs_visit_t& e){ return (int)((e.meta >> 16) & 0xffff); };
    auto plain = [](const hs_visit_t& e){ return ((e.meta >> 48) & 1) != 0; };
    int idx = 0, k = 0;
    for (;;){
      if (idx < len && !plain(E[idx]) && U_of(E[idx]) > 0 && ni_of(E[idx]) < limmax){
        slot[2 + 3*k] = pack(ni_of(E[idx]), U_of(E[idx])); slot[3 + 3*k] = bits(E[idx].logU); idx++;
      }
      if (idx < len && !plain(E[idx]) && U_of(E[idx]) == 0 && ni_of(E[idx]) < limmax){
        if (k == HS_PWK_MAX) return false;
        slot[4 + 3*k] = pack(ni_of(E[idx]), (int)((E[idx].meta >> 32) & 0xffff)); idx++; k++;
        continue;
      }
      break;
    }
    if (k < 3) return false;                          // fewer breaks: the ten-slot form's business
    int pa = 0, pb = 0;
    if (idx < len && plain(E[idx]) && ni_of(E[idx]) < limmax){
      pa = ni_of(E[idx]); pb = pa;
      while (idx < len && plain(E[idx]) && ni_of(E[idx]) < limmax){ if (ni_of(E[idx]) != pb) return false; pb++; idx++; }
    }
    if (idx != len-1 || !plain(E[idx]) || ni_of(E[idx]) < limmax) return false;       // exactly the terminal entry must remain
    slot[1] = pack(pa, pb);
    slot[0] = pack(k, ni_of(E[idx]));
    return true;
  };
  constexpr bool group_replay = true, pwk_on = true;
  uint64_t pw[HS_MAXREP + 1][HS_PW_SLOTS];
  for (int k = 0; k <= HS_MAXREP; k++){ for (int i = 0; i < HS_PW_SLOTS; i++) pw[k][i] = 0; pw[k][0] = (uint64_t)(uint32_t)-1; }   // "not piecewise"
  uint64_t pwk[HS_MAXREP + 1][HS_PWK_SLOTS];
  for (int k = 0; k <= HS_MAXREP; k++){ for (int i = 0; i < HS_PWK_SLOTS; i++) pwk[k][i] = 0; pwk[k][0] = (uint64_t)(uint32_t)-1; }
  bool any_pwk = false;
  auto try_pwk = [&](int k, int off, int len, int limmax){
    if (!pwk_on || so.shape[k] != -1 || limmax < 0) return;
    uint64_t tmp[HS_PWK_SLOTS];
    if (piecewise_k(off, len, limmax, tmp)){ memcpy(pwk[k], tmp, sizeof tmp); so.shape[k] = HS_SHAPE_PWK; any_pwk = true; }
  };
  so.shape[HS_MAXREP] = classify(so.ins_off, so.ins_len, B);
  if (so.shape[HS_MAXREP] < 0 && piecewise(so.ins_off, so.ins_len, B, pw[HS_MAXREP])) so.shape[HS_MAXREP] = HS_SHAPE_PIECEWISE;
  try_pwk(HS_MAXREP, so.ins_off, so.ins_len, B);
  HS_SOLAP(1);
  // deletion visiting lists (StutterAlignerClass.cpp:123-142); one per deletion size, shift = |D|
  for (int q = 0; q < HS_MAXREP; q++){
    const int D = -(q+1)*period;
    so.del_off[q] = (int32_t)(visits_begin + V.size());
    if (B+D < 0){ so.del_len[q] = 0; so.shape[q] = -1; continue; }
    const int* up = upstream_runs(blk, B, -D);
    int i = 0;
    for (;;){
      const int ni = -i;
      if (ni >= B+D){ V.push_back(visit(ni, 0, 0, 0, true, 0)); break; }
      const int U = up[B-1+i];
      if (U == 0){ V.push_back(visit(ni, 0, blk[B-1+i+D], blk[B-1+i], false, 0)); i -= 1; }
      else       { V.push_back(visit(ni, U, 0, 0, false, T.int_log[U])); i -= U; }
    }
    so.del_len[q] = (int32_t)(visits_begin + V.size()) - so.del_off[q];
    so.shape[q] = classify(so.del_off[q], so.del_len[q], B+D);
    if (so.shape[q] < 0 && piecewise(so.del_off[q], so.del_len[q], B+D, pw[q])) so.shape[q] = HS_SHAPE_PIECEWISE;
    try_pwk(q, so.del_off[q], so.del_len[q], B+D);
  }
  HS_SOLAP(2);
  // The forward kernels read a visiting list only where it has no closed form (shape -1: replayed entry by entry; the traceback also
  // replays the piecewise-simple ones).  An option whose lists all have one — periodic and, for the forward pass, once-or-twice
  // interrupted blocks: nearly all — keeps none of them (they were 20 % of a batch's table bytes).
  {
    bool any_generic = false;
    for (int k = 0; k <= HS_MAXREP; k++){
      const int tail = (k == HS_MAXREP) ? B : B - (k+1)*period;
      any_generic |= (tail >= 0 && (so.shape[k] == -1 || so.shape[k] == HS_SHAPE_PWK || (!forward_only && so.shape[k] < 0)));
    }
    if (any_generic) out.visits.insert(out.visits.end(), V.begin(), V.end());
    else {
      so.ins_off = (int32_t)visits_begin; so.ins_len = 0;
      for (int q = 0; q < HS_MAXREP; q++){ so.del_off[q] = (int32_t)visits_begin; so.del_len[q] = 0; }
    }
  }
  // the descriptor slots are read only where a shape says "piecewise" (hs_str_kernel_generic): 560 bytes that the periodic blocks —
  // nearly all of them — do not need
  bool any_pw = false, any_replay = false;
  for (int k = 0; k <= HS_MAXREP; k++){
    const int tail = (k == HS_MAXREP) ? B : B - (k+1)*period;
    any_pw |= (so.shape[k] == HS_SHAPE_PIECEWISE);
    any_replay |= (tail >= 0 && (so.shape[k] == -1 || so.shape[k] == HS_SHAPE_PWK));
  }
  // (round 4: hs_str_group_kernel_pw also replays the lists that have no closed form — three and more interruptions — so an option with
  //  such lists stays in the grouped layout (kind 2) as long as its block is made of A/C/G/T; before round 4 they went to
  //  hs_str_kernel_generic)
  if (any_pw || (any_replay && group_replay)){
    const size_t at = out.f64pool.size();
    out.f64pool.resize(at + (HS_MAXREP + 1)*HS_PW_SLOTS);
    memcpy(out.f64pool.data() + at, pw, sizeof pw);
  }
  if (any_pwk){                                       // (any_pwk implies any_replay: the ten-slot region is in front)
    const size_t at = out.f64pool.size();
    out.f64pool.resize(at + (HS_MAXREP + 1)*HS_PWK_SLOTS);
    memcpy(out.f64pool.data() + at, pwk, sizeof pwk);
  }
  HS_SOLAP(3);
  // tabulated closed form: only when every list the kernel can evaluate is simple and the entries fit the LDS budget
  so.tab_off = out.f64pool.size(); so.tab_len = 0;
  {
    constexpr bool no_pw_group = false;
    bool ok = true; int total = 0;
    for (int i = 0; i < B; i++){ const char c = blk[i]; ok &= (c == 'A' || c == 'C' || c == 'G' || c == 'T'); }      // hs_str_group_kernel looks emissions up by base code
    ok &= (B >= period);                                                          // ... and lets ins_probs_ cycle through block bases only
    ok &= (B <= HS_GRP_MAX_BLOCK);                                                // ... and fetches the block four bases per lane of its 256-lane workgroup
    for (int k = 0; k <= HS_MAXREP && ok; k++){
      const int tail = (k == HS_MAXREP) ? B : B - (k+1)*period;
      so.tab_base[k] = total;
      if (tail < 0) continue;                           // this deletion size is never evaluated
      if (so.shape[k] == HS_SHAPE_PIECEWISE && !no_pw_group) continue;      // evaluated from its descriptor slots (hs_str_group_kernel_pw)
      if ((so.shape[k] == -1 || so.shape[k] == HS_SHAPE_PWK) && group_replay && !no_pw_group) continue;      // replayed in the grouped layout (visit_eval_grp) or taken by the K-level closed form
      if (so.shape[k] < 0){ ok = false; break; }
      total += 2 + std::max(0, tail - so.shape[k]);
    }
    if (ok && total <= HS_TAB_CAP){
      so.kind = any_replay ? 3 : (any_pw ? 2 : 1);
      out.f64pool.resize((size_t)so.tab_off + 3*(size_t)total + 1);
      double* ent = out.f64pool.data() + so.tab_off;
      double bmin = 1e300;                              // what the kernel compares |lp0| with: the weakest guarantee of the table
      for (int k = 0; k <= HS_MAXREP; k++){
        const int tail = (k == HS_MAXREP) ? B : B - (k+1)*period;
        if (tail < 0 || so.shape[k] < 0) continue;
        const int U0 = so.shape[k], n = 2 + std::max(0, tail - U0);
        for (int e = 0; e < n; e++, ent += 3){
          const int lim = (e == 0) ? 0 : (e == 1 ? 1 : U0 + e - 1);      // a bound that maps to entry e (entry 1 is unused when U0 = 0)
          simple_table_entry(lim, U0, tail, ent);
          bmin = std::min(bmin, ent[2]);
        }
      }
      so.tab_len = total;
      *ent = bmin;
    }
  }
  out.stropts.push_back(so);
  HS_SOLAP(4);
}

// Everything emit_stropt derives from (block, period, stutter model) — visiting lists, their shapes, the closed-form table, the 20
// constants — is a function of those three alone, and the same STR options come back all the time: every round of
// SeqStutterGenotyper::genotype() re-sends a locus with one or two alleles more (seq_stutter_genotyper.cpp:603-671), both sides of
// a palindromic motif, neighbouring loci of a panel with the same motif.  A per-thread cache keyed by the three holds the option
// record with its pool slices relative to their start; a hit appends copies (about 2 KB) instead of rebuilding them.
struct StroptCached { hs_stropt_t so; std::vector<hs_visit_t> visits; std::vector<double> f64; std::vector<char> chars; };
void emit_stropt_cached(const char* blk, int B, int period, const double* stutter, const double pmf13[HS_NART], Prepared& out, int pmf_off, int twin){
  // (off by default since round 4: building an option now costs about what a hit did — a hash of ~100 bytes and 2 KB of copies;
  //  the cache stays in the source for a caller that re-sends loci, switched off)
  constexpr bool on = false;
  if (!on || g_bnd_scale.load() != 1.0){ emit_stropt(blk, B, period, pmf13, out, true, pmf_off, twin); return; }      // (the cache keeps host-written tables only)
  thread_local std::unordered_map<std::string, StroptCached> cache;
  thread_local std::string key;
  key.assign((const char*)stutter, 6*sizeof(double)); key.push_back((char)period); key.append(blk, B);
  auto it = cache.find(key);
  if (it == cache.end()){
    const size_t v0 = out.visits.size(), f0 = out.f64pool.size(), c0 = out.chars.size();
    emit_stropt(blk, B, period, pmf13, out, true);
    if (cache.size() >= 8192) cache.clear();
    StroptCached& e = cache[key];
    e.so = out.stropts.back();
    e.visits.assign(out.visits.begin() + v0, out.visits.end()); e.f64.assign(out.f64pool.begin() + f0, out.f64pool.end());
    e.chars.assign(out.chars.begin() + c0, out.chars.end());
    e.so.seq_off -= (int32_t)c0; e.so.f64_off -= (int32_t)f0; e.so.tab_off -= (int32_t)f0; e.so.ins_off -= (int32_t)v0;
    for (int q = 0; q < HS_MAXREP; q++) e.so.del_off[q] -= (int32_t)v0;
    return;
  }
  const StroptCached& e = it->second;
  const int32_t v0 = (int32_t)out.visits.size(), f0 = (int32_t)out.f64pool.size(), c0 = (int32_t)out.chars.size();
  out.visits.insert(out.visits.end(), e.visits.begin(), e.visits.end());
  out.f64pool.insert(out.f64pool.end(), e.f64.begin(), e.f64.end());
  out.chars.insert(out.chars.end(), e.chars.begin(), e.chars.end());
  hs_stropt_t so = e.so;
  so.seq_off += c0; so.f64_off += f0; so.tab_off += f0; so.ins_off += v0;
  for (int q = 0; q < HS_MAXREP; q++) so.del_off[q] += v0;
  out.stropts.push_back(so);
}

}  // namespace

// ---- helpers shared with the traceback path (trace.hip)
void fresh_flank_rows(const std::string side_seqs[3], std::vector<hs_row_t>& lead, std::vector<hs_row_t>& trail){
  SideSeqs h;
  for (int j = 0; j < 3; j++) h.s[j] = side_seqs[j];
  h.index();
  lead = flank_rows(h, 0, 0);
  trail = flank_rows(h, 2, (int)h.s[0].size() + 1);
}
void append_stropt(const std::string& blk, int period, const double* stutter, Prepared& out){
  double pmf13[HS_NART]; stutter_pmf13(stutter, period, pmf13);
  emit_stropt(blk.data(), (int)blk.size(), period, pmf13, out);
}
void debug_simple_table(int lim, int U0, int tail, double ent[3]){ g_bnd_scale = 1.0; simple_table_entry(lim, U0, tail, ent); }

// The caller's offset tables, before anything indexes with them: the boundary takes plain pointers, so what the library can know is
// whether the tables are consistent with EACH OTHER — null pointers, counts that contradict each other (a locus' haplotype count is the
// product of its blocks' option counts), offsets that are negative or decrease, CIGAR runs of non-positive length.  One pass over the
// tables (a few integers per read); everything behind it (check_locus, prepare_locus) may then take lengths as differences of
// neighbouring offsets without looking again.  (tools/fuzz_malformed.py: a random corruption per case.)
// What is NOT here: the sizes a consistent locus may have and this library does not take — more than 1024 options of a
// block, an option of more than 65536 bases, a read of more than 1 Mi bases.  Those are check_locus' refusals, per locus: one oversized
// locus in a shard fails alone (hipstr_hmm_process_reads_each, hipstr_stream_submit_each), not the call.
int validate_tables(const hipstr_batch_t* b, std::string& err){
  if (b == NULL || b->n_loci < 0){ err = "null or negative-size batch"; return 1; }
  const int nl = b->n_loci;
  if (nl == 0) return 0;
  if (!b->blk_start || !b->blk_end || !b->blk_nopts || !b->period || !b->stutter || !b->opt_off || !b->seq || !b->hap_off || !b->read_off ||
      !b->base_off || !b->bases || !b->quals || !b->read_start || !b->cigar_off || !b->cigar_op || !b->cigar_len){ err = "null table in a batch with loci"; return 1; }
  if (b->hap_off[0] < 0 || b->read_off[0] < 0 || b->opt_off[0] < 0){ err = "negative offset"; return 1; }
  int64_t nopt = 0;
  for (int l = 0; l < nl; l++){
    if (b->read_off[l+1] < b->read_off[l]){ err = "read_off must not decrease"; return 1; }
    if (b->hap_off[l+1] < b->hap_off[l]){ err = "hap_off must not decrease"; return 1; }
    // the option counts are what the walk over opt_off below trusts: they must agree with the haplotype count the caller states
    // separately (a count no caller means contradicts it; the walk never leaves the caller's table on one corrupted integer)
    int64_t A = 1;
    for (int k = 0; k < 3; k++){
      const int n = b->blk_nopts[3*l+k];
      if (n < 1){ err = "haplotype block without options"; return 1; }
      A *= n;                                        // (kept below 2^32 by the test that follows: the next factor cannot wrap it)
      if (A > ((int64_t)1 << 31)){ err = "hap_off does not match the product of block options"; return 1; }
      nopt += n;
    }
    if (A != (int64_t)b->hap_off[l+1] - b->hap_off[l]){ err = "hap_off does not match the product of block options"; return 1; }
  }
  for (int64_t o = 0; o < nopt; o++)
    if (b->opt_off[o+1] < b->opt_off[o]){ err = "opt_off must not decrease"; return 1; }
  const int nr = b->read_off[nl];
  if (nr > 0 && (b->base_off[b->read_off[0]] < 0 || b->cigar_off[b->read_off[0]] < 0)){ err = "negative offset"; return 1; }
  for (int r = b->read_off[0]; r < nr; r++){
    if (b->base_off[r+1] < b->base_off[r]){ err = "base_off must not decrease"; return 1; }
    if (b->cigar_off[r+1] < b->cigar_off[r]){ err = "cigar_off must not decrease"; return 1; }
  }
  // (a CIGAR table no caller means — more runs than bases — is an offset pointing out of the caller's memory: the walk below stays
  // within runs that can belong to the batch's bases)
  const int64_t nbases = nr > 0 ? (int64_t)b->base_off[nr] - b->base_off[b->read_off[0]] : 0;
  const int c0 = nr > 0 ? b->cigar_off[b->read_off[0]] : 0, nc = nr > 0 ? b->cigar_off[nr] : 0;
  if ((int64_t)nc - c0 > 2*nbases + 2*(int64_t)(nr - b->read_off[0])){ err = "cigar_off holds more runs than the reads have bases"; return 1; }
  for (int c = c0; c < nc; c++) if (b->cigar_len[c] <= 0){ err = "CIGAR run of non-positive length"; return 1; }
  return 0;
}

// The conditions under which prepare_batch refuses a batch, without building anything (same order, same messages): used where a
// bad locus must be turned away before it is merged with others (hipstr_stream_submit).
int check_batch(const hipstr_batch_t* b, std::string& err){
  if (validate_tables(b, err)) return 1;
  int opt_cursor = 0;
  for (int l = 0; l < b->n_loci; l++)
    if (check_locus(b, l, &opt_cursor, err)) return 1;
  return 0;
}

double locus_cost(const hipstr_batch_t* b, int l, int opt0){
  const int n0 = std::max(0, b->blk_nopts[3*l]), n1 = std::max(0, b->blk_nopts[3*l+1]), n2 = std::max(0, b->blk_nopts[3*l+2]);
  const int p = std::max(1, b->period[l]);
  auto len_of = [&](int c){ return (double)(b->opt_off[c+1] - b->opt_off[c]); };
  double f0 = 0, f2 = 0, str = 0;
  for (int o = 0; o < n0; o++) f0 += len_of(opt0 + o);
  for (int o = 0; o < n2; o++) f2 += len_of(opt0 + n0 + n1 + o);
  for (int o = 0; o < n1; o++){
    const int c = opt0 + n0 + o;
    const char* sq = b->seq + b->opt_off[c]; const int B = b->opt_off[c+1] - b->opt_off[c];
    int brk = 0;
    for (int i = p; i < B; i++) brk += (sq[i] != sq[i-p]) ? 1 : 0;        // a substituted base breaks the period twice
    str += 1.0 + 2.4*(0.5*brk);
  }
  const double flank = (n0 ? f0/n0 : 0.0) + (n2 ? f2/n2 : 0.0), str_mean = n1 ? str/n1 : 1.0;
  const int r0 = b->read_off[l], r1 = b->read_off[l+1];
  int64_t P = 0, bases = 0;
  for (int r = r0; r < r1; r++){
    if (b->realign_read && !b->realign_read[r]) continue;
    P++; bases += b->base_off[r+1] - b->base_off[r];
  }
  int64_t A = 0;
  const int h0 = b->hap_off[l], h1 = b->hap_off[l+1];
  if (b->realign_hap){ for (int h = h0; h < h1; h++) A += b->realign_hap[h] ? 1 : 0; } else A = h1 - h0;
  if (P == 0 || A == 0) return 1e-3;
  return (double)A * ((double)bases/150.0) * (1.4*flank/60.0 + str_mean);
}

// One locus of check_batch; *opt_cursor = index of the locus' first block option in opt_off, advanced past the locus.
int check_locus(const hipstr_batch_t* b, int l, int* opt_cursor_io, std::string& err, int32_t* seeds_out, int* dims_out){
  int opt_cursor = *opt_cursor_io;
  {
    const int period = b->period[l];
    if (period < 1 || period > 9){ err = "STR period must be in [1,9] (stutter_model.h:38)"; return 1; }
    int64_t A = 1;
    for (int k = 0; k < 3; k++){
      const int n = b->blk_nopts[3*l+k];
      if (n < 1){ err = "haplotype block without options"; return 1; }
      if (n > 1024){ err = "more than 1024 options for a haplotype block are not supported"; return 1; }
      A *= n;
      for (int o = 0; o < n; o++, opt_cursor++){
        const int len = b->opt_off[opt_cursor+1] - b->opt_off[opt_cursor];
        if (len < 0){ err = "opt_off must not decrease"; return 1; }
        if (len > (1 << 16)){ err = "haplotype block option longer than 65536 bases is not supported"; return 1; }
        if (k != 1 && len == 0){ err = "empty flank sequence"; return 1; }
        if (k == 1 && len == 0){ err = "empty STR allele is not supported"; return 1; }
        if (k == 1 && len > HS_MAX_STR_BP){ err = "STR allele longer than 2047 bp is not supported"; return 1; }
      }
    }
    if (A != b->hap_off[l+1]-b->hap_off[l]){ err = "hap_off does not match the product of block options"; return 1; }
    if (A >= (1 << 24)){ err = "more than 16 M candidate haplotypes for a locus are not supported"; return 1; }
    if (b->read_off[l+1] < b->read_off[l]){ err = "read_off must not decrease"; return 1; }
    int longest_B = 1, longest_read = 0;
    for (int o = 0; o < b->blk_nopts[3*l+1]; o++){ const int c = *opt_cursor_io + b->blk_nopts[3*l] + o; longest_B = std::max(longest_B, b->opt_off[c+1] - b->opt_off[c]); }
    for (int r = b->read_off[l]; r < b->read_off[l+1]; r++){
      if (b->realign_read && !b->realign_read[r]){ if (seeds_out) seeds_out[r] = HIPSTR_SEED_AUTO; continue; }
      const int len = b->base_off[r+1] - b->base_off[r];
      if (len > (1 << 20)){ err = "read longer than 1 Mi bases is not supported"; return 1; }
      if (b->cigar_off[r+1] - b->cigar_off[r] > (1 << 20)){ err = "CIGAR of more than 1 Mi runs is not supported"; return 1; }
      longest_read = std::max(longest_read, len);
      const int s = calc_seed_base(b, l, r);
      if (seeds_out) seeds_out[r] = s;
      if (s == -2){ err = "Invalid alignment seed or unrecognized CIGAR char (HapAligner.cpp:309,316)"; return 1; }
      if (s >= 0 && (s > HS_MAX_SIDE_FWD || len-s-1 > HS_MAX_SIDE_FWD)){ err = "read side longer than 1024 bases is not supported"; return 1; }
    }
    // the per-read STR kernels keep a read's tables and the allele's block in LDS: a locus whose longest read and longest allele do not fit
    // is turned away here, alone — not at upload time, where it would take the batch it shares down with it.  (The LDS is sized by the
    // BATCH's longest read and longest allele: the stream and hipstr_hmm_process_reads_each close a batch before a locus that would push
    // the combined figure over the limit — dims_out.)
    if (dims_out){ dims_out[0] = longest_read; dims_out[1] = longest_B; }
    if (hs_str_kernel_lds_bytes(longest_read, longest_B) > HS_LDS_LIMIT){ err = "reads and STR alleles this long need more than 160 KiB of LDS per workgroup (about 1.8 kb reads; less with alleles near 2 kb)"; return 1; }
  }
  *opt_cursor_io = opt_cursor;
  return 0;
}

// Everything prepare_batch derives from ONE locus, appended to `out` — a fragment holding a run of consecutive loci whose pool
// offsets are local to the fragment (merge_fragment rebases them).  Per-read records go straight to the batch-wide arrays in
// `sh` (disjoint ranges per locus), so fragments can be built by different threads.
struct PrepShared {
  hs_read_t* reads; int32_t* seeds; uint8_t* realign_read;
  const int64_t* out_off;          // [n_loci] start of every locus' block in aln_probs
  const int32_t* seed_in;
};

// HIPSTR_PREP_PROFILE=1: cycles per section of prepare_locus, summed over threads, printed by hipstr_debug_prepare
static std::atomic<uint64_t> g_lap_cycles[8];
static const bool g_lap_on = getenv("HIPSTR_PREP_PROFILE") != NULL;
void prep_profile_print(){
  if (!g_lap_on) return;
  static const char* const names[8] = { "options + strings", "STR options (emit_stropt_cached)", "boundary signatures", "allele loop (flank rows, reuse replay)", "STR order + records + read-end rows", "trailing-flank groups", "reads + seeds", "" };
  uint64_t tot = 0; for (int i = 0; i < 7; i++) tot += g_lap_cycles[i];
  for (int i = 0; i < 7; i++){ fprintf(stderr, "prepare_locus: %-40s %6.1f %%\n", names[i], tot ? 100.0*g_lap_cycles[i]/tot : 0.0); g_lap_cycles[i] = 0; }
  static const char* const so_names[5] = { "constants (stutter pmf, priors)", "insertion list", "deletion lists", "list bookkeeping + descriptor slots", "closed-form table" };
  uint64_t st = 0; for (int i = 0; i < 5; i++) st += g_so_cycles[i];
  for (int i = 0; i < 5; i++){ fprintf(stderr, "  emit_stropt: %-38s %6.1f %%\n", so_names[i], st ? 100.0*g_so_cycles[i]/st : 0.0); g_so_cycles[i] = 0; }
}
#define HS_LAP(k) do { if (g_lap_on){ const uint64_t now_ = __builtin_ia32_rdtsc(); g_lap_cycles[k] += now_ - lap_t; lap_t = now_; } } while (0)

namespace {
struct Seq { const char* p; int n; };
inline bool seq_eq(const Seq& a, const Seq& b){ return a.n == b.n && memcmp(a.p, b.p, (size_t)a.n) == 0; }
inline bool seq_less(const Seq& a, const Seq& b){        // std::string's order among sequences of one length
  return memcmp(a.p, b.p, (size_t)a.n) < 0;
}
// stable insertion sort for the short lists of a locus (options, reads of a side); longer ones take std::stable_sort
template <class T, class Less> void stable_small_sort(T* v, int n, Less less){
  if (n > 48){ std::stable_sort(v, v + n, less); return; }
  for (int i = 1; i < n; i++){
    T x = v[i]; int j = i;
    while (j > 0 && less(x, v[j-1])){ v[j] = v[j-1]; j--; }
    v[j] = x;
  }
}
// Scratch of prepare_locus, one per host thread (cleared per locus, capacity kept)
struct LocusScratch {
  std::vector<char> rev;                       // reversed STR options, then (per cache miss) the reversed flanks of one haplotype
  std::vector<char> hbuf;
  std::vector<Seq> sblk[2];                    // STR options in side orientation
  std::vector<int32_t> str_opt_of, ks, opt_rank, os, cnt;
  struct EndSig { char c_first, c_last; int run_first, run_last; };
  struct RowsKey { int opt; char c; int run; int aux; int id; };
  std::vector<EndSig> end_sig[2];
  std::vector<RowsKey> lead_tab[2], trail_tab[2];
  std::vector<int32_t> lead_sets[2];
  std::vector<int> run;                        // lrun | rrun of the three blocks of one orientation
  std::vector<int> hl;
  std::vector<hs_row_t> rows;
  std::vector<uint64_t> pairs;
};
static LocusScratch& locus_scratch(){ thread_local LocusScratch s; return s; }

// One haplotype orientation: the three block sequences + HapBlock's run-length tables (see SideSeqs above; same quirk)
struct SideView {
  Seq s[3];
  const int* lrun[3]; const int* rrun[3];
};
static void index_side(SideView& h, std::vector<int>& store){
  size_t tot = 0; for (int b = 0; b < 3; b++) tot += 2*(size_t)h.s[b].n;
  if (store.size() < tot) store.resize(tot);
  int* p = store.data();
  for (int b = 0; b < 3; b++){
    const int n = h.s[b].n; const char* q = h.s[b].p;
    int* lr = p; int* rr = p + n; p += 2*n;
    h.lrun[b] = lr; h.rrun[b] = rr;
    if (n == 0) continue;
    for (int j = 0; j < n; j++){ lr[j] = 0; rr[j] = 0; }
    int count = 0;
    for (int j = 1; j < n; j++){ count = (q[j-1] == q[j]) ? count+1 : 0; lr[j] = count; }
    for (int j = n-2; j >= 0; j--){ count = (q[j+1] == q[j]) ? count+1 : 0; rr[j] = count; }
  }
}
// Haplotype::homopolymer_length with its neighbour-block extensions (Haplotype.cpp:239-287)
static int homopolymer_len_v(const SideView& h, int bi, int pos){
  const Seq& q = h.s[bi];
  const char c = q.p[pos];
  int l = h.lrun[bi][pos], r = h.rrun[bi][pos];
  if (pos - l == 0){
    for (int nb = bi-1; nb >= 0; nb--){
      const int n = h.s[nb].n;
      if (n == 0) continue;
      if (h.s[nb].p[n-1] != c) break;
      const int ll = h.lrun[nb][n-1];
      l += 1 + ll;
      if (ll != n) break;
    }
  }
  if (pos + r == q.n-1){
    for (int nb = bi+1; nb < 3; nb++){
      const int n = h.s[nb].n;
      if (n == 0) continue;
      if (h.s[nb].p[0] != c) break;
      const int rl = h.rrun[nb][0];
      r += 1 + rl;
      if (rl != n) break;
    }
  }
  return l + r + 1;
}
// rows of flank block `bi` (0 = lead, 2 = trail) under context h; u0 = compact index of its first row
static void flank_rows_v(const SideView& h, int bi, int u0, std::vector<int>& hl, std::vector<hs_row_t>& rows){
  const Seq& q = h.s[bi];
  if ((int)hl.size() < q.n) hl.resize(q.n);
  rows.resize(q.n);
  for (int i = 0; i < q.n; i++) hl[i] = homopolymer_len_v(h, bi, i);
  for (int i = 0; i < q.n; i++){
    const int v = std::min(HIPSTR_MAX_HOMOP_LEN, std::max(hl[i], hl[std::max(0, i-1)]));
    rows[i] = HS_ROW_VALID | ((uint32_t)(u0+i) << 12) | ((uint32_t)v << 8) | (uint8_t)q.p[i];
  }
}
}  // namespace

static int prepare_locus(const hipstr_batch_t* b, int l, int opt_cursor, const PrepShared& sh, Prepared& out, std::string& err){
  const int32_t* seed_in = sh.seed_in;
  uint64_t lap_t = g_lap_on ? __builtin_ia32_rdtsc() : 0;
  LocusScratch& S = locus_scratch();
  const int period = b->period[l];
  if (period < 1 || period > 9){ err = "STR period must be in [1,9] (stutter_model.h:38)"; return 1; }
  int32_t nopts[3], opt_first[3];
  for (int k = 0; k < 3; k++){
    nopts[k] = b->blk_nopts[3*l+k];
    if (nopts[k] < 1){ err = "haplotype block without options"; return 1; }
    if (nopts[k] > 1024){ err = "more than 1024 options for a haplotype block are not supported"; return 1; }      // (the sizes check_locus refuses, in its words)
    opt_first[k] = opt_cursor; opt_cursor += nopts[k];
  }
  auto O = [&](int k, int o) -> Seq { const int c = opt_first[k] + o; return Seq{ b->seq + b->opt_off[c], b->opt_off[c+1] - b->opt_off[c] }; };
  for (int k = 0; k < 3; k += 2)
    for (int o = 0; o < nopts[k]; o++){
      if (O(k, o).n == 0){ err = "empty flank sequence"; return 1; }
      if (O(k, o).n > (1 << 16)){ err = "haplotype block option longer than 65536 bases is not supported"; return 1; }
    }
  for (int r = b->read_off[l]; r < b->read_off[l+1]; r++){
    if (b->base_off[r+1] - b->base_off[r] > (1 << 20)){ err = "read longer than 1 Mi bases is not supported"; return 1; }
    if (b->cigar_off[r+1] - b->cigar_off[r] > (1 << 20)){ err = "CIGAR of more than 1 Mi runs is not supported"; return 1; }
  }
  size_t str_total = 0;
  for (int o = 0; o < nopts[1]; o++){
    const int n = O(1, o).n;
    if (n == 0){ err = "empty STR allele is not supported"; return 1; }
    if (n > HS_MAX_STR_BP){ err = "STR allele longer than 2047 bp is not supported"; return 1; }
    out.max_B = std::max(out.max_B, (int32_t)n);
    str_total += (size_t)n;
  }
  const int64_t A64 = (int64_t)nopts[0]*nopts[1]*nopts[2];
  if (A64 != b->hap_off[l+1]-b->hap_off[l]){ err = "hap_off does not match the product of block options"; return 1; }
  if (A64 >= (1 << 24)){ err = "more than 16 M candidate haplotypes for a locus are not supported"; return 1; }
  const int A = (int)A64;

  HS_LAP(0);
  hs_locus_t loc;
  loc.out_off = sh.out_off[l]; loc.hap_begin = out.alleles.size(); loc.n_alleles = A;
  loc.read_begin = b->read_off[l]; loc.n_reads = b->read_off[l+1]-b->read_off[l];
  loc.period = period; loc.fused = 0;

  // STR options: forward then reversed orientation
  const int so_base = out.stropts.size();
  double pmf13[HS_NART]; stutter_pmf13(b->stutter + 6*l, period, pmf13);
  const int pmf_off = (int)out.pmf13.size();
  out.pmf13.insert(out.pmf13.end(), pmf13, pmf13 + HS_NART);
  S.rev.resize(str_total);
  S.sblk[0].clear(); S.sblk[1].clear();
  {
    char* rp = S.rev.data();
    for (int o = 0; o < nopts[1]; o++){
      const Seq f = O(1, o);
      S.sblk[0].push_back(f);
      for (int i = 0; i < f.n; i++) rp[i] = f.p[f.n-1-i];
      S.sblk[1].push_back(Seq{rp, f.n});
      rp += f.n;
    }
  }
  for (int side = 0; side < 2; side++)
    for (int o = 0; o < nopts[1]; o++)
      emit_stropt_cached(S.sblk[side][o].p, S.sblk[side][o].n, period, b->stutter + 6*l, pmf13, out, pmf_off, side ? so_base + o : -1);
  HS_LAP(1);
  S.str_opt_of.resize(A);                            // STR option of every allele
  const bool one_flank = nopts[0] == 1 && nopts[2] == 1;          // (the usual locus: allele k is STR option k, the STR block is what changes)
  if (one_flank) for (int k = 0; k < A; k++) S.str_opt_of[k] = k;
  else for (int k = 0; k < A; k++){ int32_t o3[3]; allele_options(nopts, k, o3); S.str_opt_of[k] = o3[1]; }

  // alleles in visit order, replaying the reference's row reuse (HapAligner.cpp:54-60, 612-634):
  // the lead block of a side is (re)computed only when reuse is off or it is the block that changed;
  // otherwise its rows — and the homopolymer context they were computed under — are inherited.
  const int rowset_first = (int)out.rowsets.size();           // rowsets are shared (by content) within the locus
  auto intern = [&](const std::vector<hs_row_t>& rows){
    const int n = (int)rows.size();
    // (the reference-order map of round 1-3 numbered distinct row vectors in order of first use: so does a scan of the locus' rowsets)
    for (int id = rowset_first; id < (int)out.rowsets.size(); id++){
      const hs_rowset_t& rs = out.rowsets[id];
      if (rs.len == n && memcmp(out.rows.data() + rs.off, rows.data(), sizeof(hs_row_t)*(size_t)n) == 0) return id;
    }
    hs_rowset_t rs; rs.off = out.rows.size(); rs.len = n;
    out.rows.insert(out.rows.end(), rows.begin(), rows.end());
    out.rowsets.push_back(rs);
    return (int)out.rowsets.size() - 1;
  };
  // boundary signature of every STR option per side: first base and HapBlock's right-run length there, last base and left-run length
  // there (HapBlock.cpp:7-30, with its counter carried from the forward into the backward pass)
  typedef LocusScratch::EndSig EndSig; typedef LocusScratch::RowsKey RowsKey;
  for (int side = 0; side < 2; side++){
    S.end_sig[side].clear(); S.lead_tab[side].clear(); S.trail_tab[side].clear(); S.lead_sets[side].clear();
    for (int o = 0; o < nopts[1]; o++){
      const Seq& q = S.sblk[side][o];
      const int n = q.n;
      // HapBlock's two passes (forward: count = run so far; backward: the same counter, NOT reset) leave, at the block's ends: the run
      // ending at the last base and the run starting at the first one, each minus one — unless the block is one run, where the backward pass
      // keeps counting from what the forward pass left: n - 1 and 2 (n - 1)
      int run_first = 0, run_last = 0;
      while (run_first < n-1 && q.p[run_first+1] == q.p[0]) run_first++;
      while (run_last < n-1 && q.p[n-2-run_last] == q.p[n-1]) run_last++;
      const int lr_last = run_last, rr_first = (run_first == n-1) ? 2*(n-1) : run_first;
      S.end_sig[side].push_back(EndSig{ q.p[0], q.p[n-1], rr_first, lr_last });
    }
  }
  HS_LAP(2);
  bool reuse = false;
  int lead_id[2] = {-1, -1};
  int n_realigned = 0;
  loc.lt_stride = 0; loc.lead_flank[0] = loc.lead_flank[1] = 0;
  const size_t allele_base = out.alleles.size();
  out.alleles.resize(allele_base + (size_t)A);
  const size_t rh_base = out.realign_hap.size();
  out.realign_hap.resize(rh_base + (size_t)A);
  for (int k = 0; k < A; k++){
    const bool realign = b->realign_hap ? b->realign_hap[b->hap_off[l]+k] != 0 : true;
    out.realign_hap[rh_base + k] = realign ? 1 : 0;
    hs_allele_t al; memset(&al, 0, sizeof al);
    al.realign = realign ? 1 : 0;
    int32_t o3[3];
    if (one_flank){ o3[0] = 0; o3[1] = k; o3[2] = 0; } else allele_options(nopts, k, o3);
    al.n_flank = O(0, o3[0]).n + O(2, o3[2]).n;
    out.max_flank = std::max(out.max_flank, al.n_flank);
    if (!realign){ reuse = false; out.alleles[allele_base + k] = al; continue; }
    al.re_ord = n_realigned++;
    loc.lt_stride = std::max(loc.lt_stride, al.n_flank);
    const int cb = k == 0 ? -1 : (one_flank ? 1 : changed_block(nopts, k));
    for (int side = 0; side < 2; side++){
      // The rows of a flank block depend on the block and, through the homopolymer run that may cross the block boundary, on the
      // STR block's first base and the run it starts (leading flank) or its last base and the run it ends (trailing flank) — never on
      // more: the reference's extension stops after one non-empty neighbour (Haplotype.cpp:239-275).  The alleles of a locus mostly
      // share flanks and motif, so the rows are built once per (flank option, boundary signature) and looked up afterwards.
      const int o_lead = side ? o3[2] : o3[0], o_trail = side ? o3[0] : o3[2];
      const int lead_len = O(side ? 2 : 0, o_lead).n;
      const EndSig& es = S.end_sig[side][o3[1]];
      SideView h; bool have_h = false;
      auto need_h = [&](){
        if (have_h) return;
        if (!side){ h.s[0] = O(0, o3[0]); h.s[1] = S.sblk[0][o3[1]]; h.s[2] = O(2, o3[2]); }
        else {
          const Seq f0 = O(2, o3[2]), f2 = O(0, o3[0]);           // side order: [reversed right flank | reversed block | reversed left flank]
          S.hbuf.resize((size_t)f0.n + f2.n);
          char* p0 = S.hbuf.data(); char* p2 = p0 + f0.n;
          for (int i = 0; i < f0.n; i++) p0[i] = f0.p[f0.n-1-i];
          for (int i = 0; i < f2.n; i++) p2[i] = f2.p[f2.n-1-i];
          h.s[0] = Seq{p0, f0.n}; h.s[1] = S.sblk[1][o3[1]]; h.s[2] = Seq{p2, f2.n};
        }
        index_side(h, S.run); have_h = true;
      };
      auto cached = [&](std::vector<RowsKey>& tab, const RowsKey& key, int bi, int u0) -> int {
        for (const RowsKey& e : tab) if (e.opt == key.opt && e.c == key.c && e.run == key.run && e.aux == key.aux) return e.id;
        need_h();
        flank_rows_v(h, bi, u0, S.hl, S.rows);
        RowsKey e = key; e.id = intern(S.rows);
        tab.push_back(e);
        return e.id;
      };
      const int side_changed = cb < 0 ? -1 : (side ? 2-cb : cb);
      // (one case looks further: a neighbour that is ONE run whose table entry equals its length lets the extension go on into the block
      //  behind it (Haplotype.cpp:262-270: `if (rl != n) break`) — with HapBlock's carried counter that is a two-base block, 2 (n - 1) == n:
      //  a homopolymer locus' two-copy allele.  Its leading-flank rows also depend on the far flank: part of the key.)
      const int lead_aux = (es.run_first == S.sblk[side][o3[1]].n) ? 1 + o_trail : 0;
      if (!reuse || side_changed <= 0) lead_id[side] = cached(S.lead_tab[side], RowsKey{o_lead, es.c_first, es.run_first, lead_aux, 0}, 0, 0);
      al.lead_rows[side]  = lead_id[side];
      {
        std::vector<int32_t>& ls = S.lead_sets[side];
        size_t slot = std::find(ls.begin(), ls.end(), lead_id[side]) - ls.begin();
        if (slot == ls.size()) ls.push_back(lead_id[side]);
        al.lead_slot[side] = (int32_t)slot;
        loc.lead_flank[side] = std::max(loc.lead_flank[side], (int32_t)lead_len);
      }
      al.trail_rows[side] = cached(S.trail_tab[side], RowsKey{o_trail, es.c_last, es.run_last, lead_len, 0}, 2, lead_len + 1);
      al.str_opt[side]    = so_base + side*nopts[1] + o3[1];
    }
    reuse = true;
    out.alleles[allele_base + k] = al;
  }

  HS_LAP(3);
  loc.n_re = n_realigned;
  loc.n_lead[0] = S.lead_sets[0].size(); loc.n_lead[1] = S.lead_sets[1].size();
  for (int side = 0; side < 2; side++){
    // STR-kernel order: by STR option, options sorted by length; an option whose block (in side orientation) ends with the
    // previous option's block continues that option's match/deletion tables (StutterAlignerClass::load_read sums run
    // from the block's right end, so a longer block with the same tail only appends terms)
    const std::vector<Seq>& blk = S.sblk[side];
    const hs_stropt_t* so_side = out.stropts.data() + so_base + side*nopts[1];
    auto block_of = [&](int k) -> const Seq& { return blk[S.str_opt_of[k]]; };
    // alleles whose closed form is tabulated come first: they are the business of hs_str_kernel, the rest of hs_str_kernel_generic
    auto kind_of = [&](int k){ return so_side[S.str_opt_of[k]].kind; };
    auto tabbed = [&](int k){ return kind_of(k) == 1; };
    // (the STR options are ranked once — tabulated first, then by length, then by sequence — and the alleles sorted by their option's rank)
    S.opt_rank.resize(nopts[1]); S.os.resize(nopts[1]);
    for (int o = 0; o < nopts[1]; o++) S.os[o] = o;
    stable_small_sort(S.os.data(), nopts[1], [&](int x, int y){
      static const int kind_rank[4] = {3, 0, 1, 2};               // tabulated, then piecewise, then replayed in the grouped layout, then the rest
      const int tx = kind_rank[so_side[x].kind], ty = kind_rank[so_side[y].kind];
      if (tx != ty) return tx < ty;
      if (blk[x].n != blk[y].n) return blk[x].n < blk[y].n;
      return seq_less(blk[x], blk[y]);
    });
    int n_rank = 0;
    for (int i = 0; i < nopts[1]; i++){
      if (i > 0 && !seq_eq(blk[S.os[i]], blk[S.os[i-1]])) n_rank++;        // equal sequences compare equal, as before
      S.opt_rank[S.os[i]] = n_rank;
    }
    n_rank++;
    // the realigned alleles sorted by their option's rank, ties in allele order (a counting sort)
    S.cnt.assign((size_t)n_rank + 1, 0);
    for (int k = 0; k < A; k++) if (out.alleles[allele_base + k].realign) S.cnt[S.opt_rank[S.str_opt_of[k]] + 1]++;
    for (int r = 0; r < n_rank; r++) S.cnt[r+1] += S.cnt[r];
    S.ks.resize(n_realigned);
    for (int k = 0; k < A; k++) if (out.alleles[allele_base + k].realign) S.ks[S.cnt[S.opt_rank[S.str_opt_of[k]]]++] = k;
    const std::vector<int32_t>& ks = S.ks;
    loc.order_off[side] = out.str_order.size();
    loc.n_tab[side] = 0; loc.n_short[side] = 0; loc.n_pw[side] = 0; loc.n_rp[side] = 0;
    loc.rec_off[side] = (int32_t)out.rec_descs.size();
    loc.ndrow_off[side] = (int32_t)out.nd_rows.size(); loc.n_ndrows[side] = 0;
    int fam_row0 = 0, fam_k = 0;                     // first row of the current family of alleles (blocks growing by one repeat unit), position in it
    Seq prev{NULL, 0};
    for (size_t i = 0; i < ks.size(); i++){
      const Seq& cur = block_of(ks[i]);
      const bool first_of_kind = (i == 0) || (kind_of(ks[i]) != kind_of(ks[i-1]));
      const bool chained = !first_of_kind && cur.n >= prev.n && memcmp(cur.p + (cur.n - prev.n), prev.p, (size_t)prev.n) == 0;
      // bit 29: a tabulated (hence periodic) block that extends the previous one by exactly one repeat unit: the read-end deletion
      // sums of the previous allele move up one size (hs_str_kernel)
      const bool one_unit = chained && tabbed(ks[i]) && cur.n == prev.n + period;
      out.str_order.push_back(ks[i] | (chained ? (1 << 30) : 0) | (one_unit ? (1 << 29) : 0));
      if (tabbed(ks[i])){
        loc.n_tab[side]++;
        {   // the position's record for hs_str_group_kernel_p (layout.h): described here, assembled on the device (hs_expand_recs_kernel)
          const hs_allele_t& al = out.alleles[allele_base + ks[i]];
          const hs_stropt_t& so = out.stropts[al.str_opt[side]];
          hs_recdesc_t rdsc;
          rdsc.stropt = al.str_opt[side]; rdsc.re_ord = al.re_ord; rdsc.nd_row = 0;
          rdsc.flags = (al.lead_slot[side] & 0x3ff) | (chained ? (1 << 30) : 0) | (one_unit ? (1 << 29) : 0);
          if (period <= HS_GRP_MAXP){
            // rows of read-end deletion sums (layout.h hs_ndrow_t): a new family opens with five rows for the first allele's larger sizes
            if (!one_unit){
              fam_row0 = loc.n_ndrows[side]; fam_k = 0;
              for (int m = 0; m < HS_MAXREP - 1; m++) out.nd_rows.push_back(hs_ndrow_t{ so.B + (m - HS_MAXREP)*period, so.tail_codes });
              loc.n_ndrows[side] += HS_MAXREP - 1;
            } else fam_k++;
            out.nd_rows.push_back(hs_ndrow_t{ so.B - period, so.tail_codes });       // row fam_k + 5: this allele's size 0
            loc.n_ndrows[side]++;
            rdsc.nd_row = fam_row0 + fam_k + HS_MAXREP - 1;
          }
          out.rec_descs.push_back(rdsc);
        }
        if (period > HS_GRP_MAXP) loc.n_short[side]++;              // no instantiation of hs_str_group_kernel_p for this period
      }
      if (kind_of(ks[i]) == 1 || kind_of(ks[i]) == 2) loc.n_pw[side]++;      // (kinds 1 and 2 come first: n_pw counts both, the piecewise ones are [n_tab, n_pw))
      if (kind_of(ks[i]) != 0) loc.n_rp[side]++;                             // (... then kind 3: [n_pw, n_rp))
      prev = cur;
    }
  }
  HS_LAP(4);
  for (int side = 0; side < 2; side++){      // alleles sharing a trailing-flank rowset run as lanes of one wavefront (<= 64 each)
    // groups in ascending rowset order, members in allele order: (rowset, allele) pairs, sorted
    S.pairs.clear();
    for (int k = 0; k < A; k++){
      const hs_allele_t& al = out.alleles[allele_base + k];
      if (al.realign) S.pairs.push_back(((uint64_t)(uint32_t)al.trail_rows[side] << 32) | (uint32_t)k);
    }
    bool sorted = true;
    for (size_t i = 1; i < S.pairs.size() && sorted; i++) sorted = S.pairs[i-1] < S.pairs[i];
    if (!sorted) std::sort(S.pairs.begin(), S.pairs.end());
    loc.tg_begin[side] = out.tgroups.size();
    for (size_t i0 = 0; i0 < S.pairs.size(); ){
      size_t i1 = i0;
      while (i1 < S.pairs.size() && (S.pairs[i1] >> 32) == (S.pairs[i0] >> 32)) i1++;
      for (size_t m0 = i0; m0 < i1; m0 += 64){
        hs_tgroup_t g; g.rowset = (int32_t)(S.pairs[i0] >> 32); g.member_off = out.tmembers.size(); g.pad = 0;
        g.n_members = (int32_t)std::min<size_t>(64, i1 - m0);
        for (int m = 0; m < g.n_members; m++) out.tmembers.push_back((int32_t)(uint32_t)S.pairs[m0 + m]);
        out.tgroups.push_back(g);
      }
      i0 = i1;
    }
    loc.tg_count[side] = out.tgroups.size() - loc.tg_begin[side];
  }
  HS_LAP(5);
  for (int side = 0; side < 2; side++){
    out.lead_off.push_back((int32_t)out.lead_ids.size());
    out.lead_ids.insert(out.lead_ids.end(), S.lead_sets[side].begin(), S.lead_sets[side].end());
  }
  for (int r = loc.read_begin; r < loc.read_begin + loc.n_reads; r++){
    hs_read_t rd;
    rd.base_off = b->base_off[r]; rd.len = b->base_off[r+1]-b->base_off[r]; rd.locus = l; rd.seed = -1;
    const bool realign = b->realign_read ? b->realign_read[r] != 0 : true;
    sh.realign_read[r] = realign ? 1 : 0;
    if (realign){
      const bool given = seed_in && seed_in[r] != HIPSTR_SEED_AUTO;
      const int s = given ? seed_in[r] : calc_seed_base(b, l, r);
      if (s == -2){ err = "Invalid alignment seed or unrecognized CIGAR char (HapAligner.cpp:309,316)"; return 1; }
      if (given && s != -1 && (s < 1 || s > rd.len - 2)){ err = "seed base must leave at least one base on either side (HapAligner.cpp:316)"; return 1; }
      rd.seed = s;
      sh.seeds[r] = s;
      if (s >= 0){
        if (s > HS_MAX_SIDE_FWD || rd.len-s-1 > HS_MAX_SIDE_FWD){ err = "read side longer than 1024 bases is not supported"; return 1; }
        out.active.push_back(r);
        out.n_alignments += n_realigned;
        out.max_read_len = std::max(out.max_read_len, rd.len);
      }
    }
    sh.reads[r] = rd;
  }
  HS_LAP(6);
  out.loci.push_back(loc);
  return 0;
}

// Sizes of the pools of a fragment = where the next fragment starts in the merged batch.
struct FragBase { size_t loci, alleles, stropts, rowsets, rows, visits, f64, chars, active, realign_hap, order, tgroups, tmembers, leads, recs, ndrows, lead_ids, gen_f64, pmf; };
static FragBase frag_sizes(const Prepared& f){
  return FragBase{ f.loci.size(), f.alleles.size(), f.stropts.size(), f.rowsets.size(), f.rows.size(), f.visits.size(), f.f64pool.size(), f.chars.size(),
                   f.active.size(), f.realign_hap.size(), f.str_order.size(), f.tgroups.size(), f.tmembers.size(), f.lead_off.size(), f.rec_descs.size(), f.nd_rows.size(),
                   f.lead_ids.size(), (size_t)f.gen_f64, f.pmf13.size() };
}

// Copies fragment `f` to its place in `out` (whose pools are already sized), turning fragment-local pool offsets into batch-wide
// ones.  Fragments write disjoint ranges: safe to run for several fragments at once.
static void place_fragment(Prepared& out, Prepared& f, const FragBase& at){
  const int32_t allele_base = (int32_t)at.alleles, stropt_base = (int32_t)at.stropts, rowset_base = (int32_t)at.rowsets;
  const int32_t rows_base = (int32_t)at.rows, visits_base = (int32_t)at.visits, f64_base = (int32_t)at.f64;
  const int32_t chars_base = (int32_t)at.chars, tg_base = (int32_t)at.tgroups, tm_base = (int32_t)at.tmembers, order_base = (int32_t)at.order;
  for (hs_locus_t& L : f.loci){
    L.hap_begin += allele_base;
    for (int s = 0; s < 2; s++){ L.tg_begin[s] += tg_base; L.order_off[s] += order_base; L.rec_off[s] += (int32_t)at.recs; L.ndrow_off[s] += (int32_t)at.ndrows; }
  }
  for (hs_recdesc_t& r : f.rec_descs) r.stropt += stropt_base;
  for (hs_allele_t& a : f.alleles)
    if (a.realign) for (int s = 0; s < 2; s++){ a.lead_rows[s] += rowset_base; a.trail_rows[s] += rowset_base; a.str_opt[s] += stropt_base; }
  for (hs_rowset_t& r : f.rowsets) r.off += rows_base;
  for (hs_stropt_t& o : f.stropts){
    if (o.gen){ o.f64_off += (int32_t)at.gen_f64; o.tab_off += (int32_t)at.gen_f64; o.pmf_off += (int32_t)at.pmf; }
    else { o.f64_off += f64_base; o.tab_off += f64_base; }
    o.seq_off += chars_base; o.ins_off += visits_base;
    for (int q = 0; q < HS_MAXREP; q++) o.del_off[q] += visits_base;
  }
  for (hs_tgroup_t& g : f.tgroups){ g.rowset += rowset_base; g.member_off += tm_base; }
  for (int32_t& id : f.lead_ids) id += rowset_base;
  for (int32_t& o : f.lead_off) o += (int32_t)at.lead_ids;
#define HS_PLACE(field, base) std::copy(f.field.begin(), f.field.end(), out.field.begin() + (base))
  HS_PLACE(loci, at.loci); HS_PLACE(alleles, at.alleles); HS_PLACE(stropts, at.stropts); HS_PLACE(rowsets, at.rowsets);
  HS_PLACE(active, at.active); HS_PLACE(realign_hap, at.realign_hap);
  HS_PLACE(str_order, at.order); HS_PLACE(tgroups, at.tgroups); HS_PLACE(tmembers, at.tmembers); HS_PLACE(nd_rows, at.ndrows);
  HS_PLACE(lead_off, at.leads); HS_PLACE(lead_ids, at.lead_ids); HS_PLACE(rec_descs, at.recs); HS_PLACE(pmf13, at.pmf);
  // rows, visits, f64pool, chars stay in the fragment (Prepared::frags): the upload gathers them
#undef HS_PLACE
}

// ---- recycled storage (prep.h)
namespace {
#define HS_PREP_VECTORS(X) X(loci) X(alleles) X(stropts) X(rowsets) X(rows) X(visits) X(f64pool) X(chars) X(reads) X(active) X(seeds) X(realign_read) \
  X(realign_hap) X(chunks) X(ws) X(lead_items) X(trail_items) X(str_items) X(tpack) X(str_order) X(nd_rows) X(rec_descs) X(pmf13) X(tgroups) X(tmembers) X(lead_off) X(lead_ids)
size_t prepared_capacity_bytes(const Prepared& p){
  size_t n = 0;
#define X(v) n += p.v.capacity()*sizeof(p.v[0]);
  HS_PREP_VECTORS(X)
#undef X
  return n;
}
void clear_prepared(Prepared& p){          // back to the state of a fresh object, capacities kept
#define X(v) p.v.clear();
  HS_PREP_VECTORS(X)
#undef X
  p.frags.clear();
  p.grp_nd_cap = 0; p.gen_f64 = 0; p.ws_mr_size = p.ws_lt_size = p.ws_lead_size = p.ws_col_size = p.ws_nd_size = 0;
  p.max_side_len = 0; p.max_B = 1; p.n_out = 0; p.n_alignments = 0; p.max_read_len = 0; p.max_flank = 0;
}
struct PrepPool {
  std::mutex m;
  std::vector<Prepared> top, frag;        // top-level objects and fragments have different large vectors: two free lists
  size_t bytes = 0, cap_bytes;
  PrepPool(){
    const char* e = getenv("HIPSTR_PREP_POOL_MIB");
    cap_bytes = (size_t)(e ? std::max(0, atoi(e)) : 2048) << 20;
  }
  void put(std::vector<Prepared>& list, Prepared&& p){
    const size_t n = prepared_capacity_bytes(p);
    if (n == 0) return;
    clear_prepared(p);
    std::lock_guard<std::mutex> g(m);
    if (bytes + n > cap_bytes) return;      // (p's storage is released by the caller's destructor)
    bytes += n;
    list.push_back(std::move(p));
  }
  bool take(std::vector<Prepared>& list, Prepared& into){
    std::lock_guard<std::mutex> g(m);
    if (list.empty()) return false;
    into = std::move(list.back());
    list.pop_back();
    bytes -= std::min(bytes, prepared_capacity_bytes(into));
    return true;
  }
};
PrepPool& prep_pool(){ static PrepPool* p = new PrepPool(); return *p; }     // (never destroyed: host threads may recycle during process exit)
}  // namespace

void recycle_prepared(Prepared& p){
  PrepPool& P = prep_pool();
  for (Prepared& f : p.frags) P.put(P.frag, std::move(f));
  p.frags.clear();
  P.put(P.top, std::move(p));
  Prepared empty; std::swap(p, empty);
}
void adopt_recycled(Prepared& p){ prep_pool().take(prep_pool().top, p); }

int prepare_batch(const hipstr_batch_t* b, Prepared& out, std::string& err, int64_t ws_budget, const int32_t* seed_in){
  host_tables();
  const bool timing = getenv("HIPSTR_TIMING") != NULL;
  auto now = [](){ return std::chrono::steady_clock::now(); };
  auto lap = [&](const char* what, std::chrono::steady_clock::time_point& t){
    if (timing){ const auto t2 = now(); fprintf(stderr, "prepare_batch: %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t2 - t).count()); t = t2; }
  };
  auto t_lap = now();
  g_bnd_scale = getenv("HIPSTR_DEBUG_BND_SCALE") ? atof(getenv("HIPSTR_DEBUG_BND_SCALE")) : 1.0;
  g_host_tables = getenv("HIPSTR_HOST_TABLES") && atoi(getenv("HIPSTR_HOST_TABLES")) != 0;
  if (validate_tables(b, err)) return 1;
  const int n_reads_total = b->n_loci > 0 ? b->read_off[b->n_loci] : 0;
  out.seeds.assign(n_reads_total, -1);
  out.realign_read.assign(n_reads_total, 1);
  out.reads.resize(n_reads_total);
  // per-locus starts in the option table and in the output, and a cost estimate to cut the loci into balanced fragments
  std::vector<int> opt_start(b->n_loci + 1, 0);
  std::vector<int64_t> out_off_v(b->n_loci + 1, 0), cost(b->n_loci + 1, 0);
  for (int l = 0; l < b->n_loci; l++){
    int64_t A = 1; int no = 0;
    for (int k = 0; k < 3; k++){
      const int n = b->blk_nopts[3*l+k];
      if (n < 1){ err = "haplotype block without options"; return 1; }
      no += n; A *= n;
    }
    opt_start[l+1] = opt_start[l] + no;
    const int64_t P = b->read_off[l+1] - b->read_off[l];
    if (P < 0){ err = "read_off must not decrease"; return 1; }
    out_off_v[l+1] = out_off_v[l] + P*A;
    cost[l+1] = cost[l] + 64*(int64_t)b->blk_nopts[3*l+1] + 4*A + P;
  }
  const int64_t out_off = out_off_v[b->n_loci];
  lap("setup", t_lap);
  PrepShared sh; sh.reads = out.reads.data(); sh.seeds = out.seeds.data(); sh.realign_read = out.realign_read.data();
  sh.out_off = out_off_v.data(); sh.seed_in = seed_in;
  int n_threads = std::min(host_threads(), std::max(1, b->n_loci / 4));
  if (n_threads <= 1){
    for (int l = 0; l < b->n_loci; l++)
      if (prepare_locus(b, l, opt_start[l], sh, out, err)) return 1;
  } else {
    // fragments of consecutive loci with near-equal cost, a few per thread so that a slow fragment does not stall the rest
    const int n_frag = std::min(b->n_loci, n_threads*4);
    std::vector<int> cut(n_frag + 1, 0);
    for (int f = 1; f < n_frag; f++)
      cut[f] = std::max(cut[f-1], (int)(std::lower_bound(cost.begin(), cost.end(), cost[b->n_loci]*f/n_frag) - cost.begin()));
    cut[n_frag] = b->n_loci;
    std::vector<Prepared> frag(n_frag);
    for (Prepared& f : frag) prep_pool().take(prep_pool().frag, f);      // recycled storage, if any (cleared)
    std::vector<std::string> frag_err(n_frag);
    std::vector<int> frag_rc(n_frag, 0);
    parallel_for(n_frag, n_threads, [&](int f){
      for (int l = cut[f]; l < cut[f+1] && !frag_rc[f]; l++)
        frag_rc[f] = prepare_locus(b, l, opt_start[l], sh, frag[f], frag_err[f]);
    });
    lap("fragments", t_lap);
    for (int f = 0; f < n_frag; f++) if (frag_rc[f]){ err = frag_err[f]; return 1; }
    std::vector<FragBase> at(n_frag + 1);
    memset(&at[0], 0, sizeof(FragBase));
    for (int f = 0; f < n_frag; f++){
      const FragBase sz = frag_sizes(frag[f]);
      const size_t* a = (const size_t*)&at[f]; const size_t* z = (const size_t*)&sz; size_t* n = (size_t*)&at[f+1];
      for (size_t i = 0; i < sizeof(FragBase)/sizeof(size_t); i++) n[i] = a[i] + z[i];
      out.max_B = std::max(out.max_B, frag[f].max_B); out.max_flank = std::max(out.max_flank, frag[f].max_flank);
      out.max_read_len = std::max(out.max_read_len, frag[f].max_read_len); out.n_alignments += frag[f].n_alignments;
    }
    const FragBase& tot = at[n_frag];
    out.loci.resize(tot.loci); out.alleles.resize(tot.alleles); out.stropts.resize(tot.stropts); out.rowsets.resize(tot.rowsets);
    out.active.resize(tot.active); out.realign_hap.resize(tot.realign_hap);
    out.str_order.resize(tot.order); out.tgroups.resize(tot.tgroups); out.tmembers.resize(tot.tmembers); out.lead_off.resize(tot.leads); out.lead_ids.resize(tot.lead_ids);
    out.rec_descs.resize(tot.recs); out.pmf13.resize(tot.pmf); out.gen_f64 = (int64_t)tot.gen_f64;
    out.nd_rows.resize(tot.ndrows);
    parallel_for(n_frag, n_threads, [&](int f){ place_fragment(out, frag[f], at[f]); });
    for (Prepared& f : frag){      // keep only the large pools of the fragments
      f.loci.clear(); f.alleles.clear(); f.stropts.clear(); f.rowsets.clear(); f.active.clear(); f.realign_hap.clear(); f.str_order.clear();
      f.tgroups.clear(); f.tmembers.clear(); f.nd_rows.clear(); f.lead_off.clear(); f.lead_ids.clear(); f.rec_descs.clear(); f.pmf13.clear();
    }
    out.frags = std::move(frag);
  }
  lap("loci", t_lap);
  out.n_out = out_off;

  // ---- launch plan: workspaces + work items, chunked so that the workspaces stay within the budget
  out.ws.resize(out.active.size());
  Prepared::Chunk ch; memset(&ch, 0, sizeof ch);
  int64_t mr = 0, lt = 0, lead = 0, col = 0, nd = 0;
  auto flush = [&](int active_end){
    ch.active_end = active_end;
    ch.lead_begin = out.lead_items.size();
    // trailing-flank items: lanes = alleles of a group; when a group has <= 32 alleles, 64/npad reads of the same locus and
    // side (sorted by side length, so that packed reads finish together) share one wavefront
    ch.trail_begin = out.trail_items.size();
    ch.str_begin = out.str_items.size();
    // the (locus, range of active reads) runs of this chunk; their items are built in blocks of consecutive runs (host threads), each
    // block into its own buffers, and appended in locus order with their offsets into the packed-read table rebased
    struct Run { int a0, a1; };
    std::vector<Run> runs;
    for (int a0 = ch.active_begin; a0 < active_end; ){
      const int locus = out.reads[out.active[a0]].locus;
      int a1 = a0;
      while (a1 < active_end && out.reads[out.active[a1]].locus == locus) a1++;
      runs.push_back(Run{a0, a1});
      a0 = a1;
    }
    lap("plan:runs", t_lap);
    struct Blk { std::vector<hs_item_t> lead, trail, str; std::vector<int32_t> tpack; int nd_cap = 0, n_long = 0; };
    const int RUNS_PER_BLK = 64;
    // one thread (a small batch, or a stream worker on a small allowance): the items go straight to the batch's arrays — one block whose
    // vectors ARE the batch's (entries of the packed-read table are then batch-wide indices already)
    const bool direct = (runs.size() < 16 || host_threads() == 1);
    const int n_blk = direct ? 1 : (int)((runs.size() + RUNS_PER_BLK - 1) / RUNS_PER_BLK);
    std::vector<Blk> blks(n_blk);
    if (direct){ blks[0].lead.swap(out.lead_items); blks[0].trail.swap(out.trail_items); blks[0].str.swap(out.str_items); blks[0].tpack.swap(out.tpack); }
    parallel_for(n_blk, direct ? 1 : host_threads(), [&](int bi){
      Blk& R = blks[bi];
      struct Bin { int cols, n; int m[16]; };
      thread_local std::vector<int> order, lens, slen, cnt;
      thread_local std::vector<Bin> bins;
      const size_t r_lo = direct ? 0 : (size_t)bi*RUNS_PER_BLK, r_hi = direct ? runs.size() : std::min(runs.size(), r_lo + RUNS_PER_BLK);
      for (size_t ri = r_lo; ri < r_hi; ri++){
      const int a0 = runs[ri].a0, a1 = runs[ri].a1, n_run = a1 - a0;
      const int locus = out.reads[out.active[a0]].locus;
      const hs_locus_t& loc = out.loci[locus];
      for (int s = 0; s < 2; s++){
        // the run's reads by side length, ties in read order: a counting sort (side lengths are at most HS_MAX_SIDE_FWD)
        lens.resize(n_run); order.resize(n_run); slen.resize(n_run);
        int maxk = 0;
        for (int a = a0; a < a1; a++){
          const hs_read_t& r = out.reads[out.active[a]];
          const int k = s ? r.len - r.seed - 1 : r.seed;
          lens[a - a0] = k; maxk = std::max(maxk, k);
        }
        cnt.assign((size_t)maxk + 2, 0);
        for (int i = 0; i < n_run; i++) cnt[lens[i] + 1]++;
        for (int k = 0; k <= maxk; k++) cnt[k + 1] += cnt[k];
        for (int i = 0; i < n_run; i++){ const int at = cnt[lens[i]]++; order[at] = a0 + i; slen[at] = lens[i]; }
        auto side_len_at = [&](int i){ return slen[i]; };
        // leading-flank items: lanes = reads (sorted by side length, so that the lanes of a wavefront finish together), one item per
        // distinct leading flank of the locus and side
        const int32_t* ls = out.lead_ids.data() + out.lead_off[2*locus + s];
        for (int slot = 0; slot < loc.n_lead[s]; slot++)
          for (int i = 0; i < n_run; i += 64){
            hs_item_t it; it.side = s | ((int32_t)slot << 1); it.rowset = ls[slot];
            it.active = (int32_t)R.tpack.size();
            it.slot = (int32_t)std::min(64, n_run - i);
            R.tpack.insert(R.tpack.end(), order.begin() + i, order.begin() + i + it.slot);
            R.lead.push_back(it);
          }
        // STR-block items (hs_str_group_kernel): reads of this locus and side whose columns, laid end to end, fill one workgroup's
        // lanes.  First fit, longest side first; a group holds at most HS_GRP_MAXREADS reads and 2 HS_GRP_COLS read-end deletion sums
        // (21 period per read).  item.active = first entry in tpack, item.slot = number of reads, item.rowset = their columns
        if (loc.n_rp[s] > 0){
          const int period = out.stropts[out.alleles[loc.hap_begin + (out.str_order[loc.order_off[s]] & 0x1fffffff)].str_opt[s]].period;
          const int max_reads = std::max(1, std::min(16, 2*HS_GRP_COLS / (21*period)));
          bins.clear(); size_t first_open = 0;
          for (int i = n_run; i-- > 0; ){
            const int a = order[i], nc = side_len_at(i);
            if (nc <= 0) continue;
            if (nc > HS_GRP_COLS){ R.n_long++; continue; }
            size_t bn = first_open;
            for (; bn < bins.size(); bn++) if (bins[bn].cols + nc <= HS_GRP_COLS && bins[bn].n < max_reads) break;
            if (bn == bins.size()){ bins.emplace_back(); bins.back().cols = 0; bins.back().n = 0; }
            bins[bn].cols += nc; bins[bn].m[bins[bn].n++] = a;
            while (first_open < bins.size() && (bins[first_open].cols >= HS_GRP_COLS || bins[first_open].n >= max_reads)) first_open++;
          }
          for (const Bin& bn : bins){
            hs_item_t it; it.side = s; it.rowset = bn.cols; it.slot = (int32_t)bn.n;
            it.active = (int32_t)R.tpack.size();
            R.tpack.insert(R.tpack.end(), bn.m, bn.m + bn.n);
            R.str.push_back(it);
            R.nd_cap = std::max(R.nd_cap, it.slot*36*period);
          }
        }
        for (int g = 0; g < loc.tg_count[s]; g++){
          const int nm = out.tgroups[loc.tg_begin[s] + g].n_members;
          int npad = 1; while (npad < nm) npad <<= 1;
          const int per_wave = 64 / npad;
          for (int i = 0; i < n_run; i += per_wave){
            hs_item_t it; it.side = s; it.slot = loc.tg_begin[s] + g;
            it.active = (int32_t)R.tpack.size();
            it.rowset = (int32_t)std::min(per_wave, n_run - i);
            R.tpack.insert(R.tpack.end(), order.begin() + i, order.begin() + i + it.rowset);
            R.trail.push_back(it);
          }
        }
      }
      }
    });
    lap("plan:items", t_lap);
    if (direct){
      blks[0].lead.swap(out.lead_items); blks[0].trail.swap(out.trail_items); blks[0].str.swap(out.str_items); blks[0].tpack.swap(out.tpack);
      out.grp_nd_cap = std::max(out.grp_nd_cap, blks[0].nd_cap); ch.n_long_sides += blks[0].n_long;
    } else {   // append the blocks' pieces in locus order: where every piece goes follows from the sizes, the copies are shared by the host threads
      struct At { size_t tp, le, tr, st; };
      std::vector<At> at(blks.size() + 1);
      at[0] = At{ out.tpack.size(), out.lead_items.size(), out.trail_items.size(), out.str_items.size() };
      for (size_t i = 0; i < blks.size(); i++){
        const Blk& R = blks[i];
        at[i+1] = At{ at[i].tp + R.tpack.size(), at[i].le + R.lead.size(), at[i].tr + R.trail.size(), at[i].st + R.str.size() };
        out.grp_nd_cap = std::max(out.grp_nd_cap, R.nd_cap);
        ch.n_long_sides += R.n_long;
      }
      const At& end = at[blks.size()];
      out.tpack.resize(end.tp); out.lead_items.resize(end.le); out.trail_items.resize(end.tr); out.str_items.resize(end.st);
      parallel_for((int)blks.size(), blks.size() >= 4 ? host_threads() : 1, [&](int i){
        Blk& R = blks[i];
        const int32_t base = (int32_t)at[i].tp;
        for (hs_item_t& it : R.lead) it.active += base;
        for (hs_item_t& it : R.trail) it.active += base;
        for (hs_item_t& it : R.str) it.active += base;
        std::copy(R.tpack.begin(), R.tpack.end(), out.tpack.begin() + at[i].tp);
        std::copy(R.lead.begin(), R.lead.end(), out.lead_items.begin() + at[i].le);
        std::copy(R.trail.begin(), R.trail.end(), out.trail_items.begin() + at[i].tr);
        std::copy(R.str.begin(), R.str.end(), out.str_items.begin() + at[i].st);
        std::vector<hs_item_t>().swap(R.lead); std::vector<hs_item_t>().swap(R.trail); std::vector<hs_item_t>().swap(R.str); std::vector<int32_t>().swap(R.tpack);
      });
    }
    lap("plan:merge", t_lap);
    ch.trail_end = out.trail_items.size();
    ch.str_end = out.str_items.size();
    ch.lead_end = out.lead_items.size();
    out.ws_mr_size = std::max(out.ws_mr_size, mr); out.ws_lt_size = std::max(out.ws_lt_size, lt); out.ws_lead_size = std::max(out.ws_lead_size, lead);
    out.ws_col_size = std::max(out.ws_col_size, col); out.ws_nd_size = std::max(out.ws_nd_size, nd);
    if (ch.active_end > ch.active_begin) out.chunks.push_back(ch);
    memset(&ch, 0, sizeof ch); ch.active_begin = active_end;
    mr = lt = lead = col = nd = 0;
  };
  // Workspace offsets are running sums over the active reads.  The usual case — everything fits one chunk — is done in blocks of reads:
  // block totals in parallel, their prefix, then the offsets in parallel; a batch that needs cutting takes the read-by-read loop below.
  bool ws_done = false;
  if (out.active.size() >= 4096){
    struct Sum { int64_t mr, lt, lead, col, nd, aln; int max_side; };
    auto need_of = [&](size_t ai, Sum& z, hs_ws_t* w){
      const hs_read_t& rd = out.reads[out.active[ai]];
      const hs_locus_t& loc = out.loci[rd.locus];
      const int n_side[2] = { rd.seed, rd.len - rd.seed - 1 };
      const int64_t per6 = HS_MAXREP*(int64_t)loc.period;
      if (w){ w->mr = z.mr; w->lt = z.lt; w->col = z.col; w->nd[0] = z.nd; w->nd[1] = z.nd + loc.n_ndrows[0]*per6; }
      z.mr += (int64_t)loc.n_re*(rd.len-1); z.lt += (int64_t)loc.n_re*loc.lt_stride; z.col += 3*(int64_t)(rd.len-1);
      z.nd += (loc.n_ndrows[0] + loc.n_ndrows[1])*per6;
      for (int s = 0; s < 2; s++){
        if (w) w->lead[s] = z.lead;
        z.lead += (int64_t)loc.n_lead[s]*(n_side[s] + loc.lead_flank[s] + 1);
        z.max_side = std::max(z.max_side, n_side[s]);
      }
      z.aln += loc.n_re;
    };
    const int nblk = host_threads()*4;
    const size_t n = out.active.size();
    std::vector<Sum> tot(nblk + 1);
    parallel_for(nblk, host_threads(), [&](int b){
      Sum z; memset(&z, 0, sizeof z);
      for (size_t ai = n*b/nblk; ai < n*(b + 1)/nblk; ai++) need_of(ai, z, NULL);
      tot[b + 1] = z;
    });
    memset(&tot[0], 0, sizeof(Sum));
    for (int b = 1; b <= nblk; b++){
      Sum& z = tot[b]; const Sum& y = tot[b - 1];
      z.mr += y.mr; z.lt += y.lt; z.lead += y.lead; z.col += y.col; z.nd += y.nd; z.aln += y.aln; z.max_side = std::max(z.max_side, y.max_side);
    }
    const Sum& all = tot[nblk];
    if (all.mr <= ws_budget && all.lt <= ws_budget && all.lead <= ws_budget && all.nd <= ws_budget){
      parallel_for(nblk, host_threads(), [&](int b){
        Sum z = tot[b]; z.max_side = 0;
        for (size_t ai = n*b/nblk; ai < n*(b + 1)/nblk; ai++){ hs_ws_t w; need_of(ai, z, &w); out.ws[ai] = w; }
      });
      mr = all.mr; lt = all.lt; lead = all.lead; col = all.col; nd = all.nd;
      ch.n_alignments = all.aln; out.max_side_len = std::max(out.max_side_len, all.max_side);
      ws_done = true;
    }
  }
  for (size_t ai = 0; !ws_done && ai < out.active.size(); ai++){
    const hs_read_t& rd = out.reads[out.active[ai]];
    const hs_locus_t& loc = out.loci[rd.locus];
    const int n_side[2] = { rd.seed, rd.len - rd.seed - 1 };
    const int64_t need_mr = (int64_t)loc.n_re*(rd.len-1), need_lt = (int64_t)loc.n_re*loc.lt_stride;
    int64_t need_lead = 0;
    for (int s = 0; s < 2; s++) need_lead += (int64_t)loc.n_lead[s]*(n_side[s] + loc.lead_flank[s] + 1);
    int64_t need_nd = 0, per6 = 0;
    if (loc.n_ndrows[0] + loc.n_ndrows[1] > 0){
      per6 = HS_MAXREP*(int64_t)loc.period;
      need_nd = (loc.n_ndrows[0] + loc.n_ndrows[1])*per6;
    }
    if (ch.active_begin < (int)ai && (mr + need_mr > ws_budget || lt + need_lt > ws_budget || lead + need_lead > ws_budget || nd + need_nd > ws_budget)) flush((int)ai);
    hs_ws_t w; w.mr = mr; w.lt = lt; w.col = col;
    w.nd[0] = nd; w.nd[1] = nd + loc.n_ndrows[0]*per6; nd += need_nd;
    for (int s = 0; s < 2; s++){
      w.lead[s] = lead;
      out.max_side_len = std::max(out.max_side_len, n_side[s]);
      lead += (int64_t)loc.n_lead[s]*(n_side[s] + loc.lead_flank[s] + 1);
    }
    out.ws[ai] = w;
    mr += need_mr; lt += need_lt; col += 3*(int64_t)(rd.len-1);
    ch.n_alignments += loc.n_re;
  }
  lap("plan:ws", t_lap);
  flush((int)out.active.size());
  lap("plan", t_lap);
  return 0;
}

}  // namespace hipstr
