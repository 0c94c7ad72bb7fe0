// HapAlignerMI355X.cpp — see HapAlignerMI355X.h.  Flattens the reference's objects into the C-ABI batch.
#include <assert.h>
#include <string.h>
#include <stdlib.h>

#include "HapAlignerMI355X.h"
#include "RepeatBlock.h"
#include "../error.h"
#include "../stutter_model.h"

#include "hipstr_hmm.h"

HapAlignerMI355X::HapAlignerMI355X(Haplotype* haplotype, std::vector<bool>& realign_to_haplotype)
  : fw_haplotype_(haplotype), realign_to_hap_(realign_to_haplotype){
  assert(realign_to_haplotype.size() == (size_t)haplotype->num_combs());
  if (haplotype->num_blocks() != HIPSTR_NUM_BLOCKS)
    printErrorAndDie("HapAlignerMI355X requires a [flank, repeat, flank] haplotype");
  opt_off_.push_back(0);
  period_ = 0;
  for (int i = 0; i < haplotype->num_blocks(); i++){
    HapBlock* block = haplotype->get_block(i);
    blk_start_.push_back(block->start());
    blk_end_.push_back(block->end());
    blk_nopts_.push_back(block->num_options());
    for (int o = 0; o < block->num_options(); o++){
      seq_ += block->get_seq(o);
      opt_off_.push_back((int32_t)seq_.size());
    }
    RepeatStutterInfo* info = block->get_repeat_info();
    if ((i == 1) != (info != NULL))
      printErrorAndDie("HapAlignerMI355X requires a [flank, repeat, flank] haplotype");
    if (info != NULL){
      period_ = info->get_period();
      StutterModel* m = info->get_stutter_model();
      stutter_.push_back(m->get_parameter(true,  'P')); stutter_.push_back(m->get_parameter(true,  'U')); stutter_.push_back(m->get_parameter(true,  'D'));
      stutter_.push_back(m->get_parameter(false, 'P')); stutter_.push_back(m->get_parameter(false, 'U')); stutter_.push_back(m->get_parameter(false, 'D'));
    }
  }
  hap_off_.push_back(0);
  hap_off_.push_back(haplotype->num_combs());
  for (size_t k = 0; k < realign_to_hap_.size(); k++)
    realign_hap_.push_back(realign_to_hap_[k] ? 1 : 0);
}

// the reads of a call in the C-ABI's flat form (global name: HapAlignerMI355X.h declares run_trace_requests over it)
struct HipstrFlatReads {
  std::vector<int32_t> read_off, base_off, read_start, cigar_off, cigar_len;
  std::vector<uint8_t> realign;
  std::string bases, quals, cigar_op;
  HipstrFlatReads(){}
  size_t size() const { return read_start.size(); }
  std::string seq_of(size_t i) const { return bases.substr(base_off[i], base_off[i+1] - base_off[i]); }
  std::string quals_of(size_t i) const { return quals.substr(base_off[i], base_off[i+1] - base_off[i]); }
  // read i is `a`, field by field (start, bases, qualities, CIGAR)
  bool same_read(size_t i, const Alignment& a) const {
    const size_t n = (size_t)(base_off[i+1] - base_off[i]);
    if (a.get_start() != read_start[i] || a.get_sequence().size() != n || a.get_base_qualities().size() != n) return false;
    if (bases.compare(base_off[i], n, a.get_sequence()) != 0 || quals.compare(base_off[i], n, a.get_base_qualities()) != 0) return false;
    const std::vector<CigarElement>& cig = a.get_cigar_list();
    if ((size_t)(cigar_off[i+1] - cigar_off[i]) != cig.size()) return false;
    for (size_t c = 0; c < cig.size(); c++) if (cig[c].get_type() != cigar_op[cigar_off[i] + c] || cig[c].get_num() != cigar_len[cigar_off[i] + c]) return false;
    return true;
  }
  HipstrFlatReads(const std::vector<Alignment>& alns, const std::vector<bool>& realign_read){
    read_off.push_back(0); base_off.push_back(0); cigar_off.push_back(0);
    for (size_t i = 0; i < alns.size(); i++){
      bases += alns[i].get_sequence();
      quals += alns[i].get_base_qualities();
      base_off.push_back((int32_t)bases.size());
      read_start.push_back(alns[i].get_start());
      const std::vector<CigarElement>& cig = alns[i].get_cigar_list();
      for (size_t c = 0; c < cig.size(); c++){ cigar_op += cig[c].get_type(); cigar_len.push_back(cig[c].get_num()); }
      cigar_off.push_back((int32_t)cigar_op.size());
      realign.push_back(realign_read[i] ? 1 : 0);
    }
    read_off.push_back((int32_t)alns.size());
    if (cigar_len.empty()) cigar_len.push_back(0);
  }
};
typedef HipstrFlatReads FlatReads;

#define FILL_BATCH(b, r)                                                                                          \
  hipstr_batch_t b;                                                                                              \
  b.n_loci = 1; b.blk_start = blk_start_.data(); b.blk_end = blk_end_.data(); b.blk_nopts = blk_nopts_.data();   \
  b.period = &period_; b.stutter = stutter_.data(); b.opt_off = opt_off_.data(); b.seq = seq_.data();            \
  b.hap_off = hap_off_.data(); b.realign_hap = realign_hap_.data(); b.read_off = r.read_off.data();              \
  b.base_off = r.base_off.data(); b.bases = r.bases.data(); b.quals = r.quals.data();                            \
  b.read_start = r.read_start.data(); b.cigar_off = r.cigar_off.data(); b.cigar_op = r.cigar_op.data();          \
  b.cigar_len = r.cigar_len.data(); b.realign_read = r.realign.data();

int HapAlignerMI355X::calc_seed_base(const Alignment& alignment){
  FlatReads r(std::vector<Alignment>(1, alignment), std::vector<bool>(1, true));
  FILL_BATCH(b, r)
  int32_t seed = -1;
  if (hipstr_calc_seed_bases(&b, &seed) != 0)
    printErrorAndDie(hipstr_last_error());
  return seed;
}

// ---- where the adapter's own host time goes (genotype_flow --profile): seconds summed over threads
#include <atomic>
#include <chrono>
static std::atomic<long long> g_adapter_ns[3];       // 0 flatten (FlatReads + batch struct)  1 fill AlignmentTrace objects  2 haplotype strings for the traces
static std::atomic<bool> g_adapter_prof(false);
namespace {
struct AdapterTimer {
  int k; std::chrono::steady_clock::time_point t0; bool on;
  explicit AdapterTimer(int k_) : k(k_), on(g_adapter_prof.load()) { if (on) t0 = std::chrono::steady_clock::now(); }
  void stop(){ if (on){ g_adapter_ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); on = false; } }
  ~AdapterTimer(){ stop(); }
};
}
void HapAlignerMI355X::profile(bool enable, double seconds[3]){
  if (seconds) for (int i = 0; i < 3; i++) seconds[i] = 1e-9*(double)g_adapter_ns[i].load();
  if (enable) for (int i = 0; i < 3; i++) g_adapter_ns[i] = 0;
  g_adapter_prof = enable;
}

static hipstr_stream_t* g_shared_stream = NULL;
void HapAlignerMI355X::use_stream(hipstr_stream* stream){ g_shared_stream = stream; }

// ---- trace_optimal_aln behind an UNEDITED caller.  SeqStutterGenotyper::retrace_alignments (seq_stutter_genotyper.cpp:805-841) asks for
// one traceback at a time — the read's pool against the likelier of its sample's two MAP haplotypes — from a HapAligner it has just
// constructed: one device round trip per read (0.4 ms each, 100 per locus) if taken literally.  What the adapter can know: the pooled
// reads of the locus (process_reads was handed the pooler's vector, seq_stutter_genotyper.cpp:524-528, on this thread) and the
// haplotypes.  So on the first request after a process_reads it aligns every pool to every haplotype once more (one forward call),
// traces every seeded pool against its two likeliest haplotypes in ONE hipstr_hmm_trace call and keeps the results by (pool, haplotype);
// a later request that is not among them is answered by one call that traces ALL pools against the haplotype asked for (the reads of a
// sample share their MAP haplotypes: a handful of distinct ones per locus).  Nothing is assumed about the caller: the pool of a request
// is recognised by its address inside the vector process_reads saw and confirmed by content, the haplotypes by a hash of their sequences
// and stutter model; anything that does not match falls back to the single traceback of round 3.
#include <map>
namespace {
std::atomic<long long> g_tc_hits(0), g_tc_misses(0), g_tc_calls(0), g_tc_ahead(0);
unsigned long long fnv(const void* p, size_t n, unsigned long long h){ const unsigned char* q = (const unsigned char*)p; for (size_t i = 0; i < n; i++){ h ^= q[i]; h *= 1099511628211ull; } return h; }
struct TraceStash {
  // The pooled reads process_reads saw, as the adapter's OWN flattened copy: nothing below ever dereferences the caller's vector again (it
  // may be gone by the time a traceback is asked for — another object on this thread calling trace_optimal_aln without a process_reads
  // before it).  base_addr / n_pools only serve to recognise a request's pool by ADDRESS arithmetic; the content check follows.
  uintptr_t base_addr; size_t n_pools; unsigned long long hap_hash;
  FlatReads flat;                                    // every pool, realign mask all true
  std::vector<int32_t> seeds;                        // calc_seed_base of every pool
  bool primed;
  std::map<std::pair<int,int>, AlignmentTrace*> ready;      // computed, not handed out yet: ours to delete
  TraceStash() : base_addr(0), n_pools(0), hap_hash(0), primed(false) {}
  void drop(){
    for (std::map<std::pair<int,int>, AlignmentTrace*>::iterator it = ready.begin(); it != ready.end(); ++it) delete it->second;
    ready.clear(); seeds.clear(); primed = false;
  }
  void forget(){ drop(); base_addr = 0; n_pools = 0; flat = FlatReads(); }
  ~TraceStash(){ drop(); }
};
thread_local TraceStash t_stash;
// what the patched Genotyper::calc_log_sample_posteriors says about the requests to come (genotyper_posteriors_mi355x.inc)
struct TraceHint { int n_reads, n_alleles; std::vector<int32_t> best; std::vector<double> rows; TraceHint() : n_reads(0), n_alleles(0) {} };
thread_local TraceHint t_hint;
bool g_trace_prefetch = !(getenv("HIPSTR_ADAPTER_PREFETCH") && atoi(getenv("HIPSTR_ADAPTER_PREFETCH")) == 0);     // 0: one device call per request, as in round 3
}
void hipstr_mi355x_trace_hint(int n_reads, int n_alleles, const int32_t* best_haplotype, const double* log_aln_probs){
  TraceHint& H = t_hint;
  H.n_reads = n_reads; H.n_alleles = n_alleles;
  H.best.assign(best_haplotype, best_haplotype + n_reads);
  H.rows.assign(log_aln_probs, log_aln_probs + (size_t)n_reads*n_alleles);
}
void HapAlignerMI355X::trace_cache_stats(long long c[4]){ c[0] = g_tc_hits; c[1] = g_tc_misses; c[2] = g_tc_calls; c[3] = g_tc_ahead; }

unsigned long long HapAlignerMI355X::haplotype_hash() const {
  unsigned long long h = fnv(seq_.data(), seq_.size(), 1469598103934665603ull);
  h = fnv(opt_off_.data(), opt_off_.size()*sizeof(int32_t), h);
  h = fnv(blk_start_.data(), blk_start_.size()*sizeof(int32_t), h);
  h = fnv(stutter_.data(), stutter_.size()*sizeof(double), h);
  return fnv(&period_, sizeof period_, h);
}

void HapAlignerMI355X::process_reads(const std::vector<Alignment>& alignments, int init_read_index, const BaseQuality* base_quality,
				     const std::vector<bool>& realign_read, double* aln_probs, int* seed_positions){
  assert(alignments.size() == realign_read.size());
  (void)base_quality;     // BaseQuality's tables are constants of the model; the device holds the same values
  // a new round of this thread's locus: tracebacks computed ahead for the previous one are void; remember where the pooled reads live
  t_stash.drop();
  AdapterTimer t_flat(0);
  FlatReads r(alignments, realign_read);
  FILL_BATCH(b, r)
  if (g_trace_prefetch && !alignments.empty()){
    t_stash.base_addr = (uintptr_t)&alignments[0]; t_stash.n_pools = alignments.size(); t_stash.hap_hash = haplotype_hash();
    t_stash.flat = r; t_stash.flat.realign.assign(alignments.size(), 1);
  } else t_stash.forget();
  t_flat.stop();
  if (g_shared_stream != NULL){      // this locus' round joins whatever the other loci in flight submitted
    const int64_t ticket = hipstr_stream_submit(g_shared_stream, &b);
    if (ticket < 0 || hipstr_stream_take(g_shared_stream, ticket, aln_probs + (size_t)init_read_index*fw_haplotype_->num_combs(),
					 (int64_t)alignments.size()*fw_haplotype_->num_combs(), seed_positions + init_read_index, (int64_t)alignments.size()) != 0)
      printErrorAndDie(hipstr_last_error());
    return;
  }
  if (hipstr_hmm_process_reads(&b, aln_probs + (size_t)init_read_index*fw_haplotype_->num_combs(), seed_positions + init_read_index) != 0)
    printErrorAndDie(hipstr_last_error());
}

void HapAlignerMI355X::process_read(const Alignment& aln, int seed_base, const BaseQuality* base_quality, bool retrace_aln,
				    double* prob_ptr, AlignmentTrace& traced_aln){
  assert(seed_base != -1);
  assert(aln.get_sequence().size() == aln.get_base_qualities().size());
  (void)base_quality;
  // the haplotypes the reference's do/while visits (HapAligner.cpp:613-692): from the current one to the last, one if fixed
  std::vector<int> visited;
  do { visited.push_back(fw_haplotype_->cur_index()); } while (fw_haplotype_->next());
  fw_haplotype_->reset();
  FlatReads r(std::vector<Alignment>(1, aln), std::vector<bool>(1, true));
  FILL_BATCH(b, r)
  std::vector<uint8_t> mask(realign_hap_.size(), 0);
  for (size_t i = 0; i < visited.size(); i++) mask[visited[i]] = realign_hap_[visited[i]];
  b.realign_hap = mask.data();
  std::vector<double> row(realign_hap_.size(), 0.0);
  int32_t seed_in = seed_base, seed_out = -1;
  if (hipstr_hmm_process_reads_seeded(&b, &seed_in, row.data(), &seed_out) != 0)
    printErrorAndDie(hipstr_last_error());
  double max_LL = -100000000;
  for (size_t i = 0; i < visited.size(); i++, prob_ptr++){
    const int k = visited[i];
    if (!mask[k]) continue;
    *prob_ptr = row[k];
    if (row[k] > max_LL){
      max_LL = row[k];
      if (retrace_aln)      // filled through the same public mutators the reference's retrace uses (HapAligner.cpp:642-684)
	run_traces(std::vector<Alignment>(1, aln), std::vector<int>(1, seed_base), std::vector<int>(1, k), std::vector<AlignmentTrace*>(1, &traced_aln));
    }
  }
}

AlignmentTrace* HapAlignerMI355X::prefetched_trace(const Alignment& orig_aln, int seed_base, int best_haplotype){
  TraceStash& S = t_stash;
  if (!g_trace_prefetch || S.n_pools == 0 || S.hap_hash != haplotype_hash()) return NULL;
  // the request's pool by address arithmetic (integers: nothing is dereferenced, unrelated objects compare fine) ...
  const uintptr_t at = (uintptr_t)&orig_aln;
  if (at < S.base_addr || at >= S.base_addr + S.n_pools*sizeof(Alignment) || (at - S.base_addr) % sizeof(Alignment) != 0) return NULL;
  const int p = (int)((at - S.base_addr) / sizeof(Alignment));
  // ... confirmed by content against the adapter's own copy
  if (!S.flat.same_read((size_t)p, orig_aln)){ S.forget(); return NULL; }
  const int A = fw_haplotype_->num_combs();
  if (best_haplotype < 0 || best_haplotype >= A || !realign_to_hap_[best_haplotype]) return NULL;
  for (int k = 0; k < A; k++) if (!realign_to_hap_[k]) return NULL;      // (the reference traces with an all-true mask: anything else takes the plain path)
  const FlatReads& pools = S.flat;
  if (!S.primed){
    // the pools, their seeds, and one forward pass over all of them: which two haplotypes does every pool fit best?
    const FlatReads& r = pools;
    FILL_BATCH(b, r)
    std::vector<double> ll(S.n_pools*(size_t)A, 0.0);
    S.seeds.assign(S.n_pools, -1);
    if (hipstr_hmm_process_reads(&b, ll.data(), S.seeds.data()) != 0) printErrorAndDie(hipstr_last_error());
    g_tc_calls++;
    if (S.seeds[p] != seed_base){ S.forget(); return NULL; }       // a caller with seeds of its own: not the loop this is made for
    std::vector<int32_t> req_read, req_seed, req_hap;
    std::map<std::pair<int,int>, bool> wanted;
    auto want = [&](int pool, int hap){
      if (pool < 0 || hap < 0 || hap >= A || S.seeds[pool] < 0 || wanted.count(std::make_pair(pool, hap))) return;
      wanted[std::make_pair(pool, hap)] = true;
      req_read.push_back(pool); req_seed.push_back(S.seeds[pool]); req_hap.push_back(hap);
    };
    want(p, best_haplotype);
    const TraceHint& H = t_hint;
    bool hinted = false;
    if (H.n_alleles == A && H.n_reads > 0){
      // the caller's reads carry their pool's row of likelihoods (seq_stutter_genotyper.cpp:536-541), the second mate of a pair the sum of
      // the two pools' rows (:551-563): that is how a read of the hint finds its pool(s) among the rows just computed
      std::map<unsigned long long, int> by_row;
      for (size_t i = 0; i < S.n_pools; i++) if (S.seeds[i] >= 0) by_row[fnv(ll.data() + i*(size_t)A, sizeof(double)*(size_t)A, 1469598103934665603ull)] = (int)i;
      int matched = 0;
      for (int r = 0; r < H.n_reads; r++){
        if (H.best[r] < 0) continue;
        const double* row = H.rows.data() + (size_t)r*A;
        std::map<unsigned long long, int>::const_iterator it = by_row.find(fnv(row, sizeof(double)*(size_t)A, 1469598103934665603ull));
        if (it != by_row.end() && memcmp(ll.data() + (size_t)it->second*A, row, sizeof(double)*(size_t)A) == 0){ want(it->second, H.best[r]); matched++; continue; }
        bool found = false;
        for (size_t i = 0; i < S.n_pools && !found; i++){
          if (S.seeds[i] < 0) continue;
          const double* ri = ll.data() + i*(size_t)A;
          for (size_t j = i; j < S.n_pools && !found; j++){
            if (S.seeds[j] < 0) continue;
            const double* rj = ll.data() + j*(size_t)A;
            bool same = true;
            for (int k = 0; k < A && same; k++) same = (ri[k] + rj[k] == row[k]);
            if (same){ want((int)i, H.best[r]); want((int)j, H.best[r]); found = true; matched++; }
          }
        }
      }
      hinted = matched*2 >= H.n_reads;        // (a hint from another locus or round matches nothing: the likelihood-based choice below takes over)
    }
    if (!hinted)
    for (size_t i = 0; i < S.n_pools; i++){
      if (S.seeds[i] < 0) continue;
      int h1 = -1, h2 = -1;
      const double* row = ll.data() + i*(size_t)A;
      for (int k = 0; k < A; k++){
        if (h1 < 0 || row[k] > row[h1]){ h2 = h1; h1 = k; }
        else if (h2 < 0 || row[k] > row[h2]) h2 = k;
      }
      want((int)i, h1); want((int)i, h2);
    }
    std::vector<AlignmentTrace*> made;
    for (size_t i = 0; i < req_read.size(); i++) made.push_back(new AlignmentTrace(fw_haplotype_->num_blocks()));
    run_trace_requests(pools, req_read, req_seed, req_hap, made);
    g_tc_calls++; g_tc_ahead += (long long)made.size();
    for (size_t i = 0; i < made.size(); i++) S.ready[std::make_pair((int)req_read[i], (int)req_hap[i])] = made[i];
    S.primed = true;
    g_tc_misses++;
  }
  else {
    if (S.seeds[p] != seed_base){ S.forget(); return NULL; }
    if (S.ready.find(std::make_pair(p, best_haplotype)) != S.ready.end()) g_tc_hits++;
    else {
      // not among the two likeliest of its pool: every pool against this haplotype, in one call (its sample's other reads will ask for it)
      std::vector<int32_t> req_read, req_seed, req_hap;
      for (size_t i = 0; i < S.n_pools; i++)
        if (S.seeds[i] >= 0 && S.ready.find(std::make_pair((int)i, best_haplotype)) == S.ready.end()){
          req_read.push_back((int32_t)i); req_seed.push_back(S.seeds[i]); req_hap.push_back(best_haplotype);
        }
      std::vector<AlignmentTrace*> made;
      for (size_t i = 0; i < req_read.size(); i++) made.push_back(new AlignmentTrace(fw_haplotype_->num_blocks()));
      run_trace_requests(pools, req_read, req_seed, req_hap, made);
      g_tc_calls++; g_tc_ahead += (long long)made.size(); g_tc_misses++;
      for (size_t i = 0; i < made.size(); i++) S.ready[std::make_pair((int)req_read[i], (int)req_hap[i])] = made[i];
    }
  }
  std::map<std::pair<int,int>, AlignmentTrace*>::iterator it = S.ready.find(std::make_pair(p, best_haplotype));
  if (it == S.ready.end()) return NULL;
  AlignmentTrace* t = it->second;        // the caller's from here on (it keeps its traces in a cache of its own and deletes them)
  S.ready.erase(it);
  return t;
}

AlignmentTrace* HapAlignerMI355X::trace_optimal_aln(const Alignment& orig_aln, int seed_base, int best_haplotype, const BaseQuality* base_quality){
  if (AlignmentTrace* ahead = prefetched_trace(orig_aln, seed_base, best_haplotype)) return ahead;
  std::vector<AlignmentTrace*> traces;
  trace_optimal_alns(std::vector<Alignment>(1, orig_aln), std::vector<int>(1, seed_base), std::vector<int>(1, best_haplotype), base_quality, traces);
  return traces[0];
}

void HapAlignerMI355X::trace_optimal_alns(const std::vector<Alignment>& alignments, const std::vector<int>& seed_bases, const std::vector<int>& best_haplotypes,
					  const BaseQuality* base_quality, std::vector<AlignmentTrace*>& traces){
  assert(alignments.size() == best_haplotypes.size() && (seed_bases.empty() || seed_bases.size() == alignments.size()));
  (void)base_quality;
  traces.clear();
  // A haplotype the aligner was told not to realign to is skipped by process_read (HapAligner.cpp:614-618), which leaves the
  // AlignmentTrace empty; the reference's own callers always trace with an all-true mask (seq_stutter_genotyper.cpp:813).
  {
    bool any_masked = false;
    for (size_t i = 0; i < best_haplotypes.size(); i++) any_masked |= !realign_to_hap_[best_haplotypes[i]];
    if (any_masked){
      std::vector<Alignment> sub_alns; std::vector<int> sub_haps, sub_seeds;
      for (size_t i = 0; i < alignments.size(); i++)
        if (realign_to_hap_[best_haplotypes[i]]){
	  sub_alns.push_back(alignments[i]); sub_haps.push_back(best_haplotypes[i]);
	  if (!seed_bases.empty()) sub_seeds.push_back(seed_bases[i]);
	}
      std::vector<AlignmentTrace*> sub;
      trace_optimal_alns(sub_alns, sub_seeds, sub_haps, base_quality, sub);
      for (size_t i = 0, k = 0; i < alignments.size(); i++)
        traces.push_back(realign_to_hap_[best_haplotypes[i]] ? sub[k++] : new AlignmentTrace(fw_haplotype_->num_blocks()));
      return;
    }
  }
  for (size_t i = 0; i < alignments.size(); i++) traces.push_back(new AlignmentTrace(fw_haplotype_->num_blocks()));
  run_traces(alignments, seed_bases, best_haplotypes, traces);
}

// One hipstr_hmm_trace_seeded call for all requests; targets[i] receives request i.
void HapAlignerMI355X::run_traces(const std::vector<Alignment>& alignments, const std::vector<int>& seed_bases, const std::vector<int>& best_haplotypes,
				  const std::vector<AlignmentTrace*>& targets){
  const int n = (int)alignments.size();
  std::vector<int32_t> req_read(n), req_seed(n, HIPSTR_SEED_AUTO), req_hap(best_haplotypes.begin(), best_haplotypes.end());
  for (int i = 0; i < n; i++){
    req_read[i] = i;
    if (!seed_bases.empty()) req_seed[i] = seed_bases[i];           // the caller's seed_base (HapAligner.h:93), not a recomputed one
  }
  run_trace_requests(alignments, req_read, req_seed, req_hap, targets);
}

void HapAlignerMI355X::run_trace_requests(const std::vector<Alignment>& reads, const std::vector<int32_t>& req_read_in, const std::vector<int32_t>& req_seed_in,
					  const std::vector<int32_t>& req_hap, const std::vector<AlignmentTrace*>& targets){
  if (req_read_in.empty()) return;
  AdapterTimer t_flat(0);
  FlatReads r(reads, std::vector<bool>(reads.size(), true));
  t_flat.stop();
  run_trace_requests(r, req_read_in, req_seed_in, req_hap, targets);
}

void HapAlignerMI355X::run_trace_requests(const FlatReads& r, const std::vector<int32_t>& req_read_in, const std::vector<int32_t>& req_seed_in,
					  const std::vector<int32_t>& req_hap, const std::vector<AlignmentTrace*>& targets){
  const int n = (int)req_read_in.size();
  if (n == 0) return;
  FILL_BATCH(b, r)
  AdapterTimer t_info(2);

  // Haplotype::get_aln_info() of every haplotype, in the order the reference visits them (Haplotype::next)
  std::vector<std::string> aln_info;
  fw_haplotype_->reset();
  do { aln_info.push_back(fw_haplotype_->get_aln_info()); } while (fw_haplotype_->next());
  fw_haplotype_->reset();
  std::vector<const char*> hap_to_ref;
  for (size_t k = 0; k < aln_info.size(); k++) hap_to_ref.push_back(aln_info[k].c_str());

  t_info.stop();
  std::vector<int32_t> req_read(req_read_in), req_allele(req_hap), req_seed(req_seed_in);
  size_t chars = 64;
  for (int i = 0; i < n; i++) chars += 2*(size_t)(r.base_off[req_read[i] + 1] - r.base_off[req_read[i]]) + 2*aln_info[req_hap[i]].size() + 64;
  const int32_t cap = (int32_t)chars;
  std::vector<double> ll(n);
  std::vector<int32_t> max_index(n), hap_aln_off(n+1), stutter_size(n), str_seq_off(n+1), flank_seq_off(2*n+1), flank_ins(n), flank_del(n),
    indel_off(n+1), indel_pos(cap), indel_size(cap), snp_off(n+1), snp_pos(cap), aln_start(n), aln_stop(n), cigar_off(n+1), cigar_len(cap), aln_str_off(n+1);
  std::vector<char> hap_aln(cap), str_seq(cap), flank_seq(cap), snp_base(cap), cigar_op(cap), aln_str(cap);
  hipstr_trace_out_t o;
  o.ll = ll.data(); o.max_index = max_index.data(); o.hap_aln_off = hap_aln_off.data(); o.hap_aln = hap_aln.data();
  o.stutter_size = stutter_size.data(); o.str_seq_off = str_seq_off.data(); o.str_seq = str_seq.data();
  o.flank_seq_off = flank_seq_off.data(); o.flank_seq = flank_seq.data(); o.flank_ins = flank_ins.data(); o.flank_del = flank_del.data();
  o.indel_off = indel_off.data(); o.indel_pos = indel_pos.data(); o.indel_size = indel_size.data();
  o.snp_off = snp_off.data(); o.snp_pos = snp_pos.data(); o.snp_base = snp_base.data();
  o.aln_start = aln_start.data(); o.aln_stop = aln_stop.data(); o.cigar_off = cigar_off.data(); o.cigar_op = cigar_op.data(); o.cigar_len = cigar_len.data();
  o.aln_str_off = aln_str_off.data(); o.aln_str = aln_str.data(); o.cap_chars = cap;
  if (hipstr_hmm_trace_seeded(&b, n, req_read.data(), req_allele.data(), req_seed.data(), hap_to_ref.data(), &o) != 0)
    printErrorAndDie(hipstr_last_error());
  AdapterTimer t_fill(1);
  for (int i = 0; i < n; i++) fill_trace(i, &o, r.quals_of((size_t)req_read[i]), r.seq_of((size_t)req_read[i]), *targets[i]);
}

void HapAlignerMI355X::fill_trace(int i, const hipstr_trace_out_t* o, const std::string& orig_quals, const std::string& orig_seq, AlignmentTrace& t) const {
  std::string s(o->hap_aln + o->hap_aln_off[i], o->hap_aln_off[i+1] - o->hap_aln_off[i]);
  t.set_hap_aln(s);
  if (o->stutter_size[i] != HIPSTR_NO_STR_DATA){
    std::string ss(o->str_seq + o->str_seq_off[i], o->str_seq_off[i+1] - o->str_seq_off[i]);
    t.add_str_data(1, o->stutter_size[i], ss);
  }
  for (int side = 0; side < 2; side++){
    std::string fs(o->flank_seq + o->flank_seq_off[2*i+side], o->flank_seq_off[2*i+side+1] - o->flank_seq_off[2*i+side]);
    t.add_flank_data(side == 0 ? 0 : 2, fs);
  }
  for (int k = 0; k < o->flank_ins[i]; k++) t.inc_flank_ins();
  for (int k = 0; k < o->flank_del[i]; k++) t.inc_flank_del();
  for (int k = o->indel_off[i]; k < o->indel_off[i+1]; k++) t.add_flank_indel(std::pair<int32_t,int32_t>(o->indel_pos[k], o->indel_size[k]));
  for (int k = o->snp_off[i]; k < o->snp_off[i+1]; k++) t.add_flank_snp(o->snp_pos[k], o->snp_base[k]);
  t.traced_aln() = Alignment(o->aln_start[i], o->aln_stop[i], false, "TRACE", orig_quals, orig_seq,
			      std::string(o->aln_str + o->aln_str_off[i], o->aln_str_off[i+1] - o->aln_str_off[i]));
  std::vector<CigarElement> cigar_list;
  for (int k = o->cigar_off[i]; k < o->cigar_off[i+1]; k++) cigar_list.push_back(CigarElement(o->cigar_op[k], o->cigar_len[k]));
  t.traced_aln().set_cigar_list(cigar_list);
}
