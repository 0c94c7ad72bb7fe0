// genotype_flow.cpp — the reference's OWN caller of the hot path, unedited, end to end:
//
//   SeqStutterGenotyper::SeqStutterGenotyper  -> init (ReadPooler, mates; seq_stutter_genotyper.cpp:490-511) + build_haplotype (:422-488)
//   SeqStutterGenotyper::genotype             -> calc_hap_aln_probs + calc_log_sample_posteriors                       (:603-640)
//                                                id_and_align_to_stutter_alleles: retrace, new alleles, align ONLY them  (:570-601, :324-415)
//                                                get_unused_alleles / remove_alleles, twice                            (:229-315, :417-420)
//                                                assemble_flanks: flank variants, realign a subset of pools/reads       (:40-217)
//   [--recompute] recompute_stutter_models    -> EMStutterGenotyper::train on the traced stutter sizes, then genotype() again (:1542-)
//
// on seeded synthetic reads of one STR locus, then a dump of everything the rounds produced.  This file is compiled TWICE by
// `make -C oracle flow`, from the same source:
//   oracle/_ref/libflow_ref.so     — against the reference's translation units only (CPU HapAligner, CPU Genotyper);
//   oracle/_ref/libflow_mi355x.so  — seq_stutter_genotyper.cpp compiled, unedited, with
//                              `-include SeqAlignment/HapAlignerMI355X.h -DHIPSTR_MI355X_AS_HAPALIGNER`, genotyper.cpp with the body of
//                              calc_log_sample_posteriors replaced by integration/genotyper_posteriors_mi355x.inc, linked with
//                              libhipstr_hmm.so: every HMM alignment, traceback and posterior of the flow runs on the MI355X.
// tests/test_genotype_flow.py requires the two dumps to agree (log-likelihoods bit for bit, posteriors to 1e-9).
//
// The parts of SeqStutterGenotyper that need htslib (write_vcf_record, VCF input) are never called; their symbols stay unresolved
// in both libraries (libflow_ref.so / libflow_mi355x.so, loaded with lazy binding by integration/flow_launcher.c), exactly like
// FastaReader in oracle/_ref/libhipstr_ref.so.  TEST INFRASTRUCTURE, not product.
#ifdef HIPSTR_MI355X_NW_PREFETCH
#include "nw_prefetch_mi355x.h"
#endif
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <set>
#include <sstream>
#include <string>
#include <vector>

// read the genotyper's private state after genotype(); access specifiers do not change the layout (test-only TU)
#define private public
#define protected public
#include "seq_stutter_genotyper.h"
#undef private
#undef protected
#include "SeqAlignment/AlignmentModel.h"
#include "SeqAlignment/NeedlemanWunsch.h"
#include "mathops.h"
#include "null_ostream.h"

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed*0x9E3779B97F4A7C15ull + 0x1234567ull) {}
  uint64_t next(){ s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
  double uni(){ return (next() >> 11) * (1.0/9007199254740992.0); }
  int below(int n){ return (int)(next() % (uint64_t)n); }
  int range(int lo, int hi){ return lo + below(hi - lo + 1); }
  char base(){ return "ACGT"[next() & 3]; }
  char other(char c){ char b; do { b = base(); } while (b == c); return b; }
};

struct SimRead { std::string name; int sample; int strand; };

// one read of `hap` (a haplotype = left context + STR allele + right context laid over the chromosome), left-aligned against the
// reference: the repeat-count difference becomes one I or D at the start of the repeat, substitutions become X
Alignment make_read(Rng& rng, const std::string& chrom, int str_start, int str_len, const std::string& motif, int copies, int snp_pos, char snp_base,
                    int read_start, int read_len, const std::string& name){
  const int p = (int)motif.size();
  std::string allele; for (int i = 0; i < copies; i++) allele += motif;
  const int delta = (int)allele.size() - str_len;
  // haplotype sequence and, per haplotype base, its reference coordinate (-1 = inserted)
  std::string hap = chrom.substr(0, str_start) + allele + chrom.substr(str_start + str_len);
  if (snp_pos >= 0) hap[snp_pos < str_start ? snp_pos : snp_pos + delta] = snp_base;
  std::vector<int> coord(hap.size());
  for (int i = 0; i < str_start; i++) coord[i] = i;
  if (delta >= 0){
    for (int i = 0; i < delta; i++) coord[str_start + i] = -1;                         // insertion right at the start of the repeat
    for (size_t i = str_start + delta; i < hap.size(); i++) coord[i] = (int)i - delta;
  } else
    for (size_t i = str_start; i < hap.size(); i++) coord[i] = (int)i - delta;         // deletion of -delta reference bases at the start
  std::string seq, aln, qual;
  std::vector<CigarElement> cigar;
  auto push = [&](char op, int n){
    if (!cigar.empty() && cigar.back().get_type() == op) cigar.back().set_num(cigar.back().get_num() + n);
    else cigar.push_back(CigarElement(op, n));
  };
  int32_t start = -1, last = -1;
  for (int i = read_start; i < read_start + read_len && i < (int)hap.size(); i++){
    char b = hap[i];
    if (rng.uni() < 0.004) b = rng.other(b);
    const int c = coord[i];
    if (c < 0){ if (start < 0) continue; push('I', 1); }
    else {
      if (start < 0) start = c;
      if (last >= 0 && c > last + 1){ push('D', c - last - 1); aln += std::string(c - last - 1, '-'); }
      push(b == chrom[c] ? '=' : 'X', 1);
      last = c;
    }
    seq += b; aln += b;
    const double u = rng.uni();
    qual += u < 0.02 ? '#' : u < 0.10 ? ',' : u < 0.30 ? ':' : 'F';
  }
  Alignment a(start, last + 1, rng.uni() < 0.5, name, qual, seq, aln);
  a.set_cigar_list(cigar);
  a.set_hap_gen_info(std::vector<bool>(1, true));
  (void)p;
  return a;
}

void dump_doubles(FILE* f, const char* key, const double* v, size_t n, bool hex){
  fprintf(f, "%s %zu", key, n);
  for (size_t i = 0; i < n; i++){
    if (hex){ uint64_t u; memcpy(&u, &v[i], 8); fprintf(f, " %016llx", (unsigned long long)u); }
    else fprintf(f, " %.17g", v[i]);
  }
  fprintf(f, "\n");
}

}  // namespace

// one locus: seeded reads -> SeqStutterGenotyper -> genotype() [-> recompute_stutter_models()] -> dump to `f`
struct LocusParams { uint64_t seed; int n_samples, reads_per_sample, period; bool recompute, reassemble, nw; };
#include <atomic>
#include <chrono>
// where a locus' wall time goes inside this driver, summed over threads (nanoseconds): simulating the reads | the reference's constructor +
// genotype() [+ recompute_stutter_models()] | dumping the result.  Only the middle part is the reference's host code (+ the device calls under it).
static std::atomic<long long> g_drv_ns[3];
static int run_locus(const LocusParams& lp, FILE* f){
  const auto t_sim0 = std::chrono::steady_clock::now();
  const uint64_t seed = lp.seed; const int n_samples = lp.n_samples, reads_per_sample = lp.reads_per_sample, period = lp.period;
  const bool recompute = lp.recompute, reassemble = lp.reassemble;
  Rng rng(seed);

  // ---- a chromosome with one pure repeat, flanks that do not continue it
  std::string motif;
  do { motif.clear(); for (int i = 0; i < period; i++) motif += rng.base(); }
  while (period > 1 && motif == std::string(period, motif[0]));
  const int c0 = std::max(3, 40/period), str_start = 600, str_len = c0*period;
  std::string chrom;
  for (int i = 0; i < 1400; i++) chrom += rng.base();
  for (int i = 0; i < str_len; i++) chrom[str_start + i] = motif[i % period];
  if (chrom[str_start - 1] == motif[period - 1]) chrom[str_start - 1] = rng.other(motif[period - 1]);
  if (chrom[str_start + str_len] == motif[0]) chrom[str_start + str_len] = rng.other(motif[0]);
  const int snp_pos = str_start - 18; const char snp_base = rng.other(chrom[snp_pos]);      // a flank variant some haplotypes carry

  // ---- samples, genotypes, reads (grouped by sample; mates share a name and follow each other, seq_stutter_genotyper.cpp:499)
  std::vector<std::string> sample_names;
  std::vector<Alignment> alns;
  std::vector< std::vector<double> > log_p1(n_samples), log_p2(n_samples);
  const int deltas[5] = { 0, -1, 1, 2, -2 };
  for (int s = 0; s < n_samples; s++){
    std::stringstream nm; nm << "S" << s; sample_names.push_back(nm.str());
    int cp[2], snp[2];
    for (int h = 0; h < 2; h++){ cp[h] = c0 + deltas[rng.below(rng.uni() < 0.7 ? 3 : 5)]; snp[h] = (s % 2 == 0 && h == 1) ? 1 : 0; }
    const int n_reads = reads_per_sample + rng.range(-2, 2);
    for (int r = 0; r < n_reads; r++){
      const int h = rng.below(2);
      int copies = cp[h];
      const double u = rng.uni();
      if (u < 0.06) copies += 1; else if (u < 0.14) copies -= 1;                           // PCR stutter
      if (copies < 1) copies = 1;
      const bool spanning = rng.uni() < 0.8;
      // starts and lengths from a coarse grid, so that identical reads occur and ReadPooler has pools to form (read_pooler.cpp:3-20)
      const int read_len = 100 + 5*rng.range(0, 5);
      const int read_start = spanning ? str_start - 22 - 6*rng.range(0, 6) : (rng.uni() < 0.5 ? str_start - rng.range(70, 95) : str_start + rng.range(4, 20));
      std::stringstream rn; rn << "r" << s << "_" << r;
      const bool paired = rng.uni() < 0.2;
      for (int mate = 0; mate < (paired ? 2 : 1); mate++){
        const int st = mate == 0 ? read_start : str_start - rng.range(25, 55);
        alns.push_back(make_read(rng, chrom, str_start, str_len, motif, copies, snp[h] ? snp_pos : -1, snp_base, st, mate == 0 ? read_len : rng.range(100, 120), rn.str()));
        if (rng.uni() < 0.35){ const double good = -rng.uni()*0.05, bad = -2 - rng.uni()*6; log_p1[s].push_back(h == 0 ? good : bad); log_p2[s].push_back(h == 0 ? bad : good); }
        else { log_p1[s].push_back(0.0); log_p2[s].push_back(0.0); }
      }
    }
  }

  // ---- [--nw] the Needleman-Wunsch call of realign() (AlignmentOps.cpp:14-26) for every simulated read: the read against its reference
  // window of ALIGN_WINDOW_WIDTH = 75 bases either side, no end penalty.  (realign() itself takes a BamAlignment, whose constructor
  // needs htslib — not built here; its NeedlemanWunsch::Align call is reproduced with the arguments :15-25 derive.)
  if (lp.nw){
#ifdef HIPSTR_MI355X_NW_PREFETCH
    // the nw variant: what integration/left_align_reads_prepass_mi355x.inc does in front of left_align_reads' loop — every pair of the
    // locus handed over in one call; the Align calls below (unchanged) are then served from the thread's table
    {
      std::vector<std::string> nw_refs, nw_reads;
      for (size_t i = 0; i < alns.size(); i++){
        const int32_t start = std::max(alns[i].get_start() - 75 - 1, 0), stop = std::min(alns[i].get_stop() + 75 - 1, (int32_t)(chrom.size() - 1));
        nw_refs.push_back(chrom.substr(start, stop - start + 1)); nw_reads.push_back(alns[i].get_sequence());
      }
      const int n_pf = hipstr_mi355x_nw_prefetch(nw_refs, nw_reads, false);
      fprintf(stderr, "nw_prefetch %d pairs of %zu reads in one call\n", n_pf, alns.size());
    }
#endif
    for (size_t i = 0; i < alns.size(); i++){
      const int32_t start = std::max(alns[i].get_start() - 75 - 1, 0), stop = std::min(alns[i].get_stop() + 75 - 1, (int32_t)(chrom.size() - 1));
      const std::string ref_seq = chrom.substr(start, stop - start + 1), read_seq = alns[i].get_sequence();
      std::string ref_al, read_al; float score = 0; std::vector<CigarOp> cigar_list;
      const bool aligned = NeedlemanWunsch::Align(ref_seq, read_seq, ref_al, read_al, &score, cigar_list);
      uint32_t sbits; memcpy(&sbits, &score, 4);
      fprintf(f, "realign %zu %d %08x ", i, aligned ? 1 : 0, sbits);
      for (size_t c = 0; c < cigar_list.size(); c++) fprintf(f, "%d%c", cigar_list[c].Length, cigar_list[c].Type);
      fprintf(f, " %s %s\n", ref_al.c_str(), read_al.c_str());
    }
#ifdef HIPSTR_MI355X_NW_PREFETCH
    fprintf(stderr, "nw_prefetch served %lld of %zu Align calls from the table\n", hipstr_mi355x_nw_hits(), alns.size());
#endif
  }

  const auto t_gen0 = std::chrono::steady_clock::now();
  g_drv_ns[0] += std::chrono::duration_cast<std::chrono::nanoseconds>(t_gen0 - t_sim0).count();
  Region region("chr1", str_start, str_start + str_len, period, "LOCUS");
  RegionGroup group(region);
  StutterModel model(0.9, 0.05, 0.05, 0.7, 0.005, 0.005, period);
  std::vector<StutterModel*> models(1, &model);
  std::stringstream log;
  SeqStutterGenotyper g(group, false, reassemble, alns, log_p1, log_p2, sample_names, chrom, models, NULL, log);
  const bool ok = g.genotype(1000, 4, 0.15, log);
  bool ok2 = true;
  if (ok && recompute) ok2 = g.recompute_stutter_models(log, 1000, 4, 0.15, 100, 0.01, 0.001);

  const auto t_dump0 = std::chrono::steady_clock::now();
  g_drv_ns[1] += std::chrono::duration_cast<std::chrono::nanoseconds>(t_dump0 - t_gen0).count();
  struct DumpTimer { std::chrono::steady_clock::time_point t0; ~DumpTimer(){ g_drv_ns[2] += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } } dump_timer{t_dump0};
  fprintf(f, "genotype_ok %d\nrecompute_ok %d\n", ok ? 1 : 0, ok2 ? 1 : 0);
  fprintf(f, "num_reads %u\nnum_samples %d\nnum_pools %d\n", g.num_reads_, g.num_samples_, g.pooler_.num_pools());
  if (ok){
    fprintf(f, "num_alleles %d\n", g.num_alleles_);
    for (int b = 0; b < g.haplotype_->num_blocks(); b++){
      HapBlock* blk = g.haplotype_->get_block(b);
      fprintf(f, "block %d %d %d %d", b, blk->start(), blk->end(), blk->num_options());
      for (int o = 0; o < blk->num_options(); o++) fprintf(f, " %s", blk->get_seq(o).c_str());
      fprintf(f, "\n");
    }
    if (lp.nw)        // Haplotype::aln_haps_to_ref (Haplotype.cpp:58-86): every haplotype against the reference haplotype, end penalty on
      for (size_t h = 0; h < g.haplotype_->hap_aln_info_.size(); h++) fprintf(f, "hap_aln_info %zu %s\n", h, g.haplotype_->hap_aln_info_[h].c_str());
    fprintf(f, "pool_index %u", g.num_reads_); for (unsigned i = 0; i < g.num_reads_; i++) fprintf(f, " %d", g.pool_index_[i]); fprintf(f, "\n");
    fprintf(f, "second_mate %u", g.num_reads_); for (unsigned i = 0; i < g.num_reads_; i++) fprintf(f, " %d", g.second_mate_[i] ? 1 : 0); fprintf(f, "\n");
    fprintf(f, "seed_positions %u", g.num_reads_); for (unsigned i = 0; i < g.num_reads_; i++) fprintf(f, " %d", g.seed_positions_[i]); fprintf(f, "\n");
    std::vector<Alignment>& pooled = g.pooler_.get_alignments();
    for (size_t i = 0; i < pooled.size(); i++) fprintf(f, "pool_qual %zu %s\n", i, pooled[i].get_base_qualities().c_str());
    dump_doubles(f, "log_aln_probs", g.log_aln_probs_, (size_t)g.num_reads_*g.num_alleles_, true);
    dump_doubles(f, "log_sample_posteriors", g.log_sample_posteriors_, (size_t)g.num_samples_*g.num_alleles_*g.num_alleles_, false);
    dump_doubles(f, "sample_total_LLs", g.sample_total_LLs_, (size_t)g.num_samples_, false);
    std::vector< std::pair<int,int> > gts;
    g.get_optimal_haplotypes(gts);
    fprintf(f, "map_haplotypes %zu", gts.size()); for (size_t i = 0; i < gts.size(); i++) fprintf(f, " %d|%d", gts[i].first, gts[i].second); fprintf(f, "\n");
    // the tracebacks the last round left in the cache: (pool, haplotype) -> alignment
    for (std::map<std::pair<int,int>, AlignmentTrace*>::iterator it = g.trace_cache_.begin(); it != g.trace_cache_.end(); ++it){
      AlignmentTrace* t = it->second;
      fprintf(f, "trace %d %d %s %d %s %d %d %s\n", it->first.first, it->first.second, t->hap_aln().c_str(),
              t->str_data_[1] != NULL ? t->stutter_size(1) : -100000, t->traced_aln().getCigarString().c_str(), t->traced_aln().get_start(), t->traced_aln().get_stop(),
              t->str_data_[1] != NULL ? t->str_seq(1).c_str() : "-");
    }
    if (recompute && ok2){
      StutterModel* m = g.hap_blocks_[1]->get_repeat_info()->get_stutter_model();
      double sp[6] = { m->get_parameter(true, 'P'), m->get_parameter(true, 'U'), m->get_parameter(true, 'D'), m->get_parameter(false, 'P'), m->get_parameter(false, 'U'), m->get_parameter(false, 'D') };
      dump_doubles(f, "stutter_model", sp, 6, false);
    }
  }
  // what the rounds did, from the genotyper's own log (counts only: "Recomputing sample posteriors after removing N ...")
  {
    std::string line; std::istringstream in(log.str()); int n = 0;
    while (std::getline(in, line))
      if (line.find("Recomputing") != std::string::npos || line.find("Identified") != std::string::npos || line.find("Aborting") != std::string::npos || line.find("candidate") != std::string::npos)
        fprintf(f, "log %d %s\n", n++, line.c_str());
  }
  return ok ? 0 : 1;
}

#ifdef FLOW_MI355X
#include "hipstr_hmm.h"
#include "hipstr_hmm_debug.h"      // hipstr_debug_api_profile: what --profile prints
#include "SeqAlignment/HapAlignerMI355X.h"
#endif
#include <atomic>
#include <chrono>
#include <thread>

extern "C" int flow_main(int argc, char** argv){
  LocusParams lp; lp.seed = 1; lp.n_samples = 10; lp.reads_per_sample = 9; lp.period = 4; lp.recompute = false; lp.reassemble = true; lp.nw = false;
  const char* out_path = NULL; int n_loci = 0, n_threads = 1; bool use_stream = false, profile = false;
  for (int i = 1; i < argc; i++){
    if (!strcmp(argv[i], "--seed")) lp.seed = strtoull(argv[++i], NULL, 10);
    else if (!strcmp(argv[i], "--samples")) lp.n_samples = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--reads")) lp.reads_per_sample = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--period")) lp.period = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--recompute")) lp.recompute = true;
    else if (!strcmp(argv[i], "--no-flanks")) lp.reassemble = false;
    else if (!strcmp(argv[i], "--nw")) lp.nw = true;                        // also dump the Needleman-Wunsch results of the locus (realign()'s call, aln_haps_to_ref)
    else if (!strcmp(argv[i], "--out")) out_path = argv[++i];
    else if (!strcmp(argv[i], "--loci")) n_loci = atoi(argv[++i]);          // many loci (seeds seed, seed+1, ...; periods cycling 2..5) ...
    else if (!strcmp(argv[i], "--threads")) n_threads = atoi(argv[++i]);    // ... one SeqStutterGenotyper per host thread at a time
    else if (!strcmp(argv[i], "--profile")) profile = true;                 // many-loci form: where the host threads' time went (MI355X build: library and adapter buckets)
    else if (!strcmp(argv[i], "--stream")) use_stream = true;               // MI355X build: alignment rounds of the loci in flight share batches
    else { fprintf(stderr, "unknown argument %s\n", argv[i]); return 2; }
  }
  precompute_integer_logs();                 // hipstr_main.cpp:352
  init_alignment_model();
  if (n_loci <= 0){
    FILE* f = out_path ? fopen(out_path, "w") : stdout;
    if (!f){ perror(out_path); return 2; }
    const int rc = run_locus(lp, f);
    if (out_path) fclose(f);
    return rc;
  }
  // ---- throughput form: n_loci loci through genotype(), n_threads at a time; per-locus dumps are kept and hashed in locus order
#ifdef FLOW_MI355X
  hipstr_stream_t* stream = NULL;
  if (use_stream){
    hipstr_stream_opts_t o; memset(&o, 0, sizeof o);
    stream = hipstr_stream_open(&o);
    if (!stream){ fprintf(stderr, "%s\n", hipstr_last_error()); return 2; }
    HapAlignerMI355X::use_stream(stream);
  }
#else
  (void)use_stream;
#endif
#ifdef FLOW_MI355X
  if (profile){
    // one locus first, outside the measurement: device initialisation, module load, the first allocations
    LocusParams q = lp; q.period = 2; char* buf = NULL; size_t len = 0; FILE* wf = open_memstream(&buf, &len); run_locus(q, wf); fclose(wf); free(buf);
    hipstr_debug_api_profile(1, 0, NULL, NULL, NULL); HapAlignerMI355X::profile(true, NULL);
  }
#endif
  for (int i = 0; i < 3; i++) g_drv_ns[i] = 0;
  std::vector<std::string> dumps(n_loci);
  std::vector<int> rcs(n_loci, 0);
  std::atomic<int> next(0);
  const auto t0 = std::chrono::steady_clock::now();
  auto work = [&](){
    for (int l = next.fetch_add(1); l < n_loci; l = next.fetch_add(1)){
      LocusParams q = lp; q.seed = lp.seed + (uint64_t)l; q.period = 2 + (l % 4);
      char* buf = NULL; size_t len = 0;
      FILE* f = open_memstream(&buf, &len);
      rcs[l] = run_locus(q, f);
      fclose(f);
      dumps[l].assign(buf, len); free(buf);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < n_threads; t++) pool.emplace_back(work);
  work();
  for (std::thread& t : pool) t.join();
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
#ifdef FLOW_MI355X
  if (stream){ HapAlignerMI355X::use_stream(NULL); hipstr_stream_close(stream); }
#endif
  // digest of what cannot differ between the CPU and the MI355X run: everything except the posterior lines (device exp/log: 1e-13)
  uint64_t h = 1469598103934665603ull; int n_ok = 0;
  for (int l = 0; l < n_loci; l++){
    n_ok += rcs[l] == 0;
    std::istringstream in(dumps[l]); std::string line;
    while (std::getline(in, line)){
      if (line.compare(0, 21, "log_sample_posteriors") == 0 || line.compare(0, 16, "sample_total_LLs") == 0) continue;
      for (char ch : line){ h ^= (uint8_t)ch; h *= 1099511628211ull; }
    }
  }
  if (profile){
    // thread seconds = threads x wall; what is not inside the library or the adapter is the reference's own host code
    // (SeqStutterGenotyper, ReadPooler, HaplotypeGenerator, Haplotype ...) and this driver's read simulation
    fprintf(stderr, "profile: %d loci, %d thread(s), wall %.3f s, thread seconds %.3f\n", n_loci, n_threads, dt, dt*n_threads);
#ifdef FLOW_MI355X
    const char* names[32]; double secs[32]; int64_t calls[32]; double ad[3];
    const int nb = hipstr_debug_api_profile(0, 32, names, secs, calls);
    HapAlignerMI355X::profile(false, ad);
    double inside = ad[0] + ad[1] + ad[2];
    for (int i = 0; i < nb; i++){
      fprintf(stderr, "profile: %-42s %9.3f ms %8lld calls\n", names[i], 1e3*secs[i], (long long)calls[i]);
      if (names[i][0] != ' ') inside += secs[i];
    }
    fprintf(stderr, "profile: %-42s %9.3f ms\nprofile: %-42s %9.3f ms\nprofile: %-42s %9.3f ms\n", "adapter: flatten reads + batch struct", 1e3*ad[0],
            "adapter: fill AlignmentTrace objects", 1e3*ad[1], "adapter: haplotype alignment strings", 1e3*ad[2]);
    const double drv_sim = 1e-9*(double)g_drv_ns[0].load(), drv_gen = 1e-9*(double)g_drv_ns[1].load(), drv_dump = 1e-9*(double)g_drv_ns[2].load();
    fprintf(stderr, "profile: %-42s %9.3f ms\nprofile: %-42s %9.3f ms\n", "driver: read simulation (not the reference)", 1e3*drv_sim, "driver: result dump (not the reference)", 1e3*drv_dump);
    fprintf(stderr, "profile: %-42s %9.3f ms\n", "constructor + genotype() [+ recompute]", 1e3*drv_gen);
    fprintf(stderr, "profile: %-42s %9.3f ms\n", "  of it: reference host code", 1e3*(drv_gen - inside));
    long long tc[4]; HapAlignerMI355X::trace_cache_stats(tc);
    fprintf(stderr, "profile: trace_optimal_aln prefetch: %lld requests served from it, %lld went to the device in %lld calls (forward + traceback), %lld tracebacks computed ahead\n",
            tc[0], tc[1], tc[2], tc[3]);
#else
    fprintf(stderr, "profile: %-42s %9.3f ms\nprofile: %-42s %9.3f ms\nprofile: %-42s %9.3f ms\n", "driver: read simulation (not the reference)", 1e-6*(double)g_drv_ns[0].load(),
            "constructor + genotype() [+ recompute]", 1e-6*(double)g_drv_ns[1].load(), "driver: result dump (not the reference)", 1e-6*(double)g_drv_ns[2].load());
#endif
  }
  FILE* f = out_path ? fopen(out_path, "w") : stdout;
  if (!f){ perror(out_path); return 2; }
  // (genotype_loci_per_s: the same with the driver's own work — simulating reads, dumping results — taken out of every thread's time)
  const double drv_own = 1e-9*(double)(g_drv_ns[0].load() + g_drv_ns[2].load()) / n_threads;
  fprintf(f, "{\"loci\": %d, \"genotyped\": %d, \"threads\": %d, \"stream\": %d, \"seconds\": %.6f, \"loci_per_s\": %.3f, \"genotype_loci_per_s\": %.3f, \"digest\": \"%016llx\"}\n",
          n_loci, n_ok, n_threads, use_stream ? 1 : 0, dt, n_loci / dt, n_loci / std::max(1e-9, dt - drv_own), (unsigned long long)h);
  if (out_path) fclose(f);
  return 0;
}
