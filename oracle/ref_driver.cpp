/*
 * ref_driver.cpp — TEST INFRASTRUCTURE ONLY.
 *
 * Thin C entry points around the *real* HipSTR v0.7 classes, compiled together
 * with the reference's own translation units straight from /root/reference (see
 * oracle/Makefile; nothing from the reference is copied into this repository).
 * The resulting oracle/_ref/libhipstr_ref.so is used to
 *   (1) pin the C restatement in oracle/hipstr_oracle.c,
 *   (2) generate the golden fixtures under tests/golden/,
 *   (3) serve as the "reference" CPU baseline in bench.py.
 * It is never linked or loaded by the product library.
 *
 * The driver only rebuilds the reference's objects (HapBlock/RepeatBlock/
 * Haplotype/Alignment/BaseQuality/StutterModel) from the flat hipstr_batch_t and
 * calls the reference's public API.
 */
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "SeqAlignment/AlignmentData.h"
#include "SeqAlignment/AlignmentModel.h"
// AlignmentTrace offers no "is the STR block set?" accessor (stutter_size()/str_seq() assert instead), so the test driver
// reads str_data_ directly.  Access specifiers do not change the class layout, and only this TU is compiled this way.
// HapAligner::trace_optimal_aln drops the likelihood process_read hands it; to pin the traced alignment's score the driver
// positions the aligner's two haplotypes the same way and calls the (public) process_read itself.
#define private public
#include "SeqAlignment/AlignmentTraceback.h"
#include "SeqAlignment/HapAligner.h"
#undef private
#include "SeqAlignment/HapBlock.h"
#include "SeqAlignment/NeedlemanWunsch.h"
#include "SeqAlignment/Haplotype.h"
#include "SeqAlignment/RepeatBlock.h"
#include "base_quality.h"
#include "read_pooler.h"
#include "genotyper.h"
#include "em_stutter_genotyper.h"
#include "mathops.h"
#include "stutter_model.h"
#include "fastonebigheader.h"

#include "../include/hipstr_hmm.h"

static bool g_ready = false;
static void ensure_ready(){
  if (!g_ready){
    precompute_integer_logs();   // hipstr_main.cpp:352
    init_alignment_model();      // seq_stutter_genotyper.cpp:631
    g_ready = true;
  }
}

namespace {
struct RefLocus {
  std::vector<HapBlock*> blocks;
  Haplotype* hap;
  StutterModel* model;
  RefLocus() : hap(NULL), model(NULL) {}
  ~RefLocus(){
    delete hap;
    for (size_t i = 0; i < blocks.size(); i++) delete blocks[i];
    delete model;
  }
};

// Rebuild the Haplotype of locus l; opt_cursor walks the flat option table.
void build_locus(const hipstr_batch_t* b, int l, int& opt_cursor, RefLocus& out){
  const double* sp = b->stutter + 6*l;
  out.model = new StutterModel(sp[0], sp[1], sp[2], sp[3], sp[4], sp[5], b->period[l]);
  for (int blk = 0; blk < 3; blk++){
    int nopts = b->blk_nopts[3*l+blk];
    std::vector<std::string> seqs;
    for (int o = 0; o < nopts; o++, opt_cursor++)
      seqs.push_back(std::string(b->seq + b->opt_off[opt_cursor], b->opt_off[opt_cursor+1]-b->opt_off[opt_cursor]));
    HapBlock* hb;
    if (blk == 1)
      hb = new RepeatBlock(b->blk_start[3*l+blk], b->blk_end[3*l+blk], seqs[0], b->period[l], out.model);
    else
      hb = new HapBlock(b->blk_start[3*l+blk], b->blk_end[3*l+blk], seqs[0]);
    for (int o = 1; o < nopts; o++)
      hb->add_alternate(seqs[o]);
    out.blocks.push_back(hb);
  }
  out.hap = new Haplotype(out.blocks);
}

Alignment make_alignment(const hipstr_batch_t* b, int r){
  int len = b->base_off[r+1]-b->base_off[r];
  std::string seq(b->bases + b->base_off[r], len), qual(b->quals + b->base_off[r], len);
  int32_t stop = b->read_start[r];
  Alignment aln(b->read_start[r], 0, false, "R", qual, seq, "");
  for (int c = b->cigar_off[r]; c < b->cigar_off[r+1]; c++){
    aln.add_cigar_element(CigarElement(b->cigar_op[c], b->cigar_len[c]));
    if (b->cigar_op[c] != 'I') stop += b->cigar_len[c];
  }
  aln.set_stop(stop);
  return aln;
}
} // namespace

extern "C" {

int ref_process_reads(const hipstr_batch_t* b, double* aln_probs, int32_t* seeds){
  ensure_ready();
  BaseQuality bq;
  int opt_cursor = 0;
  int64_t out_off = 0;
  for (int l = 0; l < b->n_loci; l++){
    RefLocus loc;
    build_locus(b, l, opt_cursor, loc);
    int A = loc.hap->num_combs();
    if (A != b->hap_off[l+1]-b->hap_off[l]){ fprintf(stderr, "ref_driver: num_combs mismatch\n"); return 1; }
    std::vector<bool> realign_hap(A, true);
    if (b->realign_hap) for (int k = 0; k < A; k++) realign_hap[k] = b->realign_hap[b->hap_off[l]+k] != 0;
    int r0 = b->read_off[l], r1 = b->read_off[l+1];
    std::vector<Alignment> alns;
    std::vector<bool> realign_read(r1-r0, true);
    for (int r = r0; r < r1; r++){
      alns.push_back(make_alignment(b, r));
      if (b->realign_read) realign_read[r-r0] = b->realign_read[r] != 0;
    }
    HapAligner aligner(loc.hap, realign_hap);
    aligner.process_reads(alns, 0, &bq, realign_read, aln_probs + out_off, seeds + r0);
    out_off += (int64_t)(r1-r0)*A;
  }
  return 0;
}

/* HapAligner::process_read (HapAligner.h:83) per read with the seed the caller names: seed_in[r] >= 0 is passed as seed_base,
 * -1 gives the row of zeros process_reads writes for such reads, HIPSTR_SEED_AUTO calls calc_seed_base first. */
int ref_process_reads_seeded(const hipstr_batch_t* b, const int32_t* seed_in, double* aln_probs, int32_t* seeds){
  ensure_ready();
  BaseQuality bq;
  int opt_cursor = 0;
  int64_t out_off = 0;
  for (int l = 0; l < b->n_loci; l++){
    RefLocus loc;
    build_locus(b, l, opt_cursor, loc);
    int A = loc.hap->num_combs();
    std::vector<bool> realign_hap(A, true);
    if (b->realign_hap) for (int k = 0; k < A; k++) realign_hap[k] = b->realign_hap[b->hap_off[l]+k] != 0;
    HapAligner aligner(loc.hap, realign_hap);
    for (int r = b->read_off[l]; r < b->read_off[l+1]; r++){
      double* row = aln_probs + out_off + (int64_t)(r - b->read_off[l])*A;
      if (b->realign_read && !b->realign_read[r]) continue;
      Alignment aln = make_alignment(b, r);
      int seed = (seed_in && seed_in[r] != HIPSTR_SEED_AUTO) ? seed_in[r] : aligner.calc_seed_base(aln);
      seeds[r] = seed;
      if (seed == -1){ for (int k = 0; k < A; k++) row[k] = 0; continue; }
      AlignmentTrace trace(loc.hap->num_blocks());
      aligner.process_read(aln, seed, &bq, false, row, trace);
    }
    out_off += (int64_t)(b->read_off[l+1] - b->read_off[l])*A;
  }
  return 0;
}

/* Haplotype sequence of allele k of locus l in Haplotype::next() order, NUL terminated (for pinning the Gray code). */
int ref_hap_sequences(const hipstr_batch_t* b, int l_want, char* out, int out_cap, int32_t* lens){
  ensure_ready();
  int opt_cursor = 0;
  for (int l = 0; l < b->n_loci; l++){
    RefLocus loc;
    build_locus(b, l, opt_cursor, loc);
    if (l != l_want) continue;
    int pos = 0, k = 0;
    do {
      std::string s = loc.hap->get_seq();
      if (pos + (int)s.size() > out_cap) return 1;
      memcpy(out+pos, s.data(), s.size());
      pos += s.size();
      lens[k++] = s.size();
    } while (loc.hap->next());
    return 0;
  }
  return 1;
}

/* ---- scalar probes used to pin host tables and the float LSE approximations ---- */
double ref_int_log(int v){ ensure_ready(); return int_log(v); }
double ref_log_thresh(){ return LOG_THRESH; }
double ref_log_one_half(){ return LOG_ONE_HALF; }
double ref_transition(int which, int h){
  ensure_ready();
  switch (which){
    case 0: return LOG_MATCH_TO_MATCH[h];
    case 1: return LOG_MATCH_TO_INS[h];
    case 2: return LOG_MATCH_TO_DEL[h];
    case 3: return LOG_INS_TO_INS;
    case 4: return LOG_INS_TO_MATCH;
    case 5: return LOG_DEL_TO_DEL;
    default: return LOG_DEL_TO_MATCH;
  }
}
double ref_base_quality(int qual_char, int correct){
  BaseQuality bq;
  return correct ? bq.log_prob_correct((char)qual_char) : bq.log_prob_error((char)qual_char);
}
double ref_stutter_pmf(const double* sp, int period, int sample_bps, int read_bps){
  StutterModel m(sp[0], sp[1], sp[2], sp[3], sp[4], sp[5], period);
  return m.log_stutter_pmf(sample_bps, read_bps);
}
double ref_fast_lse_vec(const double* v, int n){
  std::vector<double> vals(v, v+n);
  return fast_log_sum_exp(vals);
}
double ref_fast_lse2(double a, double b){ return fast_log_sum_exp(a, b); }
double ref_log_sum_exp(const double* v, int n){ return log_sum_exp(v, v+n); }

} // extern "C"

/* ---- posteriors: a subclass that exposes the protected members of Genotyper ---- */
namespace {
class ProbeGenotyper : public Genotyper {
  const double* custom_prior_;
 public:
  void set_prior(const double* p){ custom_prior_ = p; }
  // the virtual hook EMStutterGenotyper overrides (genotyper.h:69)
  void init_log_sample_priors(double* log_sample_ptr){
    if (custom_prior_ == NULL){ Genotyper::init_log_sample_priors(log_sample_ptr); return; }
    memcpy(log_sample_ptr, custom_prior_, sizeof(double)*(size_t)num_samples_*num_alleles_*num_alleles_);
  }
  ProbeGenotyper(bool haploid, const std::vector<std::string>& names,
                 const std::vector< std::vector<double> >& p1, const std::vector< std::vector<double> >& p2, int num_alleles)
    : Genotyper(haploid, names, p1, p2){
    custom_prior_          = NULL;
    num_alleles_           = num_alleles;
    log_sample_posteriors_ = new double[(size_t)num_samples_*num_alleles_*num_alleles_];
    log_aln_probs_         = new double[(size_t)num_reads_*num_alleles_];
  }
  double run(const double* LL, const int32_t* weights, double* post, double* totals, int32_t* map_gt){
    memcpy(log_aln_probs_, LL, sizeof(double)*(size_t)num_reads_*num_alleles_);
    std::vector<int> w(weights, weights+num_reads_);
    double total = calc_log_sample_posteriors(w);
    memcpy(post,   log_sample_posteriors_, sizeof(double)*(size_t)num_samples_*num_alleles_*num_alleles_);
    memcpy(totals, sample_total_LLs_,      sizeof(double)*num_samples_);
    std::vector< std::pair<int,int> > gts;
    get_optimal_haplotypes(gts);
    for (int s = 0; s < num_samples_; s++){ map_gt[2*s] = gts[s].first; map_gt[2*s+1] = gts[s].second; }
    return total;
  }
  // Genotyper::extract_genotypes_and_likelihoods (protected, genotyper.h:98-106) on the posteriors of the last run()
  void extract(int num_variants, const int32_t* h2a, const hipstr_gt_request_t* rq, hipstr_gt_out_t* o, int samp_off, int64_t& g, int64_t& pg){
    std::vector<int> hap_to_allele(h2a, h2a + num_alleles_);
    std::vector< std::pair<int,int> > haps, gts;
    std::vector<double> lp, lu, hlp, hlu, gl_diffs;
    std::vector< std::vector<double> > gls, pgls;
    std::vector< std::vector<int> > pls;
    extract_genotypes_and_likelihoods(num_variants, hap_to_allele, haps, gts, lp, lu, hlp, hlu, rq->calc_gls != 0, gls, gl_diffs,
                                      rq->calc_pls != 0, pls, rq->calc_phased_gls != 0, pgls);
    for (int s = 0; s < num_samples_; s++){
      const int so = samp_off + s;
      o->best_hap[2*so] = haps[s].first; o->best_hap[2*so+1] = haps[s].second;
      o->best_gt[2*so] = gts[s].first; o->best_gt[2*so+1] = gts[s].second;
      o->log_phased_post[so] = lp[s]; o->log_unphased_post[so] = lu[s];
      o->hap_log_phased_post[so] = hlp[s]; o->hap_log_unphased_post[so] = hlu[s];
      if (!gl_diffs.empty()) o->gl_diff[so] = gl_diffs[s];
      const int ngl = haploid_ ? num_variants : num_variants*(num_variants+1)/2, npgl = haploid_ ? num_variants : num_variants*num_variants;
      if (rq->calc_gls) for (int i = 0; i < ngl; i++) o->gls[g + i] = gls[s][i];
      if (rq->calc_pls) for (int i = 0; i < ngl; i++) o->pls[g + i] = pls[s][i];
      if (rq->calc_phased_gls) for (int i = 0; i < npgl; i++) o->phased_gls[pg + i] = pgls[s][i];
      g += ngl; pg += npgl;
    }
  }
};
} // namespace

extern "C" int ref_posteriors(const hipstr_post_batch_t* pb, double* log_post, double* sample_total_ll,
                              int32_t* map_gt, double* locus_total_ll){
  ensure_ready();
  int64_t post_off = 0, samp_off = 0, ll_off = 0;
  for (int l = 0; l < pb->n_loci; l++){
    int A = pb->n_alleles[l], S = pb->n_samples[l];
    int r0 = pb->read_off[l], r1 = pb->read_off[l+1];
    std::vector<std::string> names;
    std::vector< std::vector<double> > p1(S), p2(S);
    for (int s = 0; s < S; s++){ char buf[32]; snprintf(buf, sizeof buf, "S%d", s); names.push_back(buf); }
    int prev = 0;
    for (int r = r0; r < r1; r++){
      int s = pb->sample_label[r];
      if (s < prev || s >= S){ fprintf(stderr, "ref_driver: reads must be grouped by ascending sample\n"); return 1; }
      prev = s;
      p1[s].push_back(pb->log_p1[r]);
      p2[s].push_back(pb->log_p2[r]);
    }
    ProbeGenotyper g(pb->haploid ? pb->haploid[l] != 0 : false, names, p1, p2, A);
    if (pb->log_prior) g.set_prior(pb->log_prior + post_off);
    locus_total_ll[l] = g.run(pb->log_aln_probs + ll_off, pb->read_weight + r0, log_post + post_off,
                              sample_total_ll + samp_off, map_gt + 2*samp_off);
    post_off += (int64_t)S*A*A;
    samp_off += S;
    ll_off   += (int64_t)(r1-r0)*A;
  }
  return 0;
}

extern "C" int ref_gt_extract(const hipstr_post_batch_t* pb, const hipstr_gt_request_t* rq, hipstr_gt_out_t* o){
  ensure_ready();
  int64_t ll_off = 0, g = 0, pg = 0; int samp_off = 0, map_off = 0;
  for (int l = 0; l < pb->n_loci; l++){
    int A = pb->n_alleles[l], S = pb->n_samples[l];
    int r0 = pb->read_off[l], r1 = pb->read_off[l+1];
    std::vector<std::string> names;
    std::vector< std::vector<double> > p1(S), p2(S);
    for (int s = 0; s < S; s++){ char buf[32]; snprintf(buf, sizeof buf, "S%d", s); names.push_back(buf); }
    for (int r = r0; r < r1; r++){ p1[pb->sample_label[r]].push_back(pb->log_p1[r]); p2[pb->sample_label[r]].push_back(pb->log_p2[r]); }
    ProbeGenotyper gt(pb->haploid ? pb->haploid[l] != 0 : false, names, p1, p2, A);
    std::vector<double> post((size_t)S*A*A), totals(S); std::vector<int32_t> map_gt(2*S);
    gt.run(pb->log_aln_probs + ll_off, pb->read_weight + r0, post.data(), totals.data(), map_gt.data());
    gt.extract(rq->n_variants[l], rq->hap_to_allele + map_off, rq, o, samp_off, g, pg);
    samp_off += S; map_off += A; ll_off += (int64_t)(r1-r0)*A;
  }
  return 0;
}

/* ---- Needleman-Wunsch: NeedlemanWunsch::Align on the reference's own code ---- */
extern "C" int ref_nw_align(const hipstr_nw_batch_t* nb, hipstr_nw_out_t* o){
  o->aln_off[0] = 0; o->cigar_off[0] = 0;
  for (int i = 0; i < nb->n_pairs; i++){
    const std::string ref(nb->ref_seqs + nb->ref_off[i], nb->ref_off[i+1] - nb->ref_off[i]);
    const std::string rd(nb->read_seqs + nb->read_off[i], nb->read_off[i+1] - nb->read_off[i]);
    std::string ra, qa; float score = 0; std::vector<CigarOp> cig;
    const bool ok = NeedlemanWunsch::Align(ref, rd, ra, qa, &score, cig, nb->use_ref_end_penalty != 0);
    o->score[i] = score; o->ok[i] = ok ? 1 : 0;
    if (o->aln_off[i] + (int64_t)ra.size() > o->cap_aln || o->cigar_off[i] + (int64_t)cig.size() > o->cap_cigar) return 3;
    memcpy(o->ref_al + o->aln_off[i], ra.data(), ra.size()); memcpy(o->read_al + o->aln_off[i], qa.data(), qa.size());
    o->aln_off[i+1] = o->aln_off[i] + (int64_t)ra.size();
    for (size_t k = 0; k < cig.size(); k++){ o->cigar_op[o->cigar_off[i] + k] = cig[k].Type; o->cigar_len[o->cigar_off[i] + k] = cig[k].Length; }
    o->cigar_off[i+1] = o->cigar_off[i] + (int64_t)cig.size();
  }
  return 0;
}

/* ---- EM stutter training: EMStutterGenotyper::train on the reference's own class ---- */
namespace {
class EMProbe : public EMStutterGenotyper {       // reads the protected per-sample totals the last E-step left behind
 public:
  EMProbe(bool haploid, int motif, const std::vector< std::vector<int> >& bps, const std::vector< std::vector<double> >& p1,
          const std::vector< std::vector<double> >& p2, const std::vector<std::string>& names, int ref_allele)
    : EMStutterGenotyper(haploid, motif, bps, p1, p2, names, ref_allele) {}
  double last_total() const { double t = 0; for (int s = 0; s < num_samples_; s++) t += sample_total_LLs_[s]; return t; }   // genotyper.cpp:75
};
}
extern "C" int ref_em_train(const hipstr_em_batch_t* eb, uint8_t* trained, double* stutter, int32_t* n_iter, double* final_ll){
  ensure_ready();
  for (int l = 0; l < eb->n_loci; l++){
    const int S = eb->n_samples[l], r0 = eb->read_off[l], r1 = eb->read_off[l+1];
    std::vector< std::vector<int> > bps(S);
    std::vector< std::vector<double> > p1(S), p2(S);
    std::vector<std::string> names;
    for (int s = 0; s < S; s++){ char buf[32]; snprintf(buf, sizeof buf, "S%d", s); names.push_back(buf); }
    for (int r = r0; r < r1; r++){
      const int s = eb->sample_label[r];
      bps[s].push_back(eb->num_bps[r]); p1[s].push_back(eb->log_p1[r]); p2[s].push_back(eb->log_p2[r]);
    }
    EMProbe g(eb->haploid ? eb->haploid[l] != 0 : false, eb->period[l], bps, p1, p2, names, eb->ref_allele);
    std::ostringstream log;
    const bool ok = g.train(eb->max_iter, eb->min_ll_abs_change, eb->min_ll_frac_change, true, log);
    trained[l] = ok ? 1 : 0;
    StutterModel* m = g.get_stutter_model();
    stutter[6*l+0] = m->get_parameter(true, 'P');  stutter[6*l+1] = m->get_parameter(true, 'U');  stutter[6*l+2] = m->get_parameter(true, 'D');
    stutter[6*l+3] = m->get_parameter(false, 'P'); stutter[6*l+4] = m->get_parameter(false, 'U'); stutter[6*l+5] = m->get_parameter(false, 'D');
    // iteration count and last LL: train() keeps them in locals but prints them when disp_stats is on ("Iteration k: LL = x")
    int it = 0; double ll = 0;
    const std::string txt = log.str();
    for (size_t pos = txt.find("Iteration "); pos != std::string::npos; pos = txt.find("Iteration ", pos + 1)){
      int k; double v;
      if (sscanf(txt.c_str() + pos, "Iteration %d: LL = %lf", &k, &v) == 2){ it = k; ll = v; }
    }
    n_iter[l] = it;
    // the stream prints 6 significant digits; the exact value of the last E-step is still in the object (genotyper.cpp:75)
    final_ll[l] = g.last_total(); (void)ll;
  }
  return 0;
}

/* ---- traceback: HapAligner::trace_optimal_aln on the reference's own objects, flattened into hipstr_trace_out_t ---- */
namespace {
bool put_str(char* pool, int32_t* off, int idx, const std::string& s, int cap){
  if (off[idx] + (int)s.size() > cap) return false;
  memcpy(pool + off[idx], s.data(), s.size());
  off[idx+1] = off[idx] + (int)s.size();
  return true;
}
}

extern "C" int ref_trace_seeded(const hipstr_batch_t* b, int32_t n_req, const int32_t* req_read, const int32_t* req_allele,
                                const int32_t* req_seed, hipstr_trace_out_t* o);
extern "C" int ref_trace(const hipstr_batch_t* b, int32_t n_req, const int32_t* req_read, const int32_t* req_allele,
                         hipstr_trace_out_t* o){
  return ref_trace_seeded(b, n_req, req_read, req_allele, NULL, o);
}
/* req_seed: trace_optimal_aln's seed_base argument (HapAligner.h:93); NULL / HIPSTR_SEED_AUTO entries use calc_seed_base */
extern "C" int ref_trace_seeded(const hipstr_batch_t* b, int32_t n_req, const int32_t* req_read, const int32_t* req_allele,
                                const int32_t* req_seed, hipstr_trace_out_t* o){
  ensure_ready();
  if (b->n_loci != 1) return 1;
  BaseQuality bq;
  int opt_cursor = 0;
  RefLocus loc;
  build_locus(b, 0, opt_cursor, loc);
  const int A = loc.hap->num_combs();
  std::vector<bool> realign_hap(A, true);
  HapAligner aligner(loc.hap, realign_hap);
  o->hap_aln_off[0] = o->str_seq_off[0] = o->flank_seq_off[0] = o->indel_off[0] = o->snp_off[0] = 0;
  o->cigar_off[0] = o->aln_str_off[0] = 0;
  for (int q = 0; q < n_req; q++){
    Alignment aln = make_alignment(b, req_read[q]);
    int seed = (req_seed && req_seed[q] != HIPSTR_SEED_AUTO) ? req_seed[q] : aligner.calc_seed_base(aln);
    if (seed < 0) return 2;
    AlignmentTrace* tr = aligner.trace_optimal_aln(aln, seed, req_allele[q], &bq);
    {   /* the score of the same alignment: haplotypes positioned as trace_optimal_aln does (HapAligner.cpp:711-722) */
      double prob = 0;
      AlignmentTrace scratch(loc.hap->num_blocks());
      aligner.fw_haplotype_->go_to(req_allele[q]);  aligner.fw_haplotype_->fix();
      aligner.rev_haplotype_->go_to(req_allele[q]); aligner.rev_haplotype_->fix();
      aligner.process_read(aln, seed, &bq, false, &prob, scratch);
      aligner.fw_haplotype_->unfix(); aligner.rev_haplotype_->unfix();
      o->ll[q] = prob;
    }
    o->max_index[q] = -1;      /* a local of process_read; checked through hap_aln (the seed's 'M' splits the two sides) */
    if (!put_str(o->hap_aln, o->hap_aln_off, q, tr->hap_aln(), o->cap_chars)) return 3;
    if (tr->str_data_[1] != NULL){
      o->stutter_size[q] = tr->stutter_size(1);
      if (!put_str(o->str_seq, o->str_seq_off, q, tr->str_seq(1), o->cap_chars)) return 3;
    } else {
      o->stutter_size[q] = HIPSTR_NO_STR_DATA;
      o->str_seq_off[q+1] = o->str_seq_off[q];
    }
    if (!put_str(o->flank_seq, o->flank_seq_off, 2*q,   tr->flank_seq(0), o->cap_chars)) return 3;
    if (!put_str(o->flank_seq, o->flank_seq_off, 2*q+1, tr->flank_seq(2), o->cap_chars)) return 3;
    o->flank_ins[q] = tr->flank_ins_size(); o->flank_del[q] = tr->flank_del_size();
    int io = o->indel_off[q];
    for (size_t i = 0; i < tr->flank_indel_data().size(); i++, io++){
      if (io >= o->cap_chars) return 3;
      o->indel_pos[io] = tr->flank_indel_data()[i].first; o->indel_size[io] = tr->flank_indel_data()[i].second;
    }
    o->indel_off[q+1] = io;
    int so = o->snp_off[q];
    for (size_t i = 0; i < tr->flank_snp_data().size(); i++, so++){
      if (so >= o->cap_chars) return 3;
      o->snp_pos[so] = tr->flank_snp_data()[i].first; o->snp_base[so] = tr->flank_snp_data()[i].second;
    }
    o->snp_off[q+1] = so;
    Alignment& ta = tr->traced_aln();
    o->aln_start[q] = ta.get_start(); o->aln_stop[q] = ta.get_stop();
    int co = o->cigar_off[q];
    for (size_t i = 0; i < ta.get_cigar_list().size(); i++, co++){
      if (co >= o->cap_chars) return 3;
      o->cigar_op[co] = ta.get_cigar_list()[i].get_type(); o->cigar_len[co] = ta.get_cigar_list()[i].get_num();
    }
    o->cigar_off[q+1] = co;
    if (!put_str(o->aln_str, o->aln_str_off, q, ta.get_alignment(), o->cap_chars)) return 3;
    delete tr;
  }
  return 0;
}

/* Haplotype::get_aln_info() of every allele in visit order (NUL-separated), the input stitch_alignment_trace needs */
extern "C" int ref_hap_aln_info(const hipstr_batch_t* b, char* out, int out_cap, int32_t* offs){
  ensure_ready();
  if (b->n_loci != 1) return 1;
  int opt_cursor = 0;
  RefLocus loc;
  build_locus(b, 0, opt_cursor, loc);
  int pos = 0, k = 0;
  do {
    const std::string& s = loc.hap->get_aln_info();
    if (pos + (int)s.size() + 1 > out_cap) return 2;
    offs[k++] = pos;
    memcpy(out + pos, s.c_str(), s.size() + 1);
    pos += s.size() + 1;
  } while (loc.hap->next());
  offs[k] = pos;
  return 0;
}

/* ---- ReadPooler (read_pooler.h:13-53, read_pooler.cpp:3-20) and the pool -> read scatter + mate sums of
 * SeqStutterGenotyper::calc_hap_aln_probs (seq_stutter_genotyper.cpp:519-568).  The pooler and the aligner are the reference's
 * classes; the scatter / mate loops are a private member of SeqStutterGenotyper and are restated here line for line
 * (:531-543 and :551-564) — the whole member function is pinned through integration/genotype_flow.cpp as well. ---- */
extern "C" int ref_pool(const hipstr_batch_t* b, int32_t* pool_index, int32_t* n_pools, char* pool_quals, int32_t* pool_qual_off, int32_t cap){
  ensure_ready();
  if (b->n_loci != 1) return 1;
  BaseQuality bq;
  ReadPooler pooler;
  for (int r = 0; r < b->read_off[1]; r++){ Alignment a = make_alignment(b, r); pool_index[r] = pooler.add_alignment(a); }
  pooler.pool(bq);
  *n_pools = pooler.num_pools();
  std::vector<Alignment>& pooled = pooler.get_alignments();
  pool_qual_off[0] = 0;
  for (size_t i = 0; i < pooled.size(); i++)
    if (!put_str(pool_quals, pool_qual_off, (int)i, pooled[i].get_base_qualities(), cap)) return 3;
  return 0;
}

extern "C" int ref_pool_scatter(const hipstr_batch_t* b, const uint8_t* second_mate, const uint8_t* realign_pool, const uint8_t* copy_read,
                                double* log_aln_probs, int32_t* seed_positions){
  ensure_ready();
  if (b->n_loci != 1) return 1;
  BaseQuality bq;
  int opt_cursor = 0;
  RefLocus loc;
  build_locus(b, 0, opt_cursor, loc);
  const int num_alleles = loc.hap->num_combs(), num_reads = b->read_off[1];
  std::vector<bool> realign_to_haplotype(num_alleles, true);
  if (b->realign_hap) for (int k = 0; k < num_alleles; k++) realign_to_haplotype[k] = b->realign_hap[k] != 0;
  ReadPooler pooler;
  std::vector<int> pool_index(num_reads);
  for (int r = 0; r < num_reads; r++){ Alignment a = make_alignment(b, r); pool_index[r] = pooler.add_alignment(a); }
  pooler.pool(bq);
  std::vector<bool> realign(pooler.num_pools(), true);
  if (realign_pool) for (int i = 0; i < pooler.num_pools(); i++) realign[i] = realign_pool[i] != 0;
  HapAligner hap_aligner(loc.hap, realign_to_haplotype);                                  /* :522 */
  std::vector<Alignment>& pooled_alns = pooler.get_alignments();
  std::vector<double> log_pool_aln_probs(pooled_alns.size()*(size_t)num_alleles);
  std::vector<int> pool_seed_positions(pooled_alns.size());
  hap_aligner.process_reads(pooled_alns, 0, &bq, realign, log_pool_aln_probs.data(), pool_seed_positions.data());   /* :528 */
  double* log_aln_ptr = log_aln_probs;                                                    /* :531-543 */
  for (int i = 0; i < num_reads; i++){
    if (copy_read && !copy_read[i]){ log_aln_ptr += num_alleles; continue; }
    seed_positions[i] = pool_seed_positions[pool_index[i]];
    double* src_ptr = log_pool_aln_probs.data() + (size_t)num_alleles*pool_index[i];
    for (int j = 0; j < num_alleles; ++j, ++log_aln_ptr, ++src_ptr)
      if (realign_to_haplotype[j]) *log_aln_ptr = *src_ptr;
  }
  for (int i = 0; i < num_reads; ++i){                                                    /* :551-564 */
    if (!second_mate[i] || (copy_read && !copy_read[i])) continue;
    double* mate_one_ptr = log_aln_probs + (size_t)(i-1)*num_alleles;
    double* mate_two_ptr = log_aln_probs + (size_t)i*num_alleles;
    for (int j = 0; j < num_alleles; ++j, ++mate_one_ptr, ++mate_two_ptr)
      if (realign_to_haplotype[j]){ double total = *mate_one_ptr + *mate_two_ptr; *mate_one_ptr = total; *mate_two_ptr = total; }
  }
  return 0;
}
