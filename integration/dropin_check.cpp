// dropin_check.cpp — builds reference objects for seeded synthetic loci and runs the reference's
// HapAligner::process_reads and HapAlignerMI355X::process_reads side by side on the same Haplotype*/Alignment
// objects; exits 0 iff seeds are identical and every log-likelihood agrees bit for bit.  Needs a GPU to run.
// Built by `make -C oracle dropin` into oracle/_ref/dropin_check (travels to the GPU box prebuilt).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "SeqAlignment/AlignmentModel.h"
#include "SeqAlignment/HapAligner.h"
#include "SeqAlignment/HapAlignerMI355X.h"
#include "SeqAlignment/RepeatBlock.h"
#include "mathops.h"
#include "stutter_model.h"

#include "hipstr_hmm.h"

extern "C" {
void* synth_create(int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, int32_t, uint64_t, double);
const hipstr_batch_t* synth_batch(void*);
void synth_free(void*);
}

int main(int argc, char** argv){
  const int n_loci = argc > 1 ? atoi(argv[1]) : 6, reads = argc > 2 ? atoi(argv[2]) : 24, alleles = argc > 3 ? atoi(argv[3]) : 8;
  const int flank_opts = argc > 4 ? atoi(argv[4]) : 2;
  precompute_integer_logs();
  init_alignment_model();
  void* h = synth_create(n_loci, reads, alleles, 150, 60, 40, flank_opts, 4242, 0.2);
  const hipstr_batch_t* b = synth_batch(h);
  BaseQuality bq;
  int opt = 0; long n_cmp = 0, n_bad = 0, n_trace = 0; double max_diff = 0;
  for (int l = 0; l < b->n_loci; l++){
    const double* sp = b->stutter + 6*l;
    StutterModel model(sp[0], sp[1], sp[2], sp[3], sp[4], sp[5], b->period[l]);
    std::vector<HapBlock*> blocks;
    for (int k = 0; k < 3; k++){
      std::vector<std::string> seqs;
      for (int o = 0; o < b->blk_nopts[3*l+k]; o++, opt++) seqs.push_back(std::string(b->seq + b->opt_off[opt], b->opt_off[opt+1]-b->opt_off[opt]));
      HapBlock* hb = (k == 1) ? new RepeatBlock(b->blk_start[3*l+k], b->blk_end[3*l+k], seqs[0], b->period[l], &model)
                              : new HapBlock(b->blk_start[3*l+k], b->blk_end[3*l+k], seqs[0]);
      for (size_t o = 1; o < seqs.size(); o++) hb->add_alternate(seqs[o]);
      blocks.push_back(hb);
    }
    Haplotype hap(blocks);
    const int A = hap.num_combs();
    std::vector<bool> realign_hap(A), realign_read;
    for (int k = 0; k < A; k++) realign_hap[k] = b->realign_hap[b->hap_off[l]+k] != 0;
    std::vector<Alignment> alns;
    for (int r = b->read_off[l]; r < b->read_off[l+1]; r++){
      const int len = b->base_off[r+1]-b->base_off[r];
      Alignment a(b->read_start[r], 0, false, "R", std::string(b->quals + b->base_off[r], len), std::string(b->bases + b->base_off[r], len), "");
      for (int c = b->cigar_off[r]; c < b->cigar_off[r+1]; c++) a.add_cigar_element(CigarElement(b->cigar_op[c], b->cigar_len[c]));
      alns.push_back(a);
      realign_read.push_back(b->realign_read[r] != 0);
    }
    const size_t n = alns.size()*(size_t)A;
    std::vector<double> want(n, -4.5), got(n, -4.5);
    std::vector<int> wseed(alns.size(), -9), gseed(alns.size(), -9);
    HapAligner cpu(&hap, realign_hap);
    cpu.process_reads(alns, 0, &bq, realign_read, want.data(), wseed.data());
    HapAlignerMI355X gpu(&hap, realign_hap);
    gpu.process_reads(alns, 0, &bq, realign_read, got.data(), gseed.data());
    for (size_t i = 0; i < alns.size(); i++) if (wseed[i] != gseed[i]){ n_bad++; }
    for (size_t i = 0; i < n; i++){
      n_cmp++;
      if (memcmp(&want[i], &got[i], 8) != 0){ n_bad++; if (fabs(want[i]-got[i]) > max_diff) max_diff = fabs(want[i]-got[i]); }
    }
    if (!alns.empty() && wseed[0] >= 0 && gpu.calc_seed_base(alns[0]) != cpu.calc_seed_base(alns[0])) n_bad++;
    // process_read (HapAligner.h:83) with seeds of the CALLER's choosing: calc_seed_base's value and values off it
    for (size_t i = 0; i < alns.size() && i < 6; i++){
      if (wseed[i] < 0) continue;
      const int len = (int)alns[i].get_sequence().size();
      const int shifts[3] = { 0, -7, 11 };
      for (int sh = 0; sh < 3; sh++){
        int seed = wseed[i] + shifts[sh];
        if (seed < 1) seed = 1;
        if (seed > len - 2) seed = len - 2;
        std::vector<double> wrow(A, -4.5), grow(A, -4.5);
        AlignmentTrace wt(hap.num_blocks()), gt(hap.num_blocks());
        cpu.process_read(alns[i], seed, &bq, false, wrow.data(), wt);
        gpu.process_read(alns[i], seed, &bq, false, grow.data(), gt);
        for (int k = 0; k < A; k++){ n_cmp++; if (memcmp(&wrow[k], &grow[k], 8) != 0){ n_bad++; if (n_bad <= 3) fprintf(stderr, "process_read mismatch locus %d read %zu seed %d hap %d: %.17g vs %.17g\n", l, i, seed, k, wrow[k], grow[k]); } }
        // and the traceback from that seed (trace_optimal_aln's seed_base argument, HapAligner.h:93)
        const int best = (int)((i + sh) % A);
        AlignmentTrace* w = cpu.trace_optimal_aln(alns[i], seed, best, &bq);
        AlignmentTrace* g = gpu.trace_optimal_aln(alns[i], seed, best, &bq);
        n_trace++;
        if (w->hap_aln() != g->hap_aln() || w->traced_aln().getCigarString() != g->traced_aln().getCigarString() || w->traced_aln().get_start() != g->traced_aln().get_start()
            || w->flank_seq(0) != g->flank_seq(0) || w->flank_seq(2) != g->flank_seq(2) || w->has_stutter() != g->has_stutter()){
          n_bad++;
          if (n_bad <= 3) fprintf(stderr, "seeded trace mismatch locus %d read %zu seed %d hap %d\n  want %s\n  got  %s\n", l, i, seed, best, w->hap_aln().c_str(), g->hap_aln().c_str());
        }
        delete w; delete g;
      }
    }
    // Viterbi traceback: the reference's trace_optimal_aln per read vs one batched call on the MI355X
    {
      std::vector<Alignment> t_alns; std::vector<int> t_haps;
      for (size_t i = 0; i < alns.size(); i++)
        if (wseed[i] >= 0){ t_alns.push_back(alns[i]); t_haps.push_back((int)((i*7 + l) % A)); }
      std::vector<AlignmentTrace*> got_tr;
      gpu.trace_optimal_alns(t_alns, t_haps, &bq, got_tr);
      for (size_t i = 0; i < t_alns.size(); i++){
        AlignmentTrace* w = cpu.trace_optimal_aln(t_alns[i], cpu.calc_seed_base(t_alns[i]), t_haps[i], &bq);
        AlignmentTrace* g = got_tr[i];
        bool same = w->hap_aln() == g->hap_aln() && w->flank_ins_size() == g->flank_ins_size() && w->flank_del_size() == g->flank_del_size()
          && w->flank_seq(0) == g->flank_seq(0) && w->flank_seq(2) == g->flank_seq(2)
          && w->flank_indel_data() == g->flank_indel_data() && w->flank_snp_data() == g->flank_snp_data()
          && w->has_stutter() == g->has_stutter()
          && w->traced_aln().get_start() == g->traced_aln().get_start() && w->traced_aln().get_stop() == g->traced_aln().get_stop()
          && w->traced_aln().getCigarString() == g->traced_aln().getCigarString()
          && w->traced_aln().get_alignment() == g->traced_aln().get_alignment()
          && w->traced_aln().get_sequence() == g->traced_aln().get_sequence();
        if (same && w->has_stutter()) same = w->stutter_size(1) == g->stutter_size(1) && w->str_seq(1) == g->str_seq(1);
        n_trace++;
        if (!same){
          n_bad++;
          if (n_bad <= 3)
            fprintf(stderr, "trace mismatch locus %d read %zu hap %d\n  hap_aln want %s\n          got  %s\n  cigar want %s got %s start %d/%d stop %d/%d\n", l, i, t_haps[i],
                    w->hap_aln().c_str(), g->hap_aln().c_str(), w->traced_aln().getCigarString().c_str(), g->traced_aln().getCigarString().c_str(),
                    w->traced_aln().get_start(), g->traced_aln().get_start(), w->traced_aln().get_stop(), g->traced_aln().get_stop());
        }
        delete w; delete g;
      }
    }
    for (size_t k = 0; k < blocks.size(); k++) delete blocks[k];
  }
  synth_free(h);
  printf("dropin_check: %ld log-likelihoods and %ld tracebacks compared, %ld mismatches, max|diff| %g\n", n_cmp, n_trace, n_bad, max_diff);
  return n_bad == 0 ? 0 : 1;
}
