// Exercises include/hipstr_hmm.hpp the way seq_stutter_genotyper.cpp uses the reference classes.
//   host_api_test            host-only parts (read pooling, median qualities, seed bases)
//   host_api_test --gpu      + HapAligner::process_reads, calc_hap_aln_probs, Genotyper posteriors on the device
// Prints "key value..." lines that tests/test_host_api.py compares with the golden vectors of SURVEY.md §8(c).
#include <cstdio>
#include <iostream>
#include <cstring>
#include "hipstr_hmm.hpp"
using namespace hipstr_amd;

class ProbeGenotyper : public Genotyper {   // the reference's derived genotypers allocate these once the allele count is known
 public:
  ProbeGenotyper(bool haploid, const std::vector<std::string>& names, const std::vector< std::vector<double> >& p1,
                 const std::vector< std::vector<double> >& p2, int num_alleles, const double* LL) : Genotyper(haploid, names, p1, p2){
    num_alleles_ = num_alleles;
    log_sample_posteriors_ = new double[(size_t)num_samples_*num_alleles_*num_alleles_];
    log_aln_probs_ = new double[(size_t)num_reads_*num_alleles_];
    memcpy(log_aln_probs_, LL, sizeof(double)*(size_t)num_reads_*num_alleles_);
  }
  void run(){
    double total = calc_log_sample_posteriors();
    std::vector< std::pair<int,int> > gts; get_optimal_haplotypes(gts);
    printf("post_total %.12f\n", total);
    for (size_t s = 0; s < gts.size(); s++) printf("post_gt %d %d\n", gts[s].first, gts[s].second);
    printf("post_first %.12f %.12f %.12f\n", log_sample_posteriors_[0], log_sample_posteriors_[1], log_sample_posteriors_[2]);
    // genotype calls with haplotypes {0,1,2} carrying variants {0,1,1}
    std::vector<int> h2a; h2a.push_back(0); h2a.push_back(1); h2a.push_back(1);
    std::vector< std::pair<int,int> > haps, bgts;
    std::vector<double> lp, lu, hlp, hlu, gld;
    std::vector< std::vector<double> > gls, pgls; std::vector< std::vector<int> > pls;
    extract_genotypes_and_likelihoods(2, h2a, haps, bgts, lp, lu, hlp, hlu, true, gls, gld, true, pls, true, pgls);
    for (size_t s = 0; s < bgts.size(); s++){
      printf("gt_call %zu %d %d %d %d %.12f %.12f %.12f %.12f %.12f", s, haps[s].first, haps[s].second, bgts[s].first, bgts[s].second, lp[s], lu[s], hlp[s], hlu[s], gld[s]);
      for (size_t i = 0; i < gls[s].size(); i++) printf(" %.12f %d", gls[s][i], pls[s][i]);
      for (size_t i = 0; i < pgls[s].size(); i++) printf(" %.12f", pgls[s][i]);
      printf("\n");
    }
  }
};

// --pool / --scatter <batch file> <side file>: ReadPooler and calc_hap_aln_probs of hipstr_hmm.hpp on a case stored by
// tests/golden/make_golden.py::pool_scatter (the batch in the library's flat format, masks and the pre-filled matrix as text);
// prints pool indices, pooled qualities and — with --scatter, on the device — the matrix and seeds after the call.
static int pool_scatter_mode(bool scatter, const char* batch_path, const char* side_path){
  hipstr_batch_file_t* bf = hipstr_batch_read(batch_path);
  if (!bf){ fprintf(stderr, "%s\n", hipstr_last_error()); return 1; }
  const hipstr_batch_t* b = hipstr_batch_file_batch(bf);
  StutterModel model(b->stutter[0], b->stutter[1], b->stutter[2], b->stutter[3], b->stutter[4], b->stutter[5], b->period[0]);
  std::vector<HapBlock*> blocks;
  int opt = 0;
  for (int k = 0; k < 3; k++){
    std::vector<std::string> seqs;
    for (int o = 0; o < b->blk_nopts[k]; o++, opt++) seqs.push_back(std::string(b->seq + b->opt_off[opt], b->opt_off[opt+1] - b->opt_off[opt]));
    HapBlock* hb = k == 1 ? new RepeatBlock(b->blk_start[k], b->blk_end[k], seqs[0], b->period[0], &model) : new HapBlock(b->blk_start[k], b->blk_end[k], seqs[0]);
    for (size_t o = 1; o < seqs.size(); o++) hb->add_alternate(seqs[o]);
    blocks.push_back(hb);
  }
  Haplotype hap(blocks);
  const int R = b->read_off[1], A = hap.num_combs();
  BaseQuality bq;
  ReadPooler pooler;
  std::vector<int> pool_index(R);
  for (int r = 0; r < R; r++){
    const int len = b->base_off[r+1] - b->base_off[r];
    Alignment a(b->read_start[r], 0, false, "R", std::string(b->quals + b->base_off[r], len), std::string(b->bases + b->base_off[r], len), "");
    for (int c = b->cigar_off[r]; c < b->cigar_off[r+1]; c++) a.add_cigar_element(CigarElement(b->cigar_op[c], b->cigar_len[c]));
    pool_index[r] = pooler.add_alignment(a);          // seq_stutter_genotyper.cpp:498
  }
  pooler.pool(bq);                                    // :633
  printf("n_pools %d\npool_index", pooler.num_pools());
  for (int r = 0; r < R; r++) printf(" %d", pool_index[r]);
  printf("\n");
  for (size_t i = 0; i < pooler.get_alignments().size(); i++) printf("pool_qual %s\n", pooler.get_alignments()[i].get_base_qualities().c_str());
  if (scatter){
    std::vector<bool> second_mate, realign_pool, copy_read, realign_hap(A, true);
    std::vector<double> ll;
    FILE* f = fopen(side_path, "r");
    if (!f){ perror(side_path); return 1; }
    char key[64]; int n;
    while (fscanf(f, "%63s %d", key, &n) == 2){
      for (int i = 0; i < n; i++){
        if (!strcmp(key, "prefill")){ unsigned long long u; if (fscanf(f, "%llx", &u) != 1) return 1; double v; memcpy(&v, &u, 8); ll.push_back(v); }
        else { int x; if (fscanf(f, "%d", &x) != 1) return 1; (!strcmp(key, "second_mate") ? second_mate : !strcmp(key, "realign_pool") ? realign_pool : copy_read).push_back(x != 0); }
      }
    }
    fclose(f);
    if (b->realign_hap) for (int k = 0; k < A; k++) realign_hap[k] = b->realign_hap[k] != 0;
    std::vector<int> seeds(R, -9);
    std::vector<char> mate_flags(second_mate.begin(), second_mate.end());
    calc_hap_aln_probs(&hap, pooler, bq, pool_index.data(), (const bool*)mate_flags.data(), (unsigned)R, realign_hap, realign_pool, copy_read, ll.data(), seeds.data());
    printf("seeds"); for (int r = 0; r < R; r++) printf(" %d", seeds[r]); printf("\n");
    printf("ll"); for (size_t i = 0; i < ll.size(); i++){ unsigned long long u; memcpy(&u, &ll[i], 8); printf(" %016llx", u); } printf("\n");
  }
  for (size_t k = 0; k < blocks.size(); k++) delete blocks[k];
  hipstr_batch_file_free(bf);
  return 0;
}

int main(int argc, char** argv){
  if (argc > 3 && (!strcmp(argv[1], "--pool") || !strcmp(argv[1], "--scatter"))) return pool_scatter_mode(!strcmp(argv[1], "--scatter"), argv[2], argv[3]);
  const bool gpu = argc > 1 && strcmp(argv[1], "--gpu") == 0;
  const std::string lf = "ACGTTGCATGCATGACCTGAGTCCATGACTTGACA", rf = "TTGACCGTAGGCTAGGCTTAACGGATCCGATTAGC";
  std::string gata10, gata11, gata12, gata9;
  for (int i = 0; i < 12; i++){ if (i < 9) gata9 += "GATA"; if (i < 10) gata10 += "GATA"; if (i < 11) gata11 += "GATA"; gata12 += "GATA"; }
  StutterModel model(0.9, 0.05, 0.05, 0.7, 0.005, 0.005, 4);
  HapBlock left(100, 135, lf), right(175, 210, rf);
  RepeatBlock str(135, 175, gata10, 4, &model);
  str.add_alternate(gata11); str.add_alternate(gata12); str.add_alternate(gata9);
  std::vector<HapBlock*> blocks; blocks.push_back(&left); blocks.push_back(&str); blocks.push_back(&right);
  Haplotype hap(blocks);
  printf("num_combs %d\n", hap.num_combs());

  // three reads: two identical sequences with different qualities (one pool), their mate, plus an unrelated read
  const std::string seq = lf.substr(10) + gata11 + rf.substr(0, 25);
  std::vector<Alignment> alns;
  const char* quals[3] = {"I", "5", "F"};
  for (int i = 0; i < 3; i++){
    alns.push_back(Alignment(110, 200, false, i < 2 ? "pairA" : "readB", std::string(seq.size(), quals[i][0]), seq, ""));
    alns.back().add_cigar_element(CigarElement('=', 65)); alns.back().add_cigar_element(CigarElement('I', 4)); alns.back().add_cigar_element(CigarElement('=', 25));
  }
  const std::string seq2 = lf.substr(5) + gata10 + rf.substr(0, 30);
  alns.push_back(Alignment(105, 205, false, "readC", std::string(seq2.size(), 'I'), seq2, ""));
  alns.back().add_cigar_element(CigarElement('=', (int)seq2.size()));

  BaseQuality bq;
  ReadPooler pooler;
  std::vector<int> pool_index(alns.size()); std::vector<char> second(alns.size(), 0);
  std::string prev = "";
  for (size_t i = 0; i < alns.size(); i++){
    pool_index[i] = pooler.add_alignment(alns[i]);
    second[i] = alns[i].get_name() == prev; prev = alns[i].get_name();     // seq_stutter_genotyper.cpp:497-503
  }
  pooler.pool(bq);
  printf("num_pools %d\n", pooler.num_pools());
  printf("pool0_qual %c\n", pooler.get_alignments()[0].get_base_qualities()[0]);   // upper median of {I,5,F} = F

  std::vector<bool> all_haps(hap.num_combs(), true);
  HapAligner aligner(&hap, all_haps);
  printf("seed %d %d\n", aligner.calc_seed_base(pooler.get_alignments()[0]), aligner.calc_seed_base(pooler.get_alignments()[1]));
  if (!gpu) return 0;

  // single pooled read with quality 'I' == the known-answer vector
  std::vector<Alignment> kat(1, alns[0]);
  double probs[4]; int seeds[1];
  aligner.process_reads(kat, 0, &bq, std::vector<bool>(1, true), probs, seeds);
  printf("kat_seed %d\nkat_ll %.11f %.11f %.11f %.11f\n", seeds[0], probs[0], probs[1], probs[2], probs[3]);

  // pool -> read scatter + mate-pair summation
  std::vector<bool> realign_pool(pooler.num_pools(), true), copy_read(alns.size(), true);
  std::vector<double> ll(alns.size()*4, -777.0); std::vector<int> sd(alns.size(), -5);
  bool sm[4]; for (int i = 0; i < 4; i++) sm[i] = second[i] != 0;
  calc_hap_aln_probs(&hap, pooler, bq, pool_index.data(), sm, (unsigned)alns.size(), all_haps, realign_pool, copy_read, ll.data(), sd.data());
  for (size_t i = 0; i < alns.size(); i++) printf("read_ll %zu %d %.11f %.11f %.11f %.11f\n", i, sd[i], ll[4*i], ll[4*i+1], ll[4*i+2], ll[4*i+3]);

  // Viterbi traceback of the known-answer read against every haplotype; argv[2..5] = Haplotype::get_aln_info() strings
  if (argc >= 6){
    hap.set_aln_info(std::vector<std::string>(argv + 2, argv + 6));
    std::vector<Alignment> t_alns(4, alns[0]); std::vector<int> t_haps;
    for (int k = 0; k < 4; k++) t_haps.push_back(k);
    std::vector<AlignmentTrace*> tr;
    aligner.trace_optimal_alns(t_alns, t_haps, &bq, tr);
    for (int k = 0; k < 4; k++){
      printf("trace %d %s %d %s %s %s %d %d %d %d %s %s\n", k, tr[k]->hap_aln().c_str(), tr[k]->has_str_data(1) ? tr[k]->stutter_size(1) : -100000,
             tr[k]->has_str_data(1) ? tr[k]->str_seq(1).c_str() : "-", tr[k]->flank_seq(0).c_str(), tr[k]->flank_seq(2).c_str(),
             tr[k]->flank_ins_size(), tr[k]->flank_del_size(), tr[k]->traced_aln().get_start(), tr[k]->traced_aln().get_stop(),
             tr[k]->traced_aln().getCigarString().c_str(), tr[k]->traced_aln().get_alignment().c_str());
      delete tr[k];
    }
    AlignmentTrace* one = aligner.trace_optimal_aln(alns[0], seeds[0], 1, &bq);
    printf("trace_one %s\n", one->hap_aln().c_str());
    delete one;
    // the seed base is the caller's (HapAligner.h:83, :93): a different split point, through process_read and trace_optimal_aln, against the
    // C-ABI's seeded entry points on the same flat batch
    {
      const int seed2 = seeds[0] - 7;
      double row[4] = {-1, -1, -1, -1}; AlignmentTrace best_trace(hap.num_blocks());
      hap.reset();
      aligner.process_read(alns[0], seed2, &bq, true, row, best_trace);
      printf("seeded_row"); for (int k = 0; k < 4; k++){ unsigned long long u; memcpy(&u, &row[k], 8); printf(" %016llx", u); } printf("\n");
      int best = 0; for (int k = 1; k < 4; k++) if (row[k] > row[best]) best = k;
      AlignmentTrace* tr2 = aligner.trace_optimal_aln(alns[0], seed2, best, &bq);
      printf("seeded_trace %d %s %s\n", best, tr2->hap_aln().c_str(), best_trace.hap_aln().c_str());
      delete tr2;
      hap.go_to(2); hap.fix();                       // a fixed haplotype: one entry, the others untouched
      double one_row[2] = {-5, -5}; AlignmentTrace dummy(hap.num_blocks());
      aligner.process_read(alns[0], seed2, &bq, false, one_row, dummy);
      hap.unfix(); hap.reset();
      unsigned long long u; memcpy(&u, &one_row[0], 8);
      printf("seeded_fixed %016llx %.1f\n", u, one_row[1]);
    }
    // the same haplotype set without handing the strings in: the library derives them (NW + adjust_indels)
    Haplotype hap2(blocks);
    HapAligner aligner2(&hap2, all_haps);
    AlignmentTrace* two = aligner2.trace_optimal_aln(alns[0], seeds[0], 2, &bq);
    printf("aln_info_derived");
    for (int k = 0; k < 4; k++) printf(" %s", hap2.get_aln_info(k).c_str());
    printf("\ntrace_two %d %d %s\n", two->traced_aln().get_start(), two->traced_aln().get_stop(), two->traced_aln().getCigarString().c_str());
    delete two;
  }

  // posteriors: SURVEY §8(c) second known-answer vector
  const double LL[15] = {-4.4,-7.3,-9.6, -7.1,-4.2,-7.5, -4.5,-7.0,-9.9, -9.0,-6.0,-4.1, -9.2,-6.3,-4.0};
  std::vector<std::string> names; names.push_back("s1"); names.push_back("s2");
  std::vector< std::vector<double> > p1(2), p2(2);
  p1[0] = std::vector<double>{0, -0.01, 0}; p2[0] = std::vector<double>{0, -5, 0}; p1[1] = std::vector<double>{0, 0}; p2[1] = std::vector<double>{0, 0};
  ProbeGenotyper g(false, names, p1, p2, 3, LL);
  g.run();

  // de novo stutter model from read lengths (EMStutterGenotyper): 4 samples, period 4
  {
    const int sizes[4][8] = {{0,0,4,4,0,-4,4,0}, {8,8,8,4,8,0,0,0}, {-4,-4,-8,-4,0,0,1,0}, {0,4,0,4,8,4,0,-4}};
    const int cnt[4] = {8, 6, 7, 8};
    std::vector< std::vector<int> > bps(4); std::vector< std::vector<double> > e1(4), e2(4); std::vector<std::string> nm;
    for (int s = 0; s < 4; s++){
      nm.push_back("s");
      for (int j = 0; j < cnt[s]; j++){ bps[s].push_back(sizes[s][j]); e1[s].push_back(j % 3 == 0 ? -0.02 : 0.0); e2[s].push_back(j % 3 == 0 ? -3.5 : 0.0); }
    }
    EMStutterGenotyper em(false, 4, bps, e1, e2, nm, 0);
    const bool ok = em.train(100, 0.01, 0.001, false, std::cerr);
    const double* q = em.get_stutter_model()->parameters();
    printf("em %d %d %.15f %.15f %.15f %.15f %.15f %.15f %.12f\n", ok ? 1 : 0, em.num_iterations(), q[0], q[1], q[2], q[3], q[4], q[5], em.final_LL());
  }
  return 0;
}
