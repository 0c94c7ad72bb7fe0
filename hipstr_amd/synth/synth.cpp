/*
 * synth.cpp — seeded synthetic STR-locus generator (bench + test support).
 *
 * Produces a hipstr_batch_t whose loci look like what HipSTR's
 * SeqStutterGenotyper hands to HapAligner: a [left flank, STR, right flank]
 * haplotype (HaplotypeGenerator.cpp:339-366) with copy-number alleles of a random
 * motif, plus pooled reads sampled from those alleles with substitutions, PCR
 * stutter, rare flank indels, haplotype overhang, and a CIGAR against the
 * reference allele (so HapAligner::calc_seed_base has real work to do).
 *
 * The generator follows the recipe of SURVEY.md §8(d); it is our own code, has no
 * counterpart in the reference (which ships no benchmark), and depends on nothing
 * but this file: a splitmix64 stream per locus makes every byte reproducible
 * across machines.  Identical bytes go to the GPU path, the C oracle and the
 * compiled reference.
 */
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/hipstr_hmm.h"

namespace {

struct Rng {
  uint64_t s;
  explicit Rng(uint64_t seed) : s(seed) {}
  uint64_t next(){
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  double uni(){ return (next() >> 11) * (1.0/9007199254740992.0); }
  int below(int n){ return (int)(next() % (uint64_t)n); }      // n > 0
  int range(int lo, int hi){ return hi < lo ? lo : lo + below(hi-lo+1); }       // inclusive (an empty range: lo, no draw)
  char base(){ return "ACGT"[next() & 3]; }
  char other_base(char c){ char b; do { b = base(); } while (b == c); return b; }
};

struct Cfg {
  int32_t n_loci, reads_per_locus, n_str_alleles, read_len, flank_len, str_bp, n_flank_opts;
  uint64_t seed;
  double sub_rate, stutter_rate, indel_rate, imperfect_rate, mask_rate;
  int force_period; // HIPSTR_SYNTH_PERIOD=p (1..9): every locus gets period p instead of a draw from {2..6}
  int inherit;      // HIPSTR_SYNTH_INHERIT=k: the REFERENCE allele carries k (1..3) interrupted repeat units, inherited by every candidate allele
};

struct Synth {
  hipstr_batch_t b;
  std::vector<int32_t> blk_start, blk_end, blk_nopts, period, opt_off, hap_off, read_off, base_off, read_start, cigar_off, cigar_len;
  std::vector<double> stutter;
  std::vector<uint8_t> realign_hap, realign_read;
  std::vector<int32_t> src_allele;   // STR option each read was drawn from (for posterior benches)
  std::string seq, bases, quals, cigar_op;
};

bool has_sub_period(const std::string& m){
  int p = m.size();
  for (int q = 1; q < p; q++){
    if (p % q) continue;
    bool ok = true;
    for (int i = q; i < p && ok; i++) ok = (m[i] == m[i-q]);
    if (ok) return true;
  }
  return false;
}

std::string repeat(const std::string& m, int copies){
  std::string s; s.reserve(m.size()*copies);
  for (int i = 0; i < copies; i++) s += m;
  return s;
}

bool by_len_then_seq(const std::string& a, const std::string& b){   // stringops.cpp:35-39 ordering
  if (a.size() != b.size()) return a.size() < b.size();
  return a < b;
}

void push_cigar(std::string& ops, std::vector<int32_t>& lens, size_t first, char op, int n){
  if (n <= 0) return;
  if (ops.size() > first && ops.back() == op) lens.back() += n;
  else { ops.push_back(op); lens.push_back(n); }
}

void gen_locus(const Cfg& c, int l, Synth& out){
  Rng rng(c.seed * 0x100000001B3ull + 0x51ED270B1ull * (uint64_t)(l+1));
  // --- motif / alleles ---
  static const int periods[5] = {2,3,4,5,6};
  static const double pw[5]   = {.35,.20,.30,.10,.05};
  double u = rng.uni(); int p = 6;
  for (int i = 0; i < 5; i++){ if (u < pw[i]){ p = periods[i]; break; } u -= pw[i]; }
  if (c.force_period > 0) p = c.force_period;           // HIPSTR_SYNTH_PERIOD: periods the weights never draw (1, 7..9: stutter_model.h:38); the draw above still happens
  std::string motif;
  do { motif.clear(); for (int i = 0; i < p; i++) motif += rng.base(); } while (has_sub_period(motif));
  int c0 = std::max(2, (int)std::lround((double)c.str_bp / p));
  std::vector<int> copies;
  for (int d = 1; (int)copies.size() < c.n_str_alleles-1 && d < 10000; d++){
    if (c0-d >= 2) copies.push_back(c0-d);
    if ((int)copies.size() < c.n_str_alleles-1) copies.push_back(c0+d);
  }
  std::string ref_str = repeat(motif, c0);
  // HIPSTR_SYNTH_INHERIT=k: what real panels look like more often than not — the reference allele itself is an interrupted repeat, and the
  // candidates differ from it in the number of units of its LAST pure tract: k of the first units (not the very first) carry a
  // substitution, in the reference allele and in every candidate long enough to contain them.  (No generator draws when it is off.)
  std::vector< std::pair<int,char> > inherited;        // (position from the block's left end, base)
  if (c.inherit > 0){
    const int zone = std::max(2, c0/2);                 // units 1 .. zone-1
    std::vector<int> units;
    for (int t = 0; t < c.inherit && (int)units.size() < zone - 1; t++){
      int un; do { un = rng.range(1, zone - 1); } while (std::find(units.begin(), units.end(), un) != units.end());
      units.push_back(un);
      const int pos = un*p + rng.below(p);
      inherited.push_back(std::make_pair(pos, rng.other_base(motif[pos % p])));
    }
    for (size_t t = 0; t < inherited.size(); t++) ref_str[inherited[t].first] = inherited[t].second;
  }
  std::vector<std::string> alts;
  for (size_t i = 0; i < copies.size(); i++){
    std::string a = repeat(motif, copies[i]);
    for (size_t t = 0; t < inherited.size(); t++) if (inherited[t].first + p <= (int)a.size()) a[inherited[t].first] = inherited[t].second;
    if (rng.uni() < c.imperfect_rate && a.size() > 2){
      int pos = rng.range(1, (int)a.size()-2);
      a[pos] = rng.other_base(a[pos]);
    }
    alts.push_back(a);
  }
  std::sort(alts.begin(), alts.end(), by_len_then_seq);
  alts.erase(std::unique(alts.begin(), alts.end()), alts.end());
  std::vector<std::string> str_opts; str_opts.push_back(ref_str);
  for (size_t i = 0; i < alts.size(); i++) if (alts[i] != ref_str) str_opts.push_back(alts[i]);

  // --- flanks (option 0 = reference; extra options = a substitution or a 1-bp deletion) ---
  std::vector<std::string> fl[2];
  for (int side = 0; side < 2; side++){
    std::string f; for (int i = 0; i < c.flank_len; i++) f += rng.base();
    fl[side].push_back(f);
    for (int o = 1; o < c.n_flank_opts && c.flank_len >= 6; o++){       // (a flank of fewer than six bases has no alternatives: the edit keeps two bases from either end)
      std::string g = f;
      int pos = rng.range(2, (int)g.size()-3);
      if (o & 1) g[pos] = rng.other_base(g[pos]); else g.erase(pos, 1);
      if (std::find(fl[side].begin(), fl[side].end(), g) == fl[side].end()) fl[side].push_back(g);
    }
  }
  const int32_t start0 = 1000 + 37*(l % 11);
  const int32_t bstart[3] = { start0, start0 + c.flank_len, start0 + c.flank_len + (int32_t)ref_str.size() };
  const int32_t bend[3]   = { bstart[1], bstart[2], bstart[2] + c.flank_len };
  const std::vector<std::string>* opts[3] = { &fl[0], &str_opts, &fl[1] };
  int ncombs = 1;
  for (int k = 0; k < 3; k++){
    out.blk_start.push_back(bstart[k]); out.blk_end.push_back(bend[k]);
    out.blk_nopts.push_back((int32_t)opts[k]->size());
    ncombs *= (int)opts[k]->size();
    for (size_t o = 0; o < opts[k]->size(); o++){
      out.seq += (*opts[k])[o];
      out.opt_off.push_back((int32_t)out.seq.size());
    }
  }
  out.period.push_back(p);
  const double sm[6] = {0.9, 0.05, 0.05, 0.7, 0.005, 0.005};
  out.stutter.insert(out.stutter.end(), sm, sm+6);
  out.hap_off.push_back(out.hap_off.back() + ncombs);
  for (int k = 0; k < ncombs; k++) out.realign_hap.push_back(c.mask_rate > 0 && rng.uni() < c.mask_rate ? 0 : 1);

  // --- reads ---
  const std::string ref_hap = fl[0][0] + ref_str + fl[1][0];
  const int Lr = c.read_len;
  int het[2] = { rng.below((int)str_opts.size()), rng.below((int)str_opts.size()) };
  for (int r = 0; r < c.reads_per_locus; r++){
    if ((r % 16) == 0){ het[0] = rng.below((int)str_opts.size()); het[1] = rng.below((int)str_opts.size()); }
    int sopt = het[rng.below(2)];
    std::string str = str_opts[sopt];
    double su = rng.uni();
    if (su < c.stutter_rate) str += str.substr(str.size()-p);                       // +1 repeat
    else if (su < 2*c.stutter_rate && (int)str.size() >= 2*p) str.erase(str.size()-p);  // -1 repeat
    const std::string& lf = fl[0][rng.below((int)fl[0].size())];
    const std::string& rf = fl[1][rng.below((int)fl[1].size())];
    const int Ls = lf.size(), Bs = str.size(), Br = ref_str.size();
    const int Hs = Ls + Bs + rf.size();
    // source coordinate -> reference coordinate (or -1 for inserted bases)
    std::vector<int32_t> coord(Hs);
    for (int i = 0; i < Ls; i++) coord[i] = bstart[1] - Ls + i;      // right-justify alt flanks against the STR
    // STR length difference shows up as one indel at the start of the repeat (left-aligned, as the
    // reference's NW left-alignment would place it): extra bases are 'I', missing ones a coordinate jump.
    for (int i = 0; i < Bs; i++) coord[Ls+i] = (i < Bs-Br) ? -1 : bstart[1] + (Br-Bs) + i;
    for (size_t i = 0; i < rf.size(); i++) coord[Ls+Bs+i] = bstart[2] + i;
    const std::string src = lf + str + rf;
    int lo = std::max(-20, Ls - Lr + 1), hi = std::min(Ls + Bs - 1, Hs + 20 - Lr);
    if (hi < lo) hi = lo;
    int s = rng.range(lo, hi);
    int indel_at = -1, indel_kind = 0;
    if (rng.uni() < c.indel_rate && Lr >= 8){ indel_at = rng.range(3, Lr-4); indel_kind = rng.below(2) ? 1 : -1; }

    const size_t cig_first = out.cigar_op.size();
    std::string rb; rb.reserve(Lr);
    int32_t rstart = INT32_MIN, last_coord = INT32_MIN;
    int sp = s;
    while ((int)rb.size() < Lr){
      if ((int)rb.size() == indel_at && indel_kind == 1 && sp >= 0 && sp < Hs && coord[sp] >= 0 && (sp < Ls || sp >= Ls+Bs)){
        rb += rng.base(); push_cigar(out.cigar_op, out.cigar_len, cig_first, 'I', 1); indel_at = -1; continue;
      }
      if ((int)rb.size() == indel_at && indel_kind == -1 && sp >= 1 && sp < Hs-1 && (sp < Ls-1 || sp >= Ls+Bs+1)){
        sp++; indel_at = -1; continue;   // skip one source base: shows up as a coordinate jump -> 'D'
      }
      char bch; int32_t cd; bool in_window = (sp >= 0 && sp < Hs);
      if (in_window){ bch = src[sp]; cd = coord[sp]; }
      else { bch = rng.base(); cd = (sp < 0) ? coord[0] + sp : coord[Hs-1] + (sp - (Hs-1)); }
      if (rng.uni() < c.sub_rate) bch = rng.other_base(bch);
      if (cd < 0) push_cigar(out.cigar_op, out.cigar_len, cig_first, 'I', 1);
      else {
        if (rstart == INT32_MIN) rstart = cd;
        if (last_coord != INT32_MIN && cd > last_coord+1) push_cigar(out.cigar_op, out.cigar_len, cig_first, 'D', cd-last_coord-1);
        bool eq = true;
        int ri = cd - bstart[0];
        if (ri >= 0 && ri < (int)ref_hap.size()) eq = (ref_hap[ri] == bch);
        push_cigar(out.cigar_op, out.cigar_len, cig_first, eq ? '=' : 'X', 1);
        last_coord = cd;
      }
      rb += bch; sp++;
    }
    if (rstart == INT32_MIN) rstart = bstart[1];
    std::string q(Lr, 'F');
    for (int i = 0; i < Lr; i++){
      double qu = rng.uni();
      q[i] = qu < 0.02 ? '#' : qu < 0.10 ? ',' : qu < 0.30 ? ':' : 'F';
    }
    out.bases += rb; out.quals += q;
    out.base_off.push_back((int32_t)out.bases.size());
    out.read_start.push_back(rstart);
    out.cigar_off.push_back((int32_t)out.cigar_op.size());
    out.realign_read.push_back(c.mask_rate > 0 && rng.uni() < c.mask_rate ? 0 : 1);
    out.src_allele.push_back(sopt);
  }
  out.read_off.push_back(out.read_off.back() + c.reads_per_locus);
}

} // namespace

extern "C" {

// Loci [first_locus, first_locus + n_loci) of the seeded set: a locus depends on (seed, its index) only, so any slice of a
// large set can be regenerated alone (parity samples of a full-size batch, shards of one set across ranks).
void* synth_create_at(int32_t first_locus, int32_t n_loci, int32_t reads_per_locus, int32_t n_str_alleles, int32_t read_len, int32_t flank_len,
                      int32_t str_bp, int32_t n_flank_opts, uint64_t seed, double mask_rate){
  Cfg c; c.n_loci = n_loci; c.reads_per_locus = reads_per_locus; c.n_str_alleles = n_str_alleles; c.read_len = read_len;
  c.flank_len = flank_len; c.str_bp = str_bp; c.n_flank_opts = std::max(1, n_flank_opts); c.seed = seed;
  c.sub_rate = 0.005; c.stutter_rate = 0.05; c.indel_rate = 0.01; c.imperfect_rate = 0.05; c.mask_rate = mask_rate;
  if (const char* e = getenv("HIPSTR_SYNTH_IMPERFECT")) c.imperfect_rate = atof(e);      // experiments: share of alleles with an interrupted repeat
  c.inherit = 0; c.force_period = 0;
  if (const char* e = getenv("HIPSTR_SYNTH_PERIOD")) c.force_period = std::max(0, std::min(9, atoi(e)));
  if (const char* e = getenv("HIPSTR_SYNTH_INHERIT")) c.inherit = std::max(0, std::min(3, atoi(e)));
  Synth* s = new Synth();
  s->opt_off.push_back(0); s->hap_off.push_back(0); s->read_off.push_back(0); s->base_off.push_back(0); s->cigar_off.push_back(0);
  for (int l = 0; l < n_loci; l++) gen_locus(c, first_locus + l, *s);
  hipstr_batch_t& b = s->b;
  b.n_loci = n_loci;
  b.blk_start = s->blk_start.data(); b.blk_end = s->blk_end.data(); b.blk_nopts = s->blk_nopts.data();
  b.period = s->period.data(); b.stutter = s->stutter.data(); b.opt_off = s->opt_off.data(); b.seq = s->seq.data();
  b.hap_off = s->hap_off.data(); b.realign_hap = mask_rate > 0 ? s->realign_hap.data() : NULL;
  b.read_off = s->read_off.data(); b.base_off = s->base_off.data(); b.bases = s->bases.data(); b.quals = s->quals.data();
  b.read_start = s->read_start.data(); b.cigar_off = s->cigar_off.data(); b.cigar_op = s->cigar_op.data();
  b.cigar_len = s->cigar_len.data(); b.realign_read = mask_rate > 0 ? s->realign_read.data() : NULL;
  return s;
}

void* synth_create(int32_t n_loci, int32_t reads_per_locus, int32_t n_str_alleles, int32_t read_len, int32_t flank_len,
                   int32_t str_bp, int32_t n_flank_opts, uint64_t seed, double mask_rate){
  return synth_create_at(0, n_loci, reads_per_locus, n_str_alleles, read_len, flank_len, str_bp, n_flank_opts, seed, mask_rate);
}

const hipstr_batch_t* synth_batch(void* h){ return &((Synth*)h)->b; }
const int32_t* synth_src_allele(void* h){ return ((Synth*)h)->src_allele.data(); }
void synth_free(void* h){ delete (Synth*)h; }

} // extern "C"
