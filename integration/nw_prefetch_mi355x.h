// integration/nw_prefetch_mi355x.h — Needleman-Wunsch results of a whole locus computed in ONE hipstr_nw_align call, served to
// NeedlemanWunsch::Align's unedited callers (round 6).
//
// integration/nw_align_mi355x.inc makes NeedlemanWunsch::Align (NeedlemanWunsch.cpp:380-417) a device call — one pair per call, a round
// trip each: realign() (AlignmentOps.cpp:25) runs it once per distinct read sequence of a locus from the read loop of
// GenotyperBamProcessor::left_align_reads (genotyper_bam_processor.cpp:51-95).  With this header the loop's pairs are handed over first
// (hipstr_mi355x_nw_prefetch: integration/left_align_reads_prepass_mi355x.inc puts that call in front of the loop), and Align's body
// looks a pair up before it goes to the device for it: realign() and its bookkeeping (AlignmentOps.cpp:27-100) stay as they are, the
// locus costs one device call.  A pair that was not handed over — or a prefetch that failed — takes the one-pair call as before: results
// never depend on the table.  The table is per host thread (the reference runs a locus on one thread) and is replaced by every prefetch.
#ifndef HIPSTR_MI355X_NW_PREFETCH_H_
#define HIPSTR_MI355X_NW_PREFETCH_H_
#include <cstdint>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>
#include "hipstr_hmm.h"

struct HipstrNwPrefetched {
  bool ok; float score;
  std::string ref_al, read_al;
  std::vector< std::pair<char, int32_t> > cigar;
};

inline std::unordered_map<std::string, HipstrNwPrefetched>& hipstr_mi355x_nw_table(){
  static thread_local std::unordered_map<std::string, HipstrNwPrefetched> t;
  return t;
}
inline std::string hipstr_mi355x_nw_key(const std::string& ref_seq, const std::string& read_seq, bool use_ref_end_penalty){
  std::string k; k.reserve(ref_seq.size() + read_seq.size() + 2);
  k += use_ref_end_penalty ? '1' : '0'; k += ref_seq; k += '\n'; k += read_seq;      // (sequences hold no newline)
  return k;
}
inline long long& hipstr_mi355x_nw_hits(){ static thread_local long long n = 0; return n; }      // Align calls of this thread served from the table (diagnostics)
inline const HipstrNwPrefetched* hipstr_mi355x_nw_find(const std::string& ref_seq, const std::string& read_seq, bool use_ref_end_penalty){
  const std::unordered_map<std::string, HipstrNwPrefetched>& t = hipstr_mi355x_nw_table();
  if (t.empty()) return NULL;
  std::unordered_map<std::string, HipstrNwPrefetched>::const_iterator it = t.find(hipstr_mi355x_nw_key(ref_seq, read_seq, use_ref_end_penalty));
  if (it == t.end()) return NULL;
  hipstr_mi355x_nw_hits()++;
  return &it->second;
}
// All (reference window, read) pairs of a locus in one device call; returns the number of distinct pairs now in the table (0 if the call
// failed: Align then computes pair by pair and reports the error itself).
inline int hipstr_mi355x_nw_prefetch(const std::vector<std::string>& ref_seqs, const std::vector<std::string>& read_seqs, bool use_ref_end_penalty){
  std::unordered_map<std::string, HipstrNwPrefetched>& t = hipstr_mi355x_nw_table();
  t.clear();
  std::vector<std::string> keys; std::vector<size_t> which;
  { std::unordered_map<std::string, int> seen;
    for (size_t i = 0; i < ref_seqs.size() && i < read_seqs.size(); i++){
      std::string k = hipstr_mi355x_nw_key(ref_seqs[i], read_seqs[i], use_ref_end_penalty);
      if (seen.insert(std::make_pair(k, 1)).second){ keys.push_back(k); which.push_back(i); }
    } }
  const int32_t n = (int32_t)which.size();
  if (n == 0) return 0;
  std::vector<int32_t> ref_off(n + 1, 0), read_off(n + 1, 0);
  std::string refs, reads; int64_t cap = 0;
  for (int32_t i = 0; i < n; i++){
    refs += ref_seqs[which[i]]; reads += read_seqs[which[i]];
    ref_off[i+1] = (int32_t)refs.size(); read_off[i+1] = (int32_t)reads.size();
    cap += (int64_t)ref_seqs[which[i]].size() + (int64_t)read_seqs[which[i]].size() + 2;
  }
  hipstr_nw_batch_t nb;
  nb.n_pairs = n; nb.ref_off = ref_off.data(); nb.ref_seqs = refs.data(); nb.read_off = read_off.data(); nb.read_seqs = reads.data();
  nb.use_ref_end_penalty = use_ref_end_penalty ? 1 : 0;
  std::vector<float> score(n); std::vector<uint8_t> ok(n);
  std::vector<int64_t> aln_off(n + 1), cigar_off(n + 1);
  std::vector<char> ref_al(cap), read_al(cap), cigar_op(cap);
  std::vector<int32_t> cigar_len(cap);
  hipstr_nw_out_t out;
  out.score = score.data(); out.ok = ok.data(); out.aln_off = aln_off.data(); out.ref_al = ref_al.data(); out.read_al = read_al.data();
  out.cigar_off = cigar_off.data(); out.cigar_op = cigar_op.data(); out.cigar_len = cigar_len.data(); out.cap_aln = cap; out.cap_cigar = cap;
  if (hipstr_nw_align(&nb, &out) != 0) return 0;
  for (int32_t i = 0; i < n; i++){
    HipstrNwPrefetched r;
    r.ok = ok[i] != 0; r.score = score[i];
    r.ref_al.assign(ref_al.data() + aln_off[i], (size_t)(aln_off[i+1] - aln_off[i]));
    r.read_al.assign(read_al.data() + aln_off[i], (size_t)(aln_off[i+1] - aln_off[i]));
    for (int64_t c = cigar_off[i]; c < cigar_off[i+1]; c++) r.cigar.push_back(std::make_pair(cigar_op[c], cigar_len[c]));
    t[keys[i]] = r;
  }
  return n;
}
#endif
