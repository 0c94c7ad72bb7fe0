// gather.cpp — the host-side ordered gather of SURVEY §8(e): k-way merge of per-worker record streams by (chromosome index, position).
// Every worker (a GPU's process, or a hipstr_stream_t) emits the records of ITS loci in order; the VCF writer of the reference
// tolerates out-of-order positions only within MAX_RECORD_PAD = 50 bp (vcf_writer.h:53, vcf_writer.cpp:7-36), so shards cut from a
// sorted region list must be merged back into one globally ordered stream per chromosome.  A record can be released as soon as it is
// the smallest pending one and every stream that has not ended has a record pending (nothing smaller can still arrive).
// Host only; no device is touched.
#include <cstdint>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/hipstr_hmm.h"
#include "api_internal.h"

namespace {
struct Rec { int32_t chrom, pos; std::string bytes; };
}

struct hipstr_gather {
  std::mutex m;
  std::vector< std::deque<Rec> > q;
  std::vector<uint8_t> ended;
  std::vector<int32_t> last_chrom, last_pos;       // per stream: the order check
  int64_t pushed = 0, popped = 0;
};

extern "C" {

hipstr_gather_t* hipstr_gather_open(int32_t n_streams){
  if (n_streams < 1){ hipstr::api_fail("a gather needs at least one stream"); return NULL; }
  hipstr_gather* g = new hipstr_gather();
  g->q.resize(n_streams); g->ended.assign(n_streams, 0); g->last_chrom.assign(n_streams, INT32_MIN); g->last_pos.assign(n_streams, INT32_MIN);
  return g;
}

int hipstr_gather_push(hipstr_gather_t* g, int32_t stream, int32_t chrom_index, int32_t pos, const void* record, int64_t bytes){
  if (!g || stream < 0 || stream >= (int32_t)g->q.size() || bytes < 0 || (bytes > 0 && !record)) return hipstr::api_fail("bad argument");
  std::lock_guard<std::mutex> lock(g->m);
  if (g->ended[stream]) return hipstr::api_fail("stream has ended");
  if (chrom_index < g->last_chrom[stream] || (chrom_index == g->last_chrom[stream] && pos < g->last_pos[stream]))
    return hipstr::api_fail("records of a stream must arrive in (chromosome, position) order");
  g->last_chrom[stream] = chrom_index; g->last_pos[stream] = pos;
  g->q[stream].push_back(Rec{chrom_index, pos, std::string((const char*)record, (size_t)bytes)});
  g->pushed++;
  return 0;
}

int hipstr_gather_end(hipstr_gather_t* g, int32_t stream){
  if (!g || stream < 0 || stream >= (int32_t)g->q.size()) return hipstr::api_fail("bad argument");
  std::lock_guard<std::mutex> lock(g->m);
  g->ended[stream] = 1;
  return 0;
}

// 0: a record was released; 2: nothing can be released yet (a stream that has not ended has nothing pending); 3: every stream has
// ended and everything was released; 1: error (buffer too small: *bytes says how much is needed, the record stays queued)
int hipstr_gather_pop(hipstr_gather_t* g, int32_t* stream, int32_t* chrom_index, int32_t* pos, void* out, int64_t cap, int64_t* bytes){
  if (!g) return hipstr::api_fail("bad argument");
  std::lock_guard<std::mutex> lock(g->m);
  int best = -1; bool all_done = true;
  for (size_t s = 0; s < g->q.size(); s++){
    if (g->q[s].empty()){ if (!g->ended[s]) return 2; continue; }
    all_done = false;
    const Rec& r = g->q[s].front();
    // ties go to the lower stream index: with contiguous shards that is the earlier shard
    if (best < 0 || r.chrom < g->q[best].front().chrom || (r.chrom == g->q[best].front().chrom && r.pos < g->q[best].front().pos)) best = (int)s;
  }
  if (best < 0) return all_done ? 3 : 2;
  Rec& r = g->q[best].front();
  if (bytes) *bytes = (int64_t)r.bytes.size();
  if ((int64_t)r.bytes.size() > cap) return hipstr::api_fail("record buffer too small");
  if (stream) *stream = best;
  if (chrom_index) *chrom_index = r.chrom;
  if (pos) *pos = r.pos;
  if (!r.bytes.empty()) memcpy(out, r.bytes.data(), r.bytes.size());
  g->q[best].pop_front();
  g->popped++;
  return 0;
}

void hipstr_gather_close(hipstr_gather_t* g){ delete g; }

}  // extern "C"
