// stream.hip — the host pipeline in front of the batched kernels: loci are submitted as they arrive — one
// region at a time is how the reference's caller produces them (bam_processor.cpp:550-617, genotyper_bam_processor.cpp:229-243) —
// and results come back in submission order, which is region order, the order the VCF writer needs (vcf_writer.cpp:7-36).
//
//   submit ──► pending batch ──(threshold / flush)──► ready queue ──► worker thread: prepare_batch on the host threads, tables to a
//   pinned staging block, H2D on the copy stream, the phase kernels on the compute stream, D2H of aln_probs on the copy stream
//   ──► in-flight queue ──► hipstr_stream_next: waits for the front batch's D2H event, applies the reference's output contract
//   for the ticket's loci into the caller's arrays.
//
// While batch k runs on the device the worker prepares and uploads batch k+1 (launches are asynchronous), so with two or more
// slots the device only idles when the host cannot keep up.  Host code only; every byte of arithmetic is in the kernels.
#include <hip/hip_runtime.h>
#include <atomic>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <pthread.h>
#include <time.h>
#include <vector>

#include "../../include/hipstr_hmm.h"
#include "api_internal.h"
#include "prep.h"

namespace {

double thread_cpu_now(){ timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9*(double)ts.tv_nsec; }

// A growing byte buffer in pinned host memory (from the context's block cache; plain malloc without a context): the reads' bases and
// qualities of a stream batch are copied ONCE, from the caller's arrays into it, and travel to the device from where they lie.
struct PinBuf {
  hipstr::Ctx* ctx = NULL; char* p = NULL; size_t n = 0, cap = 0;
  PinBuf(){}
  PinBuf(const PinBuf&) = delete; PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf(){ drop(); }
  void drop(){ if (p){ if (ctx) hipstr::pin_free(ctx, p); else free(p); } p = NULL; n = cap = 0; }
  bool reserve(size_t want){
    if (want <= cap) return true;
    const size_t nc = std::max<size_t>(std::max(want, cap + cap/2), (size_t)1 << 16);
    if (ctx) hipstr::api_bind(ctx);
    char* q = ctx ? (char*)hipstr::pin_alloc(ctx, nc) : (char*)malloc(nc);
    if (!q) return false;
    if (n) memcpy(q, p, n);
    if (p){ if (ctx) hipstr::pin_free(ctx, p); else free(p); }
    p = q; cap = nc;
    return true;
  }
  char* grow(size_t add){ if (!reserve(n + add)) return NULL; char* at = p + n; n += add; return at; }
  size_t size() const { return n; }
  const char* data() const { return p ? p : ""; }
};

// A batch the library owns: deep copies of the submitted arrays, concatenated.
struct OwnedBatch {
  std::vector<int32_t> blk_start, blk_end, blk_nopts, period, opt_off, hap_off, read_off, base_off, read_start, cigar_off, cigar_len;
  std::vector<int32_t> seed;            // per read: the seed base the submission's check computed (HIPSTR_SEED_AUTO where it did not)
  std::vector<double> stutter;
  std::vector<uint8_t> realign_hap, realign_read;
  std::string seq, cigar_op;
  PinBuf bases, quals;
  hipstr_batch_t view;
  struct Ticket { int64_t id; int32_t l0, l1, r0, r1; int64_t out0, out1; };
  std::vector<Ticket> tickets;
  int64_t n_out = 0, work = 0;          // doubles of output; (reads x haplotypes) submitted
  // the longest realigned read and the longest STR allele so far: the per-read STR kernels size their LDS by the batch's maxima, so a locus
  // that would push the combined need over HS_LDS_LIMIT goes into the next batch (each locus fits alone: check_locus)
  int max_read = 0, max_B = 1;
  bool fits(int longest_read, int longest_B) const { return hs_str_kernel_lds_bytes(std::max(max_read, longest_read), std::max(max_B, longest_B)) <= HS_LDS_LIMIT; }
  void note(int longest_read, int longest_B){ max_read = std::max(max_read, longest_read); max_B = std::max(max_B, longest_B); }
  // Copies of bases / qualities still running OUTSIDE the stream's lock (append_run reserves their place under the lock and copies
  // afterwards, so that collectors and workers are not held up for the milliseconds a megabyte copy takes).  A buffer is not moved
  // (grown) and the batch not prepared while this is non-zero.
  std::atomic<int> writers{0};
  struct Deferred { char* dst; const char* src; size_t n; };
  void wait_writers() const { while (writers.load(std::memory_order_acquire) != 0) std::this_thread::yield(); }
  explicit OwnedBatch(hipstr::Ctx* pin_ctx = NULL){ bases.ctx = pin_ctx; quals.ctx = pin_ctx; reset(); }
  // empty again, storage kept (the stream recycles its batches: their vectors and pinned buffers are page-faulted and registered once)
  void reset(){
    blk_start.clear(); blk_end.clear(); blk_nopts.clear(); period.clear(); opt_off.clear(); hap_off.clear(); read_off.clear(); base_off.clear();
    read_start.clear(); cigar_off.clear(); cigar_len.clear(); seed.clear(); stutter.clear(); realign_hap.clear(); realign_read.clear();
    seq.clear(); cigar_op.clear(); bases.n = 0; quals.n = 0; tickets.clear(); n_out = 0; work = 0; max_read = 0; max_B = 1;
    opt_off.push_back(0); hap_off.push_back(0); read_off.push_back(0); base_off.push_back(0); cigar_off.push_back(0);
  }
  void add_seeds(const int32_t* seeds, int r0, int r1){        // seeds: indexed by the caller's read index, or NULL
    if (seeds) seed.insert(seed.end(), seeds + r0, seeds + r1); else seed.insert(seed.end(), (size_t)(r1 - r0), HIPSTR_SEED_AUTO);
  }

  // 0 or an error message
  const char* append(const hipstr_batch_t* b, int64_t ticket, const int32_t* seeds = NULL){
    if (b->n_loci < 0) return "negative locus count";
    const int n = b->n_loci;
    int64_t n_opts = 0;
    for (int i = 0; i < 3*n; i++){ if (b->blk_nopts[i] < 1) return "haplotype block without options"; n_opts += b->blk_nopts[i]; }
    const int32_t n_haps = n ? b->hap_off[n] : 0, n_reads = n ? b->read_off[n] : 0;
    const int32_t n_seq = n_opts ? b->opt_off[n_opts] : 0, n_bases = n_reads ? b->base_off[n_reads] : 0, n_cig = n_reads ? b->cigar_off[n_reads] : 0;
    if ((int64_t)bases.size() + n_bases > INT32_MAX || (int64_t)seq.size() + n_seq > INT32_MAX) return "pending batch exceeds 2 GiB of bases";
    Ticket t; t.id = ticket; t.l0 = (int32_t)period.size(); t.l1 = t.l0 + n; t.r0 = read_off.back(); t.r1 = t.r0 + n_reads; t.out0 = n_out;
    blk_start.insert(blk_start.end(), b->blk_start, b->blk_start + 3*n); blk_end.insert(blk_end.end(), b->blk_end, b->blk_end + 3*n);
    blk_nopts.insert(blk_nopts.end(), b->blk_nopts, b->blk_nopts + 3*n); period.insert(period.end(), b->period, b->period + n);
    stutter.insert(stutter.end(), b->stutter, b->stutter + 6*n);
    const int32_t seq0 = (int32_t)seq.size(), hap0 = hap_off.back(), rd0 = read_off.back(), base0 = (int32_t)bases.size(), cig0 = (int32_t)cigar_op.size();
    for (int64_t i = 1; i <= n_opts; i++) opt_off.push_back(seq0 + b->opt_off[i]);
    seq.append(b->seq, n_seq);
    for (int l = 1; l <= n; l++){ hap_off.push_back(hap0 + b->hap_off[l]); read_off.push_back(rd0 + b->read_off[l]); }
    for (int l = 0; l < n; l++){
      const int64_t P = b->read_off[l+1] - b->read_off[l], A = b->hap_off[l+1] - b->hap_off[l];
      if (P < 0 || A < 1) return "inconsistent read_off / hap_off";
      n_out += P*A; work += P*A;
    }
    if (b->realign_hap) realign_hap.insert(realign_hap.end(), b->realign_hap, b->realign_hap + n_haps); else realign_hap.insert(realign_hap.end(), n_haps, 1);
    for (int r = 1; r <= n_reads; r++){ base_off.push_back(base0 + b->base_off[r]); cigar_off.push_back(cig0 + b->cigar_off[r]); }
    if (bases.size() + (size_t)n_bases > bases.cap || quals.size() + (size_t)n_bases > quals.cap) wait_writers();
    { char* pb = bases.grow((size_t)n_bases); char* pq = quals.grow((size_t)n_bases);
      if (!pb || !pq) return "out of pinned host memory for the reads";
      memcpy(pb, b->bases, (size_t)n_bases); memcpy(pq, b->quals, (size_t)n_bases); }
    read_start.insert(read_start.end(), b->read_start, b->read_start + n_reads);
    cigar_op.append(b->cigar_op, n_cig); cigar_len.insert(cigar_len.end(), b->cigar_len, b->cigar_len + n_cig);
    if (b->realign_read) realign_read.insert(realign_read.end(), b->realign_read, b->realign_read + n_reads); else realign_read.insert(realign_read.end(), n_reads, 1);
    add_seeds(seeds, 0, n_reads);
    t.out1 = n_out;
    tickets.push_back(t);
    return NULL;
  }
  // Locus l of `b` (whose first block option is opt0 in opt_off) as a submission of its own: what append() does for a one-locus batch,
  // straight from the caller's arrays.
  const char* append_locus(const hipstr_batch_t* b, int l, int opt0, int64_t ticket, const int32_t* seeds = NULL){
    const int nopt = b->blk_nopts[3*l] + b->blk_nopts[3*l+1] + b->blk_nopts[3*l+2];
    const int r0 = b->read_off[l], r1 = b->read_off[l+1], h0 = b->hap_off[l], h1 = b->hap_off[l+1];
    const int32_t s0 = b->opt_off[opt0], s1 = b->opt_off[opt0 + nopt], b0 = b->base_off[r0], b1 = b->base_off[r1], c0 = b->cigar_off[r0], c1 = b->cigar_off[r1];
    const int64_t P = r1 - r0, A = h1 - h0;
    if (P < 0 || A < 1) return "inconsistent read_off / hap_off";
    if ((int64_t)bases.size() + (b1 - b0) > INT32_MAX || (int64_t)seq.size() + (s1 - s0) > INT32_MAX) return "pending batch exceeds 2 GiB of bases";
    Ticket t; t.id = ticket; t.l0 = (int32_t)period.size(); t.l1 = t.l0 + 1; t.r0 = read_off.back(); t.r1 = t.r0 + (int32_t)P; t.out0 = n_out;
    blk_start.insert(blk_start.end(), b->blk_start + 3*l, b->blk_start + 3*l + 3); blk_end.insert(blk_end.end(), b->blk_end + 3*l, b->blk_end + 3*l + 3);
    blk_nopts.insert(blk_nopts.end(), b->blk_nopts + 3*l, b->blk_nopts + 3*l + 3); period.push_back(b->period[l]);
    stutter.insert(stutter.end(), b->stutter + 6*l, b->stutter + 6*l + 6);
    const int32_t seq0 = (int32_t)seq.size() - s0, base0 = (int32_t)bases.size() - b0, cig0 = (int32_t)cigar_op.size() - c0;
    for (int i = 1; i <= nopt; i++) opt_off.push_back(seq0 + b->opt_off[opt0 + i]);
    seq.append(b->seq + s0, s1 - s0);
    hap_off.push_back(hap_off.back() + (int32_t)A); read_off.push_back(read_off.back() + (int32_t)P);
    n_out += P*A; work += P*A;
    if (b->realign_hap) realign_hap.insert(realign_hap.end(), b->realign_hap + h0, b->realign_hap + h1); else realign_hap.insert(realign_hap.end(), (size_t)A, 1);
    for (int r = r0 + 1; r <= r1; r++){ base_off.push_back(base0 + b->base_off[r]); cigar_off.push_back(cig0 + b->cigar_off[r]); }
    if (bases.size() + (size_t)(b1 - b0) > bases.cap || quals.size() + (size_t)(b1 - b0) > quals.cap) wait_writers();
    { char* pb = bases.grow((size_t)(b1 - b0)); char* pq = quals.grow((size_t)(b1 - b0));
      if (!pb || !pq) return "out of pinned host memory for the reads";
      memcpy(pb, b->bases + b0, (size_t)(b1 - b0)); memcpy(pq, b->quals + b0, (size_t)(b1 - b0)); }
    read_start.insert(read_start.end(), b->read_start + r0, b->read_start + r1);
    cigar_op.append(b->cigar_op + c0, c1 - c0); cigar_len.insert(cigar_len.end(), b->cigar_len + c0, b->cigar_len + c1);
    if (b->realign_read) realign_read.insert(realign_read.end(), b->realign_read + r0, b->realign_read + r1); else realign_read.insert(realign_read.end(), (size_t)P, 1);
    add_seeds(seeds, r0, r1);
    t.out1 = n_out;
    tickets.push_back(t);
    return NULL;
  }
  // Loci [l0, l1) of `b`, every one a submission of its own with consecutive tickets from ticket0: what append_locus does l1 - l0
  // times, but the loci of a batch lie next to each other in every array, so each pool takes ONE copy (the reads' bases and qualities —
  // 12 KB per 40-read locus — spread over the host threads) and the offset arrays one rebasing pass.  opt0[l] = first block option of locus l.
  // (deferred: the two large copies are left to the caller, to be made after it has released the stream's lock — writers was incremented
  //  for them and is decremented by the caller when they are done)
  const char* append_run(const hipstr_batch_t* b, int l0, int l1, const int* opt0, int64_t ticket0, std::vector< std::pair<int64_t,int64_t> >& sizes, const int32_t* seeds,
                         Deferred deferred[2]){
    const int n = l1 - l0;
    const int r0 = b->read_off[l0], r1 = b->read_off[l1], h0 = b->hap_off[l0], h1 = b->hap_off[l1];
    const int32_t s0 = b->opt_off[opt0[l0]], s1 = b->opt_off[opt0[l1]], b0 = b->base_off[r0], b1 = b->base_off[r1], c0 = b->cigar_off[r0], c1 = b->cigar_off[r1];
    if (r1 < r0 || h1 < h0 + n) return "inconsistent read_off / hap_off";
    if ((int64_t)bases.size() + (b1 - b0) > INT32_MAX || (int64_t)seq.size() + (s1 - s0) > INT32_MAX) return "pending batch exceeds 2 GiB of bases";
    for (int l = l0; l < l1; l++)                       // (nothing is appended unless everything can be)
      if (b->read_off[l+1] < b->read_off[l] || b->hap_off[l+1] - b->hap_off[l] < 1) return "inconsistent read_off / hap_off";
    if (bases.size() + (size_t)(b1 - b0) > bases.cap || quals.size() + (size_t)(b1 - b0) > quals.cap) wait_writers();      // growing moves the buffer
    if (!bases.reserve(bases.size() + (size_t)(b1 - b0)) || !quals.reserve(quals.size() + (size_t)(b1 - b0))) return "out of pinned host memory for the reads";
    const int32_t loc_base = (int32_t)period.size();
    blk_start.insert(blk_start.end(), b->blk_start + 3*l0, b->blk_start + 3*l1); blk_end.insert(blk_end.end(), b->blk_end + 3*l0, b->blk_end + 3*l1);
    blk_nopts.insert(blk_nopts.end(), b->blk_nopts + 3*l0, b->blk_nopts + 3*l1); period.insert(period.end(), b->period + l0, b->period + l1);
    stutter.insert(stutter.end(), b->stutter + 6*l0, b->stutter + 6*l1);
    const int32_t seq0 = (int32_t)seq.size() - s0, base0 = (int32_t)bases.size() - b0, cig0 = (int32_t)cigar_op.size() - c0;
    for (int i = opt0[l0] + 1; i <= opt0[l1]; i++) opt_off.push_back(seq0 + b->opt_off[i]);
    seq.append(b->seq + s0, s1 - s0);
    const int32_t hap0 = hap_off.back() - h0, rd0 = read_off.back() - r0;
    for (int l = l0; l < l1; l++){
      const int64_t P = b->read_off[l+1] - b->read_off[l], A = b->hap_off[l+1] - b->hap_off[l];
      Ticket t; t.id = ticket0 + (l - l0); t.l0 = loc_base + (l - l0); t.l1 = t.l0 + 1; t.r0 = rd0 + b->read_off[l]; t.r1 = t.r0 + (int32_t)P; t.out0 = n_out;
      n_out += P*A; work += P*A; t.out1 = n_out;
      tickets.push_back(t);
      sizes.push_back(std::make_pair(P*A, P));
      hap_off.push_back(hap0 + b->hap_off[l+1]); read_off.push_back(rd0 + b->read_off[l+1]);
    }
    if (b->realign_hap) realign_hap.insert(realign_hap.end(), b->realign_hap + h0, b->realign_hap + h1); else realign_hap.insert(realign_hap.end(), (size_t)(h1 - h0), 1);
    {
      const size_t at = base_off.size(); base_off.resize(at + (r1 - r0)); cigar_off.resize(at + (r1 - r0));
      for (int r = r0 + 1; r <= r1; r++){ base_off[at + (r - r0 - 1)] = base0 + b->base_off[r]; cigar_off[at + (r - r0 - 1)] = cig0 + b->cigar_off[r]; }
    }
    {
      const size_t nb = (size_t)(b1 - b0);
      char* pb = bases.grow(nb); char* pq = quals.grow(nb);          // (reserved above: cannot fail)
      deferred[0] = Deferred{pb, b->bases + b0, nb}; deferred[1] = Deferred{pq, b->quals + b0, nb};
      writers.fetch_add(1, std::memory_order_acq_rel);
    }
    read_start.insert(read_start.end(), b->read_start + r0, b->read_start + r1);
    cigar_op.append(b->cigar_op + c0, c1 - c0); cigar_len.insert(cigar_len.end(), b->cigar_len + c0, b->cigar_len + c1);
    if (b->realign_read) realign_read.insert(realign_read.end(), b->realign_read + r0, b->realign_read + r1); else realign_read.insert(realign_read.end(), (size_t)(r1 - r0), 1);
    add_seeds(seeds, r0, r1);
    return NULL;
  }
  const hipstr_batch_t* finish(){
    if (cigar_len.empty()) cigar_len.push_back(0);
    view.n_loci = (int32_t)period.size();
    view.blk_start = blk_start.data(); view.blk_end = blk_end.data(); view.blk_nopts = blk_nopts.data(); view.period = period.data();
    view.stutter = stutter.data(); view.opt_off = opt_off.data(); view.seq = seq.data(); view.hap_off = hap_off.data();
    view.realign_hap = realign_hap.data(); view.read_off = read_off.data(); view.base_off = base_off.data(); view.bases = bases.data();
    view.quals = quals.data(); view.read_start = read_start.data(); view.cigar_off = cigar_off.data(); view.cigar_op = cigar_op.data();
    view.cigar_len = cigar_len.data(); view.realign_read = realign_read.data();
    return &view;
  }
};

struct InFlight {
  OwnedBatch* ob = NULL;
  hipstr_dev_batch_t* dev = NULL;
  std::vector<uint8_t> taken;     // per ticket of the batch
  size_t n_taken = 0;
  int busy = 0;                   // collectors currently copying out of this batch
  bool failed = false, landed = false;
  std::mutex land_m;              // the first collector waits for the copy back; the others wait for it
  std::string err;
};

}  // namespace

struct hipstr_stream {
  hipstr::Ctx* ctx = NULL;
  hipStream_t copy_stream = NULL, d2h_stream = NULL;     // tables to the device / results back: neither waits for the other
  // (the kernels of consecutive batches on alternating streams instead of the context's one: tried at the
  // end of round 4 against the 8 % a stream of 2 Mi batches loses to the resident rate (the tails and gaps of 60 kernels per pass instead
  // of 8): no effect, 129.5 ms per pass with 1, 2 and 3 streams — a persistent trailing-flank kernel holds every SIMD's registers until
  // it ends.  What helped is fewer, larger batches (below).  One stream; the arrays below are what is left of the experiment.)
  hipStream_t compute[4] = {NULL, NULL, NULL, NULL};
  int n_compute = 0;
  std::atomic<unsigned> launch_seq{0};
  int slots = 6;
  int64_t batch_work = (int64_t)2 << 20;       // (2 Mi pairs: a 30x batch's tables then stay within the last-level cache while they are built and packed)
  // ... for batches of MANY loci.  What the device sees is pairs: with heavy loci (the north-star shape: 16 000 pairs each) 2 Mi pairs are
  // 131 loci, every kernel of the batch is a sixth of a millisecond to a few milliseconds long, and the batch runs at 127 M pairs/s where
  // 1000 loci run at 134 (end of round 4: resident rate by batch size; end to end 129.5 -> 124.4 ms per pass with 8 Mi batches).  A batch
  // therefore closes at batch_work pairs only once it holds 2048 loci, else at `big_mult` times that; the batches in flight are bounded
  // by their pairs (slots x batch_work, two batches at least) so that the workspaces in flight do not grow with the batch.
  // (only where the caller left the batch size to the library; with few host threads — a rank's two at eight GPUs — a batch is prepared
  //  almost serially and the large ones starve the device: 122.6 -> 120.2 M pairs/s measured, so the factor follows the host threads)
  bool adaptive = true;
  int big_mult = 1;
  // (and not for loci of more than batch_work / 64 pairs each — configs[3]'s 1000-sample loci, 160 000 pairs: their batches are a dozen loci
  //  whose kernels are long already and whose preparation is milliseconds per locus; with 8 Mi batches a pass of 100 such loci is two
  //  batches and nothing overlaps: 118 -> 97 M pairs/s measured)
  bool full(int64_t work, size_t n_loci) const {
    if (work < batch_work) return false;
    if (!adaptive || n_loci >= 2048 || (int64_t)n_loci * batch_work < 64 * work) return true;
    // (at most a third of the pairs allowed in flight: three such batches hold what `slots` small ones did before — a large batch must not
    //  raise the stream's memory, which at 2.3 KB of workspace per pair is 38 GB for the default 16 Mi pairs in flight)
    return work >= std::min(batch_work * big_mult, std::max(batch_work, (int64_t)slots * batch_work / 3));
  }
  int64_t in_worker_work = 0;       // pairs of the batches the workers have popped but not yet pushed to `flying`
  std::mutex m;
  std::condition_variable cv_work, cv_done, cv_slots;
  OwnedBatch* pending = NULL;
  std::vector<OwnedBatch*> spare;  // retired batches, emptied: the next pending batch starts with their storage (vectors, pinned read buffers)
  std::deque<OwnedBatch*> ready;
  std::deque<InFlight*> flying;   // launched (or failed), in submission order
  int in_worker = 0;              // batches the worker has popped but not yet pushed to `flying`
  int waiting = 0;                // collectors blocked on a ticket that has not been launched yet
  std::multiset<int64_t> wait_tickets;   // ... and the tickets they wait for
  int64_t next_ticket = 0, next_deliver = 0;      // next_deliver: the lowest ticket not collected yet
  std::set<int64_t> taken_ahead;                  // tickets above next_deliver that were collected out of order (hipstr_stream_take)
  std::vector< std::pair<int64_t,int64_t> > sizes;     // per ticket not yet delivered: (n_out, n_reads), indexed by ticket - sizes_base
  int64_t sizes_base = 0;
  bool closing = false;
  std::vector<std::thread> workers;              // each prepares, uploads and launches whole batches; their kernels share the context's stream
  hipstr_stream_stats_t stats;
  std::chrono::steady_clock::time_point t_open;
};

namespace {

void flush_locked(hipstr_stream* s);

// (with s->m held)
OwnedBatch* new_batch_locked(hipstr_stream* s){
  if (!s->spare.empty()){ OwnedBatch* ob = s->spare.back(); s->spare.pop_back(); return ob; }
  return new OwnedBatch(s->ctx);
}
void retire_batch(hipstr_stream* s, OwnedBatch* ob){
  ob->reset();
  {
    std::lock_guard<std::mutex> g(s->m);
    if (!s->closing && (int)s->spare.size() < s->slots + 2){ s->spare.push_back(ob); return; }
  }
  delete ob;
}

void worker_loop(hipstr_stream* s, int n_workers){
  pthread_setname_np(pthread_self(), "hipstr-worker");
  hipstr::api_bind(s->ctx);
  // the workers prepare different batches at the same time: each takes its share of the host threads (one thread each on a two-core
  // allowance — then a batch is prepared without fragments, merges or hand-overs to pool threads)
  {
    int budget = std::max(1, hipstr::host_threads() / std::max(1, n_workers));
    if (const char* e = getenv("HIPSTR_STREAM_WORKER_THREADS")){ const int v = atoi(e); if (v >= 1) budget = v; }
    hipstr::set_thread_budget(budget);
  }
  for (;;){
    OwnedBatch* ob = NULL;
    {
      std::unique_lock<std::mutex> g(s->m);
      // work = a batch that was sent, or — when a collector is waiting for a ticket that still sits in the pending batch — the pending
      // batch as it is: whatever accumulated while the previous batch was being prepared goes out together
      // (only when no other worker is preparing a batch: the ticket waited for is most likely in that one, and the pending batch keeps filling)
      auto have_work = [&]{ return !s->ready.empty() || (s->waiting > 0 && s->in_worker == 0 && s->pending && !s->pending->tickets.empty()); };
      // `slots` bounds the batches in flight — unless a collector waits for a ticket that is not launched yet while every slot is held
      // by batches with uncollected EARLIER tickets (tickets may be taken in any order): then the batch goes out all the same
      // (only for a batch that brings an awaited ticket closer: one whose first ticket is not beyond the latest awaited one — while the
      //  awaited ticket sits in a batch another worker is still preparing, nothing overshoots)
      auto next_first = [&]() -> int64_t {
        if (!s->ready.empty()) return s->ready.front()->tickets.front().id;
        if (s->pending && !s->pending->tickets.empty()) return s->pending->tickets.front().id;
        return INT64_MAX;
      };
      auto may_overshoot = [&]{ return !s->wait_tickets.empty() && next_first() <= *s->wait_tickets.rbegin(); };
      auto room = [&]{                                   // batches in flight: at most `slots`, and at most slots x batch_work pairs beyond the second batch
        const int n_fly = (int)s->flying.size() + s->in_worker;
        if (n_fly >= s->slots) return false;
        if (n_fly < 2) return true;
        int64_t w = s->in_worker_work;
        for (const InFlight* f : s->flying) w += f->ob->work;
        const int64_t next = !s->ready.empty() ? s->ready.front()->work : (s->pending ? s->pending->work : 0);
        // (a caller's own batch size does not multiply the memory in flight: slots x 2 Mi pairs as with the default — ~40 GB of workspaces at the
        //  north-star shape, twice that for loci of few alleles —, two batches at least.  Round 6: 8 slots x 8 Mi pairs of 8-allele loci ran the
        //  device out of memory, tools/r06_rt_long.sh)
        const int64_t cap = std::max((int64_t)s->slots * std::min(s->batch_work, (int64_t)2 << 20), 2 * s->batch_work);
        return w + next <= cap;
      };
      s->cv_work.wait(g, [&]{ return s->closing || (have_work() && (room() || may_overshoot())); });
      if (s->closing) return;
      if (s->ready.empty()) flush_locked(s);
      ob = s->ready.front(); s->ready.pop_front(); s->in_worker++; s->in_worker_work += ob->work;
    }
    ob->wait_writers();              // copies into the batch that a submitter is still making outside the lock
    InFlight* f = new InFlight(); f->ob = ob; f->taken.assign(ob->tickets.size(), 0);
    const auto t0 = std::chrono::steady_clock::now();
    const double c0 = thread_cpu_now();
    const hipStream_t cs = s->n_compute > 0 ? s->compute[s->launch_seq.fetch_add(1) % (unsigned)s->n_compute] : hipstr::ctx_stream(s->ctx);
    f->dev = hipstr::upload_on(s->ctx, ob->finish(), ob->seed.data(), s->copy_stream, cs, true);
    if (!f->dev){ f->failed = true; f->err = hipstr_last_error(); }
    else if (hipstr_hmm_align(f->dev, NULL) != 0 || hipstr::fetch_begin(f->dev, cs, s->d2h_stream) != 0){
      f->failed = true; f->err = hipstr_last_error();
    }
    const double host_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (getenv("HIPSTR_TIMING"))
      fprintf(stderr, "stream: batch of %zu tickets (%lld pairs) prepared + launched %.3f .. %.3f ms after open\n", ob->tickets.size(), (long long)ob->work,
              1e3*std::chrono::duration<double>(t0 - s->t_open).count(), 1e3*std::chrono::duration<double>(std::chrono::steady_clock::now() - s->t_open).count());
    {
      // (passing by: the copy back of earlier batches whose kernels have finished meanwhile — api.hip fetch_begin)
      std::lock_guard<std::mutex> g(s->m);
      for (InFlight* e : s->flying) if (e->dev && !e->failed && !e->landed) hipstr::fetch_poll(e->dev);
    }
    {
      std::lock_guard<std::mutex> g(s->m);
      s->flying.push_back(f); s->in_worker--; s->in_worker_work -= ob->work;
      s->stats.batches++; s->stats.host_seconds += host_s; s->stats.alignment_slots += ob->work;
      const double cpu = thread_cpu_now() - c0, prep = f->dev ? hipstr::batch_prepare_seconds(f->dev) : 0.0;
      s->stats.cpu_prepare_seconds += std::min(cpu, prep); s->stats.cpu_upload_seconds += std::max(0.0, cpu - prep);
    }
    s->cv_done.notify_all();
  }
}

void flush_locked(hipstr_stream* s){
  if (s->pending && !s->pending->tickets.empty()){ s->ready.push_back(s->pending); s->pending = NULL; s->cv_work.notify_one(); }
}

}  // namespace

extern "C" {

hipstr_stream_t* hipstr_stream_open(const hipstr_stream_opts_t* opts){
  const int device = opts ? opts->device : 0;
  if (hipstr_hmm_init(device) != 0) return NULL;
  hipstr_stream* s = new hipstr_stream();
  s->ctx = hipstr::api_current_ctx();
  if (!s->ctx){ delete s; return NULL; }
  if (opts && opts->slots > 0) s->slots = opts->slots;
  if (opts && opts->batch_alignments > 0){ s->batch_work = opts->batch_alignments; s->adaptive = false; }
  {
    const int ht = hipstr::host_threads();
    s->big_mult = ht >= 8 ? 4 : (ht >= 4 ? 2 : 1);
  }
  if (hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&s->d2h_stream, hipStreamNonBlocking) != hipSuccess){
    hipstr::api_fail("hipStreamCreate failed"); delete s; return NULL; }
  {
    int nc = 1;
    if (nc > 1) for (int i = 0; i < nc; i++){
      if (hipStreamCreateWithFlags(&s->compute[i], hipStreamNonBlocking) != hipSuccess){
        hipstr::api_fail("hipStreamCreate failed");
        for (int k = 0; k < s->n_compute; k++) hipStreamDestroy(s->compute[k]);
        hipStreamDestroy(s->copy_stream); hipStreamDestroy(s->d2h_stream); delete s; return NULL;
      }
      s->n_compute = i + 1;
    }
  }
  memset(&s->stats, 0, sizeof s->stats);
  s->t_open = std::chrono::steady_clock::now();
  // Three workers by default (HIPSTR_STREAM_WORKERS): preparing a batch has serial stretches between its parallel ones (merging the
  // fragments, the launch plan, packing the staging block), during which a second batch's parallel stretches keep the host threads
  // busy; with small loci (30x, ten alleles) the host side is what bounds the stream.
  int nw = 3;
  if (const char* e = getenv("HIPSTR_STREAM_WORKERS")){ const int v = atoi(e); if (v >= 1 && v <= 8) nw = v; }
  for (int w = 0; w < nw; w++) s->workers.emplace_back(worker_loop, s, nw);
  return s;
}

namespace { struct CpuAdd { hipstr_stream* s; double* slot; double c0; CpuAdd(hipstr_stream* s_, double* slot_) : s(s_), slot(slot_), c0(thread_cpu_now()) {}
                    ~CpuAdd(){ const double d = thread_cpu_now() - c0; std::lock_guard<std::mutex> g(s->m); *slot += d; } }; }

int64_t hipstr_stream_submit(hipstr_stream_t* s, const hipstr_batch_t* loci){
  hipstr::ApiTimer prof_t(hipstr::PB_STREAM_SUBMIT);
  if (!s || !loci){ hipstr::api_fail("null argument"); return -1; }
  CpuAdd cpu_t(s, &s->stats.cpu_submit_seconds);
  // a submission that prepare_batch would refuse is turned away here, before it shares a batch with others; the seed bases the check
  // computes go along with the reads
  thread_local std::vector<int32_t> seeds;
  int sub_read = 0, sub_B = 1;
  {
    std::string why;
    if (hipstr::validate_tables(loci, why)){ hipstr::api_fail(why); return -1; }
    seeds.resize(loci->n_loci > 0 ? (size_t)std::max(0, loci->read_off[loci->n_loci]) : 0);
    int cursor = 0;
    for (int l = 0; l < loci->n_loci; l++){
      int dims[2];
      if (hipstr::check_locus(loci, l, &cursor, why, seeds.data(), dims)){ hipstr::api_fail(why); return -1; }
      sub_read = std::max(sub_read, dims[0]); sub_B = std::max(sub_B, dims[1]);
    }
    // (a submission is one ticket, hence one batch: its loci must fit the per-read STR kernels' LDS together)
    if (hs_str_kernel_lds_bytes(sub_read, sub_B) > HS_LDS_LIMIT){ hipstr::api_fail("the loci of this submission together need more than 160 KiB of LDS per workgroup (its longest read with its longest STR allele): submit them separately"); return -1; }
  }
  std::lock_guard<std::mutex> g(s->m);
  if (s->closing){ hipstr::api_fail("stream is closing"); return -1; }
  if (s->pending && !s->pending->tickets.empty() && !s->pending->fits(sub_read, sub_B)) flush_locked(s);
  if (!s->pending) s->pending = new_batch_locked(s);
  s->pending->note(sub_read, sub_B);
  const int64_t ticket = s->next_ticket;
  const int64_t out_before = s->pending->n_out; const int32_t reads_before = s->pending->read_off.back();
  if (const char* why = s->pending->append(loci, ticket, seeds.data())){ hipstr::api_fail(why); return -1; }
  s->next_ticket++;
  s->sizes.push_back(std::make_pair(s->pending->n_out - out_before, (int64_t)(s->pending->read_off.back() - reads_before)));
  if (s->full(s->pending->work, s->pending->period.size())) flush_locked(s);
  return ticket;
}

// Every locus of `loci` as its own submission, in order (what a region loop does, without a call per region crossing a language
// boundary): *first_ticket = the ticket of locus 0, the rest follow consecutively.  Stops at the first locus that is refused.
int hipstr_stream_submit_each(hipstr_stream_t* s, const hipstr_batch_t* loci, int64_t* first_ticket){
  if (!s || !loci) return hipstr::api_fail("null argument");
  CpuAdd cpu_t(s, &s->stats.cpu_submit_seconds);
  { std::string bad; if (hipstr::validate_tables(loci, bad)) return hipstr::api_fail(bad); }
  const int n = loci->n_loci;
  const auto t_sub0 = std::chrono::steady_clock::now();
  // A locus that prepare_batch would refuse is turned away here, before it shares a batch with others.  The checks (a seed per read
  // among them) are independent per locus: a large call spreads them over the host threads; the appends below only copy.
  std::vector<int> opt0((size_t)n + 1, 0);
  for (int l = 0; l < n; l++){ int c = 0; for (int k = 0; k < 3; k++) c += std::max(0, loci->blk_nopts[3*l+k]); opt0[l+1] = opt0[l] + c; }
  std::atomic<int> first_bad(n);
  std::mutex why_m; std::string why; int why_l = n;
  std::vector<int32_t> seeds(n > 0 ? (size_t)std::max(0, loci->read_off[n]) : 0);        // the checks compute every read's seed base: kept for the preparation
  std::vector<int> dims(2*(size_t)std::max(n, 0), 0);                                      // per locus: longest realigned read, longest STR allele
  const int CH = 128, n_ch = (n + CH - 1)/CH;
  hipstr::parallel_for(n_ch, n >= 4*CH ? hipstr::host_threads() : 1, [&](int c){
    for (int l = c*CH; l < std::min(n, (c+1)*CH); l++){
      if (l > first_bad.load(std::memory_order_relaxed)) return;
      int cur = opt0[l]; std::string w;
      if (hipstr::check_locus(loci, l, &cur, w, seeds.data(), &dims[2*(size_t)l])){
        std::lock_guard<std::mutex> g(why_m);
        if (l < why_l){ why_l = l; why = w; }
        int fb = first_bad.load(); while (l < fb && !first_bad.compare_exchange_weak(fb, l)){}
        return;
      }
    }
  });
  const int n_ok_total = std::min(n, why_l);
  const auto t_chk = std::chrono::steady_clock::now();
  // The loci go in as runs: a run ends where the pending batch reaches its size (it is handed to the workers there) or after 1024 loci
  // (the collectors take the stream's lock between runs).
  for (int l0 = 0; l0 < n_ok_total; ){
    OwnedBatch::Deferred cp[2]; OwnedBatch* ob = NULL; int l1 = l0;
    {
      std::lock_guard<std::mutex> g(s->m);
      if (s->closing) return hipstr::api_fail("stream is closing");
      // (a locus that does not fit the pending batch's LDS figure closes it: every locus fits alone)
      if (s->pending && !s->pending->tickets.empty() && !s->pending->fits(dims[2*(size_t)l0], dims[2*(size_t)l0 + 1])) flush_locked(s);
      if (!s->pending) s->pending = new_batch_locked(s);
      int64_t w = s->pending->work;
      const size_t nl0 = s->pending->period.size();
      while (l1 < n_ok_total && l1 - l0 < 1024 && !s->full(w, nl0 + (size_t)(l1 - l0)) && s->pending->fits(dims[2*(size_t)l1], dims[2*(size_t)l1 + 1])){
        s->pending->note(dims[2*(size_t)l1], dims[2*(size_t)l1 + 1]);
        w += (int64_t)(loci->read_off[l1+1] - loci->read_off[l1])*(loci->hap_off[l1+1] - loci->hap_off[l1]); l1++;
      }
      const int64_t ticket0 = s->next_ticket;
      ob = s->pending;
      if (const char* w2 = ob->append_run(loci, l0, l1, opt0.data(), ticket0, s->sizes, seeds.data(), cp)) return hipstr::api_fail(w2);
      s->next_ticket += l1 - l0;
      if (l0 == 0 && first_ticket) *first_ticket = ticket0;
      if (s->full(s->pending->work, s->pending->period.size())) flush_locked(s);
    }
    // the reads' bases and qualities (12 KB per 40-read locus), outside the lock: a worker that picks the batch up waits for `writers`
    {
      const size_t CH = (size_t)1 << 20; const size_t nb = cp[0].n; const int n_ch = (int)((nb + CH - 1)/CH);
      hipstr::parallel_for(2*n_ch, nb > 4*CH ? hipstr::host_threads() : 1, [&](int i){
        const size_t o = (size_t)(i >> 1)*CH, m = std::min(CH, nb - o);
        memcpy(cp[i & 1].dst + o, cp[i & 1].src + o, m);
      });
      ob->writers.fetch_sub(1, std::memory_order_acq_rel);
    }
    l0 = l1;
  }
  if (n >= 1024 && getenv("HIPSTR_TIMING"))
    fprintf(stderr, "stream: submit_each of %d loci: checks %.3f ms, appends %.3f ms\n", n, 1e3*std::chrono::duration<double>(t_chk - t_sub0).count(),
            1e3*std::chrono::duration<double>(std::chrono::steady_clock::now() - t_chk).count());
  if (n_ok_total < n) return hipstr::api_fail(why);     // the loci before the refused one are in
  return 0;
}

// Collects the next `n_tickets` submissions in order into back-to-back buffers (ticket i's piece starts where ticket i-1's ends);
// *n_out / *n_reads = what was written.  The counterpart of hipstr_stream_submit_each for callers that want a shard's results at once.
int hipstr_stream_collect(hipstr_stream_t* s, int64_t n_tickets, double* aln_probs, int64_t cap_probs, int32_t* seeds, int64_t cap_seeds,
                          int64_t* n_out, int64_t* n_reads){
  if (!s || n_tickets < 0) return hipstr::api_fail("bad argument");
  int64_t po = 0, ro = 0;
  for (int64_t i = 0; i < n_tickets; i++){
    int64_t t, a, b;
    const int rs = hipstr_stream_next_size(s, &t, &a, &b);
    if (rs != 0) return rs == 2 ? hipstr::api_fail("fewer submissions outstanding than asked for") : 1;
    if (po + a > cap_probs || ro + b > cap_seeds) return hipstr::api_fail("output buffers are too small");
    if (hipstr_stream_next(s, NULL, aln_probs + po, cap_probs - po, seeds + ro, cap_seeds - ro) != 0) return 1;
    po += a; ro += b;
  }
  if (n_out) *n_out = po;
  if (n_reads) *n_reads = ro;
  return 0;
}

// One-shot call that refuses loci one by one (hipstr_hmm.h): the loci that pass check_locus are gathered into a batch of their own,
// processed by hipstr_hmm_process_reads, and their blocks copied back to where the caller's layout has them.
int hipstr_hmm_process_reads_each(const hipstr_batch_t* batch, double* aln_probs, int32_t* seeds, int32_t* locus_status){
  if (!batch || !aln_probs || !seeds || !locus_status) return hipstr::api_fail("null argument");
  { std::string bad; if (hipstr::validate_tables(batch, bad)) return hipstr::api_fail(bad); }      // (tables that contradict each other fail the call, not a locus)
  const int n = batch->n_loci;
  OwnedBatch ob;
  std::vector<int> good; std::vector<int64_t> out_off(n + 1, 0);
  std::string first_err;
  // the gathered loci run as one batch — or as several, where the longest read of one locus and the longest allele of another would not fit
  // the per-read STR kernels' LDS together (each fits alone: check_locus)
  auto run_gathered = [&]() -> int {
    if (good.empty()) return 0;
    std::vector<double> p((size_t)ob.n_out); std::vector<int32_t> sd((size_t)ob.read_off.back());
    for (size_t g = 0; g < good.size(); g++){              // entries the library leaves untouched keep the caller's values
      const int l = good[g]; const OwnedBatch::Ticket& t = ob.tickets[g];
      std::copy(aln_probs + out_off[l], aln_probs + out_off[l+1], p.begin() + t.out0);
      std::copy(seeds + batch->read_off[l], seeds + batch->read_off[l+1], sd.begin() + t.r0);
    }
    if (hipstr_hmm_process_reads(ob.finish(), p.data(), sd.data()) != 0) return 1;
    for (size_t g = 0; g < good.size(); g++){
      const int l = good[g]; const OwnedBatch::Ticket& t = ob.tickets[g];
      std::copy(p.begin() + t.out0, p.begin() + t.out1, aln_probs + out_off[l]);
      std::copy(sd.begin() + t.r0, sd.begin() + t.r1, seeds + batch->read_off[l]);
    }
    good.clear(); ob.reset();
    return 0;
  };
  for (int l = 0; l < n; l++) out_off[l+1] = out_off[l] + (int64_t)(batch->read_off[l+1] - batch->read_off[l])*(batch->hap_off[l+1] - batch->hap_off[l]);
  int opt = 0;
  for (int l = 0; l < n; l++){
    const int opt0 = opt; std::string why;
    int nopt = 0; for (int k = 0; k < 3; k++) nopt += std::max(0, batch->blk_nopts[3*l+k]);
    int cursor = opt0, dims[2] = {0, 1};
    const bool bad = hipstr::check_locus(batch, l, &cursor, why, NULL, dims) != 0;
    opt = opt0 + nopt;                                     // (check_locus stops advancing where it refuses)
    locus_status[l] = bad ? 1 : 0;
    if (bad){ if (first_err.empty()) first_err = "locus " + std::to_string(l) + ": " + why; continue; }
    if (!good.empty() && !ob.fits(dims[0], dims[1]) && run_gathered()) return 1;
    if (const char* w2 = ob.append_locus(batch, l, opt0, (int64_t)good.size())) return hipstr::api_fail(w2);
    ob.note(dims[0], dims[1]);
    good.push_back(l);
  }
  if (run_gathered()) return 1;
  if (!first_err.empty()) hipstr::api_fail(first_err);
  return 0;
}

int hipstr_stream_flush(hipstr_stream_t* s){
  if (!s) return hipstr::api_fail("null argument");
  if (getenv("HIPSTR_TIMING")) fprintf(stderr, "stream: flush called %.3f ms after open\n", 1e3*std::chrono::duration<double>(std::chrono::steady_clock::now() - s->t_open).count());
  std::lock_guard<std::mutex> g(s->m);
  flush_locked(s);
  return 0;
}

int hipstr_stream_next_size(hipstr_stream_t* s, int64_t* ticket, int64_t* n_out, int64_t* n_reads){
  if (!s) return hipstr::api_fail("null argument");
  std::lock_guard<std::mutex> g(s->m);
  if (s->next_deliver >= s->next_ticket) return 2;
  const std::pair<int64_t,int64_t>& z = s->sizes[(size_t)(s->next_deliver - s->sizes_base)];
  if (ticket) *ticket = s->next_deliver;
  if (n_out) *n_out = z.first;
  if (n_reads) *n_reads = z.second;
  return 0;
}

// Collects ONE submission, whichever: blocks until its batch has run.  This is what lets many loci be in flight at once — the
// reference's genotype() is a per-locus state machine (align, posteriors, tracebacks, new alleles, align again ...): one host thread
// per locus submits its round and waits for ITS ticket while the rounds of the other loci share the batches.
int hipstr_stream_take(hipstr_stream_t* s, int64_t ticket, double* aln_probs, int64_t cap_probs, int32_t* seeds, int64_t cap_seeds){
  hipstr::ApiTimer prof_t(hipstr::PB_STREAM_TAKE);
  if (!s) return hipstr::api_fail("null argument");
  CpuAdd cpu_t(s, &s->stats.cpu_collect_seconds);
  InFlight* f = NULL; size_t idx = 0;
  {
    std::unique_lock<std::mutex> g(s->m);
    if (ticket < 0 || ticket >= s->next_ticket) return hipstr::api_fail("no such ticket");
    if (ticket < s->next_deliver || s->taken_ahead.count(ticket)) return hipstr::api_fail("ticket was collected already");
    for (;;){
      for (InFlight* c : s->flying){
        const int64_t first = c->ob->tickets.front().id;
        if (ticket >= first && ticket < first + (int64_t)c->ob->tickets.size()){ f = c; idx = (size_t)(ticket - first); break; }
      }
      if (f) break;
      // not launched yet: tell the worker somebody is waiting (it sends the pending batch as soon as it is free), then wait for it
      if (s->closing) return hipstr::api_fail("stream is closing");
      s->waiting++; s->wait_tickets.insert(ticket);
      s->cv_work.notify_one();
      s->cv_done.wait(g);
      s->waiting--; s->wait_tickets.erase(s->wait_tickets.find(ticket));
      if (s->closing){ s->cv_done.notify_all(); return hipstr::api_fail("stream is closing"); }
    }
    if (f->taken[idx]) return hipstr::api_fail("ticket was collected already");
    f->taken[idx] = 2;               // claimed: a second collector of the same ticket is turned away while this one copies
    f->busy++;
  }
  const OwnedBatch::Ticket& t = f->ob->tickets[idx];
  int rc = 0; bool small = false;
  if (t.out1 - t.out0 > cap_probs || t.r1 - t.r0 > cap_seeds){ hipstr::api_fail("output buffers are too small for this ticket (hipstr_stream_next_size)"); rc = 3; small = true; }
  else {
    {
      std::lock_guard<std::mutex> lg(f->land_m);
      if (!f->landed && !f->failed){
        const auto t0 = std::chrono::steady_clock::now();
        if (hipstr::results_wait(f->dev) != 0){ f->failed = true; f->err = hipstr_last_error(); }
        f->landed = true;
        if (getenv("HIPSTR_TIMING"))
          fprintf(stderr, "stream: batch of %zu tickets landed %.3f ms after open (collector waited from %.3f)\n", f->ob->tickets.size(),
                  1e3*std::chrono::duration<double>(std::chrono::steady_clock::now() - s->t_open).count(), 1e3*std::chrono::duration<double>(t0 - s->t_open).count());
        std::lock_guard<std::mutex> g(s->m);
        s->stats.wait_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      }
    }
    if (f->failed) rc = hipstr::api_fail("batch failed: " + f->err);
    else if ((t.out1 - t.out0) > ((int64_t)1 << 20) && t.l1 - t.l0 >= 8){       // a big ticket: loci are independent, share them among the host threads
      const hipstr_dev_batch_t* dev = f->dev;
      const int nl = t.l1 - t.l0, parts = std::min(nl, hipstr::host_threads()*2);
      const OwnedBatch* ob = f->ob;
      hipstr::parallel_for(parts, hipstr::host_threads(), [&](int p){
        const int a = t.l0 + (int)((int64_t)nl*p/parts), b = t.l0 + (int)((int64_t)nl*(p+1)/parts);
        if (b <= a) return;
        int64_t out_a = t.out0;       // output offset of locus a relative to the ticket: sum of P*A of the loci before it
        for (int l = t.l0; l < a; l++) out_a += (int64_t)(ob->read_off[l+1] - ob->read_off[l])*(ob->hap_off[l+1] - ob->hap_off[l]);
        hipstr::scatter_loci(dev, a, b, aln_probs + (out_a - t.out0), seeds + (ob->read_off[a] - t.r0));
      });
    } else hipstr::scatter_loci(f->dev, t.l0, t.l1, aln_probs, seeds);
  }
  bool retire = false;
  {
    std::lock_guard<std::mutex> g(s->m);
    f->busy--;
    if (small) f->taken[idx] = 0;        // "buffers too small" (return code 3) leaves the ticket for another try; a failed batch consumes its tickets
    else {
      f->taken[idx] = 1; f->n_taken++;
      s->stats.tickets++;
      if (ticket == s->next_deliver){
        s->next_deliver++;
        while (!s->taken_ahead.empty() && *s->taken_ahead.begin() == s->next_deliver){ s->taken_ahead.erase(s->taken_ahead.begin()); s->next_deliver++; }
      } else s->taken_ahead.insert(ticket);
      // drop delivered entries of the size table now and then
      if (s->next_deliver - s->sizes_base > 4096){ s->sizes.erase(s->sizes.begin(), s->sizes.begin() + (size_t)(s->next_deliver - s->sizes_base)); s->sizes_base = s->next_deliver; }
    }
    if (f->n_taken == f->ob->tickets.size() && f->busy == 0){
      for (std::deque<InFlight*>::iterator it = s->flying.begin(); it != s->flying.end(); ++it) if (*it == f){ s->flying.erase(it); break; }
      retire = true;
    }
  }
  if (retire){
    if (f->dev) hipstr::free_landed(f->dev, f->landed && !f->failed);
    retire_batch(s, f->ob); delete f;
    s->cv_work.notify_one();           // a slot is free
  }
  return rc;
}

int hipstr_stream_next(hipstr_stream_t* s, int64_t* ticket, double* aln_probs, int64_t cap_probs, int32_t* seeds, int64_t cap_seeds){
  if (!s) return hipstr::api_fail("null argument");
  int64_t t;
  {
    std::lock_guard<std::mutex> g(s->m);
    if (s->next_deliver >= s->next_ticket) return 2;                     // nothing outstanding
    t = s->next_deliver;
  }
  if (ticket) *ticket = t;
  return hipstr_stream_take(s, t, aln_probs, cap_probs, seeds, cap_seeds);
}

int hipstr_stream_stats(hipstr_stream_t* s, hipstr_stream_stats_t* out){
  if (!s || !out) return hipstr::api_fail("null argument");
  std::lock_guard<std::mutex> g(s->m);
  *out = s->stats;
  out->open_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - s->t_open).count();
  return 0;
}

int hipstr_stream_close(hipstr_stream_t* s){
  if (!s) return 0;
  {
    std::unique_lock<std::mutex> g(s->m);
    s->closing = true;
    for (OwnedBatch* ob : s->ready) delete ob;          // undelivered work is dropped
    s->ready.clear();
    delete s->pending; s->pending = NULL;
    // collectors blocked on a ticket that will never be launched return with an error before the stream goes away
    s->cv_done.notify_all();
    s->cv_done.wait(g, [&]{ return s->waiting == 0; });
  }
  s->cv_work.notify_all();
  for (std::thread& t : s->workers) if (t.joinable()) t.join();
  hipstr::api_bind(s->ctx);
  for (InFlight* f : s->flying){ if (f->dev) hipstr::free_landed(f->dev, false); delete f->ob; delete f; }
  hipStreamSynchronize(s->copy_stream); hipStreamSynchronize(s->d2h_stream);
  for (int i = 0; i < s->n_compute; i++){ hipStreamSynchronize(s->compute[i]); hipStreamDestroy(s->compute[i]); }
  for (OwnedBatch* ob : s->spare) delete ob;
  s->spare.clear();
  hipStreamDestroy(s->copy_stream); hipStreamDestroy(s->d2h_stream);
  delete s;
  return 0;
}


// ---------------------------------------------------------------------------------------------------------------- several GPUs, one process
// The region list cut into contiguous blocks, block i on device i mod n (one hipstr_stream_t per device), results handed back in GLOBAL
// submission order: the in-process form of SURVEY §8(e) — loci are independent, so nothing crosses between devices but the order.
}  // extern "C"

struct hipstr_multi {
  std::vector<hipstr_stream_t*> streams;
  int64_t block_work = (int64_t)16 << 20;
  std::mutex m;
  std::deque<int> owner;          // device slot of every ticket not yet delivered, in submission order
  int cur = 0;                    // slot receiving the current block
  int64_t cur_work = 0, next_ticket = 0;
  std::vector<double> dealt;      // per device slot: estimated work of everything submitted to it so far (hipstr::locus_cost)
};

extern "C" {

hipstr_multi_t* hipstr_multi_open(int32_t n_devices, const int32_t* devices, int64_t block_alignments, const hipstr_stream_opts_t* per_stream){
  if (n_devices < 1){ hipstr::api_fail("at least one device"); return NULL; }
  hipstr_multi* mm = new hipstr_multi();
  if (block_alignments > 0) mm->block_work = block_alignments;
  for (int i = 0; i < n_devices; i++){
    hipstr_stream_opts_t o; memset(&o, 0, sizeof o);
    if (per_stream) o = *per_stream;
    o.device = devices ? devices[i] : i;
    hipstr_stream_t* s = hipstr_stream_open(&o);
    if (!s){ for (hipstr_stream_t* t : mm->streams) hipstr_stream_close(t); delete mm; return NULL; }
    mm->streams.push_back(s);
  }
  mm->dealt.assign(mm->streams.size(), 0.0);
  return mm;
}

int64_t hipstr_multi_submit(hipstr_multi_t* mm, const hipstr_batch_t* loci){
  if (!mm || !loci){ hipstr::api_fail("null argument"); return -1; }
  { std::string bad; if (hipstr::validate_tables(loci, bad)){ hipstr::api_fail(bad); return -1; } }
  std::lock_guard<std::mutex> g(mm->m);
  int64_t work = 0; double cost = 0.0;
  for (int l = 0, opt0 = 0; l < loci->n_loci; l++){
    work += (int64_t)(loci->read_off[l+1] - loci->read_off[l])*(loci->hap_off[l+1] - loci->hap_off[l]);
    int nopt = 0; bool ok = true;
    for (int k = 0; k < 3; k++){ ok &= loci->blk_nopts[3*l+k] >= 1; nopt += std::max(0, loci->blk_nopts[3*l+k]); }
    if (ok && loci->read_off[l+1] >= loci->read_off[l]) cost += hipstr::locus_cost(loci, l, opt0);       // (a malformed locus is refused by the stream below)
    opt0 += nopt;
  }
  if (mm->cur_work > 0 && mm->cur_work + work > mm->block_work){        // the block is full: send it ...
    hipstr_stream_flush(mm->streams[mm->cur]);
    // ... and the next block goes to the device that has been dealt the least WORK so far (locus_cost: reads x alleles x read length x
    // [flank rows + STR block by its interruptions]) — not round-robin by pair counts: interrupted-repeat loci cost 3-8x a
    // periodic one's pairs, and the slowest device sets the pace.  Ties (and the first round) go to the lowest slot.
    int best = 0;
    for (int i = 1; i < (int)mm->streams.size(); i++) if (mm->dealt[i] < mm->dealt[best]) best = i;
    mm->cur = best; mm->cur_work = 0;
  }
  if (hipstr_stream_submit(mm->streams[mm->cur], loci) < 0) return -1;
  mm->cur_work += work; mm->dealt[mm->cur] += cost;
  mm->owner.push_back(mm->cur);
  return mm->next_ticket++;
}

int hipstr_multi_dealt(hipstr_multi_t* mm, double* cost_per_device, int32_t cap){
  if (!mm) return hipstr::api_fail("null argument");
  std::lock_guard<std::mutex> g(mm->m);
  for (int i = 0; i < std::min<int>(cap, (int)mm->dealt.size()); i++) cost_per_device[i] = mm->dealt[i];
  return (int)mm->dealt.size();
}

int hipstr_locus_costs(const hipstr_batch_t* batch, double* costs){
  if (!batch || !costs) return hipstr::api_fail("null argument");
  { std::string bad; if (hipstr::validate_tables(batch, bad)) return hipstr::api_fail(bad); }
  for (int l = 0, opt0 = 0; l < batch->n_loci; l++){
    int nopt = 0;
    for (int k = 0; k < 3; k++){ if (batch->blk_nopts[3*l+k] < 1) return hipstr::api_fail("haplotype block without options"); nopt += batch->blk_nopts[3*l+k]; }
    if (batch->read_off[l+1] < batch->read_off[l]) return hipstr::api_fail("read_off must not decrease");
    costs[l] = hipstr::locus_cost(batch, l, opt0);
    opt0 += nopt;
  }
  return 0;
}

int hipstr_multi_flush(hipstr_multi_t* mm){
  if (!mm) return hipstr::api_fail("null argument");
  std::lock_guard<std::mutex> g(mm->m);
  for (hipstr_stream_t* s : mm->streams) hipstr_stream_flush(s);
  return 0;
}

int hipstr_multi_next_size(hipstr_multi_t* mm, int64_t* ticket, int64_t* n_out, int64_t* n_reads){
  if (!mm) return hipstr::api_fail("null argument");
  int slot; int64_t t;
  { std::lock_guard<std::mutex> g(mm->m); if (mm->owner.empty()) return 2; slot = mm->owner.front(); t = mm->next_ticket - (int64_t)mm->owner.size(); }
  if (ticket) *ticket = t;
  return hipstr_stream_next_size(mm->streams[slot], NULL, n_out, n_reads);
}

int hipstr_multi_next(hipstr_multi_t* mm, int64_t* ticket, double* aln_probs, int64_t cap_probs, int32_t* seeds, int64_t cap_seeds){
  if (!mm) return hipstr::api_fail("null argument");
  int slot; int64_t t;
  { std::lock_guard<std::mutex> g(mm->m); if (mm->owner.empty()) return 2; slot = mm->owner.front(); t = mm->next_ticket - (int64_t)mm->owner.size(); }
  const int rc = hipstr_stream_next(mm->streams[slot], NULL, aln_probs, cap_probs, seeds, cap_seeds);     // that device's next = the globally next
  if (rc == 2) return hipstr::api_fail("internal error: device stream has nothing outstanding");
  if (ticket) *ticket = t;
  if (rc == 3) return rc;               // buffers too small: the device stream kept the ticket, so does the owner queue (retry with larger ones)
  { std::lock_guard<std::mutex> g(mm->m); if (!mm->owner.empty()) mm->owner.pop_front(); }
  return rc;
}

int hipstr_multi_close(hipstr_multi_t* mm){
  if (!mm) return 0;
  for (hipstr_stream_t* s : mm->streams) hipstr_stream_close(s);
  delete mm;
  return 0;
}

}  // extern "C"
