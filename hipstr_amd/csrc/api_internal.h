// api_internal.h — state of api.hip shared with the other translation units of libhipstr_hmm.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <string.h>

namespace hipstr {

// One context per device the process uses: constant tables, the library's stream, and two block caches (device memory and pinned
// host memory).  hipstr_hmm_init(d) makes device d the calling thread's current context; every object created afterwards
// (device batches, posterior runs) belongs to that context and may be used from any thread.
struct Ctx;
Ctx* api_current_ctx();                       // the calling thread's context; initialises device 0 on first use; NULL + last error on failure
int  api_bind(Ctx* ctx);                      // hipSetDevice(ctx's device) for the calling thread

struct ApiTables {                 // device copies of HostTables, owned by the context
  const double *int_log, *qual_correct, *qual_error, *m2m, *m2i;
  hipStream_t stream;
  Ctx* ctx;
};
int api_device_tables(ApiTables* t);          // tables of the calling thread's current context; 1 + hipstr_last_error() on failure
int api_fail(const std::string& message);     // records hipstr_last_error(); returns 1

// Cached blocks: a freed block goes back to its context's free list instead of hipFree / hipHostFree (both synchronise the
// device and cost 0.1-10 ms); a request is served from the list when a block of at most 1.25x the size is there.
// Waiting for a stream from a host thread.  hipStreamSynchronize / hipEventSynchronize spin at 100 % of a core on this stack (tools/wait_probe.hip),
// which is the fastest way to learn that a 0.2 ms call is done as long as every waiting thread has a core of its own — and the worst one
// when the caller runs more host threads than cores to keep the device busy (the reference's genotype() over the adapter: 16 threads on 16
// cores 755 loci/s, 64 threads 291): the waiters keep the cores from the threads that have host work.  wait_stream polls hipStreamQuery and
// yields the core between polls (sched_yield returns at once when nothing else is runnable: the single-thread latency stays what it was),
// sleeping 50 us per poll once a wait has lasted 2 ms.  HIPSTR_WAIT=spin: the driver's own wait; =sleep: sleep from the first poll.
hipError_t wait_stream(hipStream_t st);

void* dev_alloc(Ctx* ctx, size_t bytes);      // NULL + last error on failure
void  dev_free(Ctx* ctx, void* p);
void* pin_alloc(Ctx* ctx, size_t bytes);
void  pin_free(Ctx* ctx, void* p);


// Host arrays that travel together: packed into ONE pinned block and sent with ONE asynchronous copy (a synchronous hipMemcpy from
// pageable memory costs 10-20 us each; the traceback and Needleman-Wunsch calls of a locus had ten of them).  add() returns a piece's offset;
// after send() the piece sits at dev + offset.  Both blocks go back to the context's caches when the arena dies; the caches serve other
// host threads, so the arena first waits for the stream its copy went to (an error path may leave while the copy is in flight; on the
// normal path the results were fetched already and the wait costs a few microseconds).
struct HostArena {
  struct Piece { const void* src; size_t bytes, off; };
  std::vector<Piece> pieces; size_t total = 0;
  Ctx* ctx = NULL; char* pin = NULL; char* dev = NULL;
  hipStream_t sent_on = NULL; bool sent = false;
  ~HostArena(){ if (ctx){ if (sent) hipStreamSynchronize(sent_on); if (pin) pin_free(ctx, pin); if (dev) dev_free(ctx, dev); } }
  size_t add(const void* src, size_t bytes){
    const size_t off = total; pieces.push_back(Piece{src, bytes, off}); total = (total + (bytes ? bytes : 1) + 255) & ~(size_t)255; return off;
  }
  int reserve(Ctx* c){        // blocks only (the caller still has pointers into `dev` to fill in before the pieces are packed)
    ctx = c; pin = (char*)pin_alloc(ctx, total ? total : 256); dev = (char*)dev_alloc(ctx, total ? total : 256);
    return (pin && dev) ? 0 : 1;
  }
  int send(hipStream_t st){   // pack and copy
    for (const Piece& pc : pieces) if (pc.bytes && pc.src) memcpy(pin + pc.off, pc.src, pc.bytes);
    sent_on = st; sent = true;
    return (total && hipMemcpyAsync(dev, pin, total, hipMemcpyHostToDevice, st) != hipSuccess) ? api_fail("hipMemcpyAsync (host to device) failed") : 0;
  }
  template <typename T> T* at(size_t off) const { return (T*)(dev + off); }
};

// ---- where the host time of the entry points goes (hipstr_debug_api_profile, include/hipstr_hmm.h): wall-clock seconds and calls per
// bucket, summed over all threads while enabled.  Buckets nest: the indented ones are parts of the entry point above them.
enum ApiBucket { PB_PROCESS_READS, PB_PR_PREPARE, PB_PR_STAGE, PB_PR_UPLOAD_REST, PB_PR_LAUNCH, PB_PR_FETCH, PB_PR_FREE,
                 PB_TRACE, PB_TRACE_REPLAY, PB_POST_RUN, PB_POST_EXTRACT, PB_EM_TRAIN, PB_NW_ALIGN, PB_STREAM_SUBMIT, PB_STREAM_TAKE, PB_SEED_BASES,
                 PB_UP_BLOCKS, PB_UP_MEMCPY, PB_UP_EXPAND, PB_UP_EVENTS, PB_COUNT };        // (the last four: parts of "blocks + H2D enqueue")
bool api_profile_on();
void api_profile_add(int bucket, double seconds, int calls = 1);
struct ApiTimer {                  // adds its lifetime to a bucket
  int bucket; bool on; double t0;
  static double now();
  explicit ApiTimer(int b) : bucket(b), on(api_profile_on()), t0(on ? now() : 0.0) {}
  ~ApiTimer(){ if (on) api_profile_add(bucket, now() - t0); }
};

// ---- the pieces of hipstr_hmm_process_reads, split for pipelined use (stream.hip)
}  // namespace hipstr
struct hipstr_batch; struct hipstr_dev_batch;
namespace hipstr {
// reads_pinned: batch->bases / quals lie in pinned host memory that outlives the copy: they are sent from there, not through the staging block
hipstr_dev_batch* upload_on(Ctx* ctx, const hipstr_batch* batch, const int32_t* seed_base, hipStream_t copy_stream, hipStream_t compute_stream, bool reads_pinned = false);
int  fetch_begin(hipstr_dev_batch* dev, hipStream_t compute_stream, hipStream_t copy_stream);
int  fetch_poll(hipstr_dev_batch* dev);         // queues the copy back of a batch whose kernels are done (0 queued or nothing pending, 1 still running, -1 error)
int  results_wait(hipstr_dev_batch* dev);
double batch_prepare_seconds(const hipstr_dev_batch* dev);      // wall time of the batch's prepare_batch
void scatter_loci(const hipstr_dev_batch* dev, int l0, int l1, double* aln_probs, int32_t* seeds);   // outputs based at locus l0
void free_landed(hipstr_dev_batch* dev, bool landed);
hipStream_t ctx_stream(Ctx* c);
int ctx_device(Ctx* c);

}  // namespace hipstr
