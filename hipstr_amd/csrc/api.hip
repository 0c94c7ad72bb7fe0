// api.hip — the C-ABI of include/hipstr_hmm.h on top of the HIP kernels.
//
// Host responsibilities only: prepare (prep.cpp), move bytes, launch, time with HIP events on
// the launch stream, and re-impose the reference's "leave untouched what was not realigned"
// output contract.  There is no CPU compute path here: without a gfx950 device every entry
// point fails and hipstr_last_error() says so.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <sched.h>
#include <atomic>

#include <cfloat>
#include <chrono>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/hipstr_hmm.h"
#include "../../include/hipstr_hmm_debug.h"
#include "layout.h"
#include "post_layout.h"
#include "prep.h"
#include "api_internal.h"

extern "C" __global__ void hs_str_kernel(const hs_dev_t* dp, int active_begin, int only_long);
extern "C" __global__ void hs_str_kernel_generic(const hs_dev_t* dp, int active_begin, int pw_grouped);
extern "C" __global__ void hs_combine_kernel(const hs_dev_t* dp, int active_begin);
extern "C" int hs_combine_waves();
extern "C" __global__ void hs_posterior_kernel(const hs_post_dev_t* dp);
extern "C" __global__ void hs_posterior_accumulate_kernel(const hs_post_dev_t* dp);
extern "C" __global__ void hs_posterior_finish_kernel(const hs_post_dev_t* dp);
extern "C" __global__ void hs_genotype_kernel(const hs_gt_dev_t* dp);
extern "C" __global__ void hs_cr_math_kernel(int which, const double* x, double* y, int64_t n);
extern "C" size_t hs_str_lds_bytes(int lds_len, int max_B);
extern "C" __global__ void hs_str_group_kernel(const hs_dev_t* dp, int item_begin, int short_only);
extern "C" __global__ void hs_str_group_kernel_pw(const hs_dev_t* dp, int item_begin);
extern "C" __global__ void hs_str_group_kernel_rp(const hs_dev_t* dp, int item_begin);
extern "C" __global__ void hs_str_group_kernel_p(const hs_dev_t* dp, int item_begin);
extern "C" __global__ void hs_nd_kernel(const hs_dev_t* dp, int active_begin);
extern "C" __global__ void hs_expand_stropts_kernel(const hs_dev_t* dp);
extern "C" __global__ void hs_expand_recs_kernel(const hs_dev_t* dp);
extern "C" size_t hs_str_group_lds_bytes(int max_B, int nd_cap, int with_ilog);
extern "C" size_t hs_str_group_p_lds_bytes();
extern "C" void hs_launch_lead2(unsigned n_active, unsigned n_wavefronts, hipStream_t st, const hs_dev_t* dp, int active_begin, int item_begin, int item_end, int chunk, int max_cols, int n_clear);
extern "C" void hs_launch_trail(unsigned n_wavefronts, hipStream_t st, const hs_dev_t* dp, int item_begin, int item_end, int chunk, int max_cols, int max_rows);

namespace {

thread_local std::string g_err;
int fail(const std::string& m){ g_err = m; return 1; }

#define HS_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess){ \
  g_err = std::string(#call) + ": " + hipGetErrorString(e_); return 1; } } while (0)
#define HS_HIP_NULL(call) do { hipError_t e_ = (call); if (e_ != hipSuccess){ \
  g_err = std::string(#call) + ": " + hipGetErrorString(e_); return NULL; } } while (0)

// Free lists of device / pinned blocks (see api_internal.h).  Sizes are rounded up to 256 B below 1 MiB and to 1/16 of the
// next power of two above, so that batches of similar size hit the same classes.
std::atomic<int64_t> g_driver_allocs(0);
// Blocks are carved from large chunks taken from the driver (256 MiB at first, doubling up to 16 GiB of device / 2 GiB of pinned memory)
// and never go back to it one by one: a hipMalloc / hipHostMalloc next to running kernels stalls for 0.1-0.3 s (seen in the middle of a
// stream: profiles/r04_notes.md), a hipFree synchronises the device.  A freed block goes to a free list by size; a request takes the
// smallest free block that holds it if that is at most twice as large, else a fresh piece of the current chunk.  No coalescing: the
// requests of a stream repeat, so the free list saturates after a few batches and the driver is not called again.  `cap` bounds the IDLE
// bytes (free blocks): a put() that leaves more gives the chunks nobody uses back to the driver; when the driver refuses a new chunk,
// get() falls back to any free block that is large enough, then trims idle chunks and tries once more.
struct BlockCache {
  bool pinned = false;
  std::mutex m;
  std::multimap<size_t, void*> free_;
  std::unordered_map<void*, size_t> size_;
  struct Chunk { char* base; size_t size, used; int live = 0; bool small = false; };       // live: blocks of the chunk that are out
  // Requests of up to 1 MiB are carved from chunks of their own (32 MiB each): the handful of small, fixed-size blocks every batch takes
  // would otherwise be re-used out of every old chunk and keep all of them from ever being trimmed.
  static constexpr size_t SMALL_MAX = (size_t)1 << 20, SMALL_CHUNK = (size_t)32 << 20;
  std::vector<Chunk> chunks;
  size_t next_chunk = (size_t)256 << 20;
  size_t cached = 0, cap = 0, in_use = 0;
  int64_t n_fallback = 0, n_trim_retry = 0;       // requests served by a free block beyond the 2x window / by a chunk taken after trimming (driver refusals survived)
  static size_t round_up(size_t n){
    if (n < 256) return 256;
    if (n <= ((size_t)1 << 20)) return (n + 255) & ~(size_t)255;
    size_t p2 = (size_t)1 << 20; while (p2 < n) p2 <<= 1;
    const size_t step = n > ((size_t)64 << 20) ? p2 >> 3 : p2 >> 4;
    return (n + step - 1) / step * step;
  }
  void* driver_alloc(size_t bytes){
    void* p = NULL;
    g_driver_allocs++;
    if (getenv("HIPSTR_TIMING")) fprintf(stderr, "block cache: %s chunk of %zu bytes from the driver\n", pinned ? "pinned" : "device", bytes);
    // tests: a driver that runs out at HIPSTR_DEBUG_DRIVER_LIMIT_MIB of device chunks (the recovery paths of get() without filling 288 GB)
    static const size_t dbg_limit = getenv("HIPSTR_DEBUG_DRIVER_LIMIT_MIB") ? (size_t)atol(getenv("HIPSTR_DEBUG_DRIVER_LIMIT_MIB")) << 20 : 0;
    if (dbg_limit && !pinned){
      size_t held = 0; for (const Chunk& c : chunks) held += c.size;
      if (held + bytes > dbg_limit){ g_err = "hipMalloc: out of memory (HIPSTR_DEBUG_DRIVER_LIMIT_MIB)"; return NULL; }
    }
    const hipError_t e = pinned ? hipHostMalloc(&p, bytes, hipHostMallocDefault) : hipMalloc(&p, bytes);
    if (e != hipSuccess){ g_err = std::string(pinned ? "hipHostMalloc: " : "hipMalloc: ") + hipGetErrorString(e); return NULL; }
    return p;
  }
  // (lock held) a free block of at least `want` bytes, the smallest one; `limit`: largest size accepted (0 = any)
  void* take_free(size_t want, size_t limit){
    auto it = free_.lower_bound(want);
    if (it == free_.end() || (limit && it->first > limit)) return NULL;
    void* p = it->second; cached -= it->first; free_.erase(it); in_use++;
    if (Chunk* c = chunk_of(p)) c->live++;
    return p;
  }
  void* get(size_t bytes){
    size_t want = round_up(bytes);
    std::unique_lock<std::mutex> g(m);
    if (void* p = take_free(want, 2*want)) return p;
    const size_t exact = want;
    if (want > ((size_t)16 << 20)) want = round_up(want + want/8);        // a new large block comes with headroom for its successors
    const bool small = exact <= SMALL_MAX;
    int ci = -1;                                         // the newest chunk of this class with room
    for (int i = (int)chunks.size() - 1; i >= 0; i--) if (chunks[i].small == small){ if (chunks[i].used + want <= chunks[i].size) ci = i; break; }
    if (ci < 0){
      const size_t max_chunk = pinned ? (size_t)2 << 30 : (size_t)16 << 30;
      size_t sz = small ? SMALL_CHUNK : std::max(next_chunk, want);
      if (!small) next_chunk = std::min(max_chunk, next_chunk*2);
      char* base = (char*)driver_alloc(sz);
      if (!base && sz > want) base = (char*)driver_alloc(sz = want);       // the device is nearly full: just what is needed
      if (!base){
        // the driver has no more: a larger free block, whatever its size, before giving up ...
        if (void* p = take_free(exact, 0)){ g_err.clear(); n_fallback++; return p; }
        // ... then the chunks without a block in use go back to the driver and the request is tried once more
        g.unlock();
        const size_t freed = trim();
        g.lock();
        if (void* p = take_free(exact, 0)){ g_err.clear(); n_fallback++; return p; }    // (another thread may have returned one meanwhile)
        if (freed){ g_err.clear(); base = (char*)driver_alloc(sz = exact); if (base) n_trim_retry++; }
        if (!base) return NULL;                                            // (the driver's message is in g_err)
        want = exact;
      }
      Chunk nc{base, sz, 0, 0}; nc.small = small;
      chunks.push_back(nc);
      ci = (int)chunks.size() - 1;
    }
    Chunk& c = chunks[ci];
    void* p = c.base + c.used;
    c.used += (want + 255) & ~(size_t)255;
    size_[p] = want;
    in_use++; c.live++;
    return p;
  }
  Chunk* chunk_of(const void* p){ for (Chunk& c : chunks) if ((const char*)p >= c.base && (const char*)p < c.base + c.size) return &c; return NULL; }
  void put(void* p){
    if (!p) return;
    std::unique_lock<std::mutex> g(m);
    auto it = size_.find(p);
    if (it == size_.end()) return;
    bool over;
    {
      free_.insert(std::make_pair(it->second, p)); cached += it->second;
      if (in_use > 0) in_use--;
      if (Chunk* c = chunk_of(p)) if (c->live > 0) c->live--;
      over = cap && cached > cap;
      if (over){                                   // (nothing to give back while every chunk has a block out: no second pass over the lists)
        over = false;
        for (const Chunk& c : chunks) if (c.live == 0){ over = true; break; }
      }
    }
    // more idle bytes than the cap allows (HIPSTR_DEV_CACHE_GIB / HIPSTR_PIN_CACHE_GIB): chunks nobody uses go back to the driver — down to
    // a low-water mark of 0.8 x cap, not all of them: a stream that sits near the cap would otherwise alternate a hipFree here (it
    // synchronises the device) with a hipMalloc at its next get(), the stall the cache exists to avoid
    if (over){ g.unlock(); trim(cap - cap/5); }
  }
  // chunks none of whose blocks is out go back to the driver (hipstr_hmm_trim: after a stream of large batches a process may sit on tens
  // of gigabytes it no longer needs — another process on the device, a child of this one, then fails its kernel launches with "out of
  // memory"); returns the bytes released
  // keep_idle: stop once the idle bytes are down to this (0: every chunk without a block out, hipstr_hmm_trim's meaning)
  size_t trim(size_t keep_idle = 0){
    std::vector<Chunk> gone; size_t bytes = 0;
    {
      std::lock_guard<std::mutex> g(m);
      for (size_t i = 0; i < chunks.size(); ){
        if (keep_idle && cached <= keep_idle) break;
        if (chunks[i].live != 0){ i++; continue; }
        const Chunk c = chunks[i];
        for (auto it = free_.begin(); it != free_.end(); ){
          if ((char*)it->second >= c.base && (char*)it->second < c.base + c.size){ cached -= it->first; size_.erase(it->second); it = free_.erase(it); } else ++it;
        }
        gone.push_back(c); bytes += c.size;
        chunks.erase(chunks.begin() + (long)i);
      }
      if (chunks.empty()) next_chunk = (size_t)256 << 20;
    }
    for (Chunk& c : gone){ if (pinned) hipHostFree(c.base); else hipFree(c.base); }
    return bytes;
  }
  // everything back to the driver — only when no block is out (hipstr_hmm_shutdown)
  void release(){
    std::vector<Chunk> v;
    { std::lock_guard<std::mutex> g(m); if (in_use > 0) return; v.swap(chunks); free_.clear(); size_.clear(); cached = 0; next_chunk = (size_t)256 << 20; }
    for (Chunk& c : v){ if (pinned) hipHostFree(c.base); else hipFree(c.base); }
  }
};

}  // namespace

namespace hipstr {
struct Ctx {
  int device = -1;
  double *int_log = NULL, *qc = NULL, *qe = NULL, *m2m = NULL, *m2i = NULL;
  hipStream_t stream = NULL;
  BlockCache dev_cache, pin_cache;
  // events are pooled like the blocks: a device batch needs five (hipEventCreate / Destroy cost ~10 us each: a tenth of a one-locus call)
  std::mutex ev_m;
  std::vector<hipEvent_t> ev_timing, ev_plain;
  hipEvent_t get_event(bool timing){
    { std::lock_guard<std::mutex> g(ev_m);
      std::vector<hipEvent_t>& v = timing ? ev_timing : ev_plain;
      if (!v.empty()){ hipEvent_t e = v.back(); v.pop_back(); return e; } }
    hipEvent_t e = NULL;
    if ((timing ? hipEventCreate(&e) : hipEventCreateWithFlags(&e, hipEventDisableTiming)) != hipSuccess) return NULL;
    return e;
  }
  void put_event(hipEvent_t e, bool timing){ if (!e) return; std::lock_guard<std::mutex> g(ev_m); (timing ? ev_timing : ev_plain).push_back(e); }
};
}  // namespace hipstr

namespace {
using hipstr::Ctx;
std::mutex g_ctx_mutex;
std::map<int, Ctx*> g_ctxs;                 // by device ordinal; contexts live until hipstr_hmm_shutdown
thread_local Ctx* t_ctx = NULL;             // what hipstr_hmm_init selected on this thread

int upload_table(const std::vector<double>& v, double** out){
  *out = NULL;
  HS_HIP(hipMalloc((void**)out, v.size()*sizeof(double)));
  HS_HIP(hipMemcpy(*out, v.data(), v.size()*sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

Ctx* ctx_for_device(int device_ordinal){
  std::lock_guard<std::mutex> lock(g_ctx_mutex);          // concurrent first calls must not both build the tables
  auto it = g_ctxs.find(device_ordinal);
  if (it != g_ctxs.end()){ if (hipSetDevice(device_ordinal) != hipSuccess){ g_err = "hipSetDevice failed"; return NULL; } return it->second; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0){
    g_err = "no HIP device available: this library has no CPU path (build/run on an MI355X)"; return NULL; }
  if (device_ordinal < 0 || device_ordinal >= ndev){ g_err = "device ordinal out of range"; return NULL; }
  HS_HIP_NULL(hipSetDevice(device_ordinal));
  hipDeviceProp_t prop;
  HS_HIP_NULL(hipGetDeviceProperties(&prop, device_ordinal));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0){
    g_err = std::string("kernels are built for gfx950 only; device is ") + prop.gcnArchName; return NULL; }
  const hipstr::HostTables& T = hipstr::host_tables();
  std::vector<double> m2m(T.m2m, T.m2m+16), m2i(T.m2i, T.m2i+16);
  Ctx* c = new Ctx();
  c->device = device_ordinal;
  c->pin_cache.pinned = true;
  // free blocks the caches may hold: every batch in flight returns its workspaces at once when a stream drains (eight 2 Mi-pair batches of
  // 500-read loci: 60 GB), and a block given back to the driver is a hipFree (which synchronises the device) plus a hipMalloc later —
  // so: 70 % of the device's memory (288 GB on an MI355X), 24 GiB of pinned host memory as the most IDLE bytes a cache keeps (BlockCache::put
  // trims the chunks nobody uses beyond that; BlockCache::get reuses any large-enough free block and trims when the driver refuses)
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) total_b = (size_t)96 << 30;
  c->dev_cache.cap = getenv("HIPSTR_DEV_CACHE_GIB") ? (size_t)(atof(getenv("HIPSTR_DEV_CACHE_GIB"))*1073741824.0) : (size_t)(0.7*(double)total_b);
  c->pin_cache.cap = (size_t)((getenv("HIPSTR_PIN_CACHE_GIB") ? atof(getenv("HIPSTR_PIN_CACHE_GIB")) : 24.0)*1073741824.0);
  if (upload_table(T.int_log, &c->int_log) || upload_table(T.qual_correct, &c->qc) || upload_table(T.qual_error, &c->qe) ||
      upload_table(m2m, &c->m2m) || upload_table(m2i, &c->m2i) ||
      hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess){
    if (g_err.empty()) g_err = "hipStreamCreate failed";
    delete c; return NULL;
  }
  g_ctxs[device_ordinal] = c;
  return c;
}

Ctx* current_ctx(){
  if (t_ctx) return t_ctx;
  {     // a thread that never called hipstr_hmm_init uses the process' first context, or device 0
    std::lock_guard<std::mutex> lock(g_ctx_mutex);
    if (!g_ctxs.empty()) t_ctx = g_ctxs.begin()->second;
  }
  if (!t_ctx) t_ctx = ctx_for_device(0);
  return t_ctx;
}

int bind(Ctx* c){ HS_HIP(hipSetDevice(c->device)); return 0; }
}  // namespace

hipError_t hipstr::wait_stream(hipStream_t st){
  static const int mode = []{ const char* e = getenv("HIPSTR_WAIT"); return !e ? 1 : !strcmp(e, "spin") ? 0 : !strcmp(e, "sleep") ? 2 : 1; }();
  if (mode == 0) return hipStreamSynchronize(st);
  const auto t0 = std::chrono::steady_clock::now();
  bool slow = (mode == 2);
  // (round 6: an event recorded behind the stream's work and hipEventQuery — a hipStreamQuery that finds the stream busy leaves work for the
  //  HSA runtime's event thread: 0.45 s of its CPU per 3 s of polling against 0.03 with an event, tools/r06_rt_probe.py; sixteen host threads
  //  of one-shot calls share that one thread)
  // (the event comes from this thread's context: only when that is the device the entry point bound — the stream's)
  Ctx* c = t_ctx;
  { int cur = -1; if (!c || hipGetDevice(&cur) != hipSuccess || cur != c->device) c = NULL; }
  hipEvent_t ev = c ? c->get_event(false) : NULL;
  if (ev && hipEventRecord(ev, st) != hipSuccess){ c->put_event(ev, false); ev = NULL; }
  for (unsigned n = 0;; n++){
    const hipError_t e = ev ? hipEventQuery(ev) : hipStreamQuery(st);
    if (e != hipErrorNotReady){ if (ev) c->put_event(ev, false); return e; }
    if (slow) usleep(50);
    else {
      sched_yield();
      if ((n & 15) == 15 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) slow = true;
    }
  }
}

namespace {

// Device-to-host copy into the caller's (pageable) array: small ones directly, large ones through a pinned block from the cache
// (the driver stages pageable copies itself at a fraction of the link rate) and onto the caller's pages by the host threads.
int fetch_array(Ctx* c, hipStream_t st, void* dst, const void* src, size_t bytes){
  if (bytes == 0) return 0;
  if (bytes < ((size_t)4 << 20)){ HS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, st)); HS_HIP(hipstr::wait_stream(st)); return 0; }
  char* pin = (char*)c->pin_cache.get(bytes);
  if (!pin) return 1;
  if (hipMemcpyAsync(pin, src, bytes, hipMemcpyDeviceToHost, st) != hipSuccess || hipstr::wait_stream(st) != hipSuccess){
    c->pin_cache.put(pin); return fail("device-to-host copy failed"); }
  const size_t span = (size_t)8 << 20; const int n = (int)((bytes + span - 1)/span);
  hipstr::parallel_for(n, hipstr::host_threads(), [&](int i){ const size_t o = (size_t)i*span; memcpy((char*)dst + o, pin + o, std::min(span, bytes - o)); });
  c->pin_cache.put(pin);
  return 0;
}

// One stream per (host thread, device) for the one-shot and resident entry points: calls made from different host threads — several
// loci in flight, one thread each — neither queue behind one another nor wait for each other's work when they synchronise.
// Objects remember the stream of the thread that created them; streams live until the process ends.
hipStream_t thread_stream(Ctx* c){
  thread_local std::map<Ctx*, hipStream_t> mine;
  auto it = mine.find(c);
  if (it != mine.end()) return it->second;
  hipStream_t st = NULL;
  if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = c->stream;
  mine[c] = st;
  return st;
}

// A second stream per (host thread, device): the tables the device builds for itself (expand_kernels.hip) are inputs of the STR-block kernels
// only, so on the one-shot path they are built here while the caller's stream runs the column tables and the leading flanks (20 us of a
// 190 us one-locus call).  NULL when it cannot be made: the expansion then stays on the upload's stream.
hipStream_t thread_aux_stream(Ctx* c){
  thread_local std::map<Ctx*, hipStream_t> mine;
  auto it = mine.find(c);
  if (it != mine.end()) return it->second;
  hipStream_t st = NULL;
  if (hipSetDevice(c->device) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = NULL;
  mine[c] = st;
  return st;
}

}  // namespace

// what the other translation units of the library (trace.hip, em.hip, nw.hip) need from this one: api_internal.h
namespace hipstr {
int api_fail(const std::string& m){ return fail(m); }
Ctx* api_current_ctx(){ Ctx* c = current_ctx(); if (c && bind(c)) return NULL; return c; }
int api_bind(Ctx* c){ return bind(c); }
int api_device_tables(ApiTables* t){
  Ctx* c = api_current_ctx();
  if (!c) return 1;
  t->int_log = c->int_log; t->qual_correct = c->qc; t->qual_error = c->qe; t->m2m = c->m2m; t->m2i = c->m2i;
  t->stream = thread_stream(c); t->ctx = c;
  return 0;
}
void* dev_alloc(Ctx* c, size_t bytes){ return c->dev_cache.get(bytes); }
void  dev_free(Ctx* c, void* p){ c->dev_cache.put(p); }
void* pin_alloc(Ctx* c, size_t bytes){ return c->pin_cache.get(bytes); }
void  pin_free(Ctx* c, void* p){ c->pin_cache.put(p); }
}  // namespace hipstr

struct hipstr_dev_batch {
  Ctx* ctx = NULL;
  hipstr::Prepared prep;
  hs_dev_t h;             // host copy of the argument block (device pointers inside)
  hs_dev_t* d_args = NULL;
  std::vector<void*> dev_blocks, pin_blocks;      // from the context's caches
  int grid_y = 1, max_alleles = 1, n_lead_items = 0, n_trail_items = 0, trail_waves = 1;
  int max_rows = 0;              // longest flank rowset of the batch (rows of a flank block): picks the band shape of the trailing-flank sweep
  size_t grp_lds_bytes = 0, grp_pw_lds_bytes = 0;
  bool any_pw = false;           // some locus has alleles with piecewise simple lists (hs_str_group_kernel_pw)
  bool any_rp = false;           // ... with lists replayed in the grouped layout (hs_str_group_kernel_rp)
  bool any_short = false;        // some locus has tabulated alleles hs_str_group_kernel_p does not take (period above HS_GRP_MAXP)
  size_t lds_bytes = 0;
  hipEvent_t ev0 = NULL, ev1 = NULL;
  hipEvent_t ev_expand = NULL;            // the device-built tables are ready (recorded on the thread's aux stream; the first pass waits for it before the STR-block kernels)
  hipStream_t aux_pw_stream = NULL;      // the side stream a pass launched the interrupted alleles' kernels on (joined back by an event; hipstr_hmm_free waits for it all the same)
  hipStream_t aux_stream = NULL; bool expand_joined = false; hipStream_t expand_joined_on = NULL;      // (the stream whose first pass waited for ev_expand)
  hipEvent_t ev_h2d = NULL, ev_done = NULL, ev_d2h = NULL;     // upload finished / last pass finished / results in host_out (pipelined use)
  hipStream_t stream = NULL;                // launches, copies and waits of this batch default to it (the creating thread's stream)
  hipStream_t h2d_stream = NULL, d2h_stream = NULL;
  double* host_out = NULL;                  // pinned copy of aln_probs (fetch_begin)
  bool profiling = false, foreign_stream = false, sleepy_wait = false;
  bool d2h_pending = false;            // fetch_begin: the copy back is queued once ev_done has fired (fetch_poll / results_wait), not behind a cross-stream wait
  std::mutex d2h_m;
  std::vector<hipEvent_t> prof_pool;        // reusable events; every pass records 5 per chunk (phase boundaries)
  size_t prof_used = 0;
  int64_t algo_bytes = 0, dp_cells = 0;
  double t_prepare = 0, t_stage = 0;        // seconds spent in the host preparation / in packing the staging buffer
};

extern "C" {

const char* hipstr_last_error(void){ return g_err.c_str(); }

int hipstr_batch_out_offsets(const hipstr_batch_t* b, int64_t* out_off){
  if (!b || !out_off) return fail("null argument");
  int64_t acc = 0;
  for (int l = 0; l < b->n_loci; l++){
    out_off[l] = acc;
    acc += (int64_t)(b->read_off[l+1]-b->read_off[l]) * (b->hap_off[l+1]-b->hap_off[l]);
  }
  out_off[b->n_loci] = acc;
  return 0;
}

int hipstr_hmm_init(int device_ordinal){
  Ctx* c = ctx_for_device(device_ordinal);
  if (!c) return 1;
  t_ctx = c;
  return 0;
}

int64_t hipstr_hmm_trim(void){
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  int64_t bytes = 0;
  for (auto& kv : g_ctxs){
    Ctx* c = kv.second;
    if (hipSetDevice(c->device) != hipSuccess) continue;
    bytes += (int64_t)c->dev_cache.trim() + (int64_t)c->pin_cache.trim();
  }
  return bytes;
}

void hipstr_hmm_shutdown(void){
  std::lock_guard<std::mutex> lock(g_ctx_mutex);
  for (auto& kv : g_ctxs){
    Ctx* c = kv.second;
    if (hipSetDevice(c->device) != hipSuccess) continue;
    hipDeviceSynchronize();
    c->dev_cache.release(); c->pin_cache.release();
    hipFree(c->int_log); hipFree(c->qc); hipFree(c->qe); hipFree(c->m2m); hipFree(c->m2i);
    hipStreamDestroy(c->stream);
    delete c;
  }
  g_ctxs.clear();
  t_ctx = NULL;
}

int hipstr_calc_seed_bases(const hipstr_batch_t* b, int32_t* seeds){
  hipstr::ApiTimer prof_t(hipstr::PB_SEED_BASES);
  if (!b || !seeds) return fail("null argument");
  { std::string bad; if (hipstr::validate_tables(b, bad)) return fail(bad); }
  for (int l = 0; l < b->n_loci; l++)
    for (int r = b->read_off[l]; r < b->read_off[l+1]; r++){
      seeds[r] = hipstr::calc_seed_base(b, l, r);
      if (seeds[r] == -2) return fail("Invalid alignment seed or unrecognized CIGAR char (HapAligner.cpp:309,316)");
    }
  return 0;
}

void hipstr_hmm_free(hipstr_dev_batch_t* dev){
  if (!dev) return;
  if (dev->ctx && bind(dev->ctx) == 0){
    // blocks go back to the cache, which hands them to the next batch: whatever still runs on them must have finished
    if (!dev->dev_blocks.empty() || !dev->pin_blocks.empty()){
      hipStreamSynchronize(dev->stream);
      if (dev->h2d_stream && dev->h2d_stream != dev->stream) hipStreamSynchronize(dev->h2d_stream);
      if (dev->d2h_stream && dev->d2h_stream != dev->stream) hipStreamSynchronize(dev->d2h_stream);
      if (dev->ev_expand) hipStreamSynchronize(dev->aux_stream);       // (a batch that was never aligned)
      if (dev->aux_pw_stream && hipStreamQuery(dev->aux_pw_stream) != hipSuccess) hipStreamSynchronize(dev->aux_pw_stream); // (a pass ran the interrupted alleles' kernels on the aligning thread's side stream)
    }
    for (void* p : dev->dev_blocks) dev->ctx->dev_cache.put(p);
    for (void* p : dev->pin_blocks) dev->ctx->pin_cache.put(p);
  }
  if (dev->ctx){         // back to the context's pool (an event that is still pending is simply recorded again by its next user)
    dev->ctx->put_event(dev->ev0, true); dev->ctx->put_event(dev->ev1, true);
    dev->ctx->put_event(dev->ev_h2d, false); dev->ctx->put_event(dev->ev_done, false); dev->ctx->put_event(dev->ev_d2h, false);
    dev->ctx->put_event(dev->ev_expand, false);
  }
  for (hipEvent_t e : dev->prof_pool) hipEventDestroy(e);
  hipstr::recycle_prepared(dev->prep);       // the tables' host storage goes to the next batch (prep.h)
  delete dev;
}

hipstr_dev_batch_t* hipstr_hmm_upload(const hipstr_batch_t* batch){ return hipstr_hmm_upload_seeded(batch, NULL); }

// failure inside hipstr_hmm_upload: release what the half-built batch already owns
#define HS_HIP_DEV(call) do { hipError_t e_ = (call); if (e_ != hipSuccess){ \
  g_err = std::string(#call) + ": " + hipGetErrorString(e_); hipstr_hmm_free(dev); return NULL; } } while (0)

hipstr_dev_batch_t* hipstr_hmm_upload_seeded(const hipstr_batch_t* batch, const int32_t* seed_base){
  Ctx* ctx = hipstr::api_current_ctx();
  if (!ctx) return NULL;
  hipStream_t st = thread_stream(ctx);
  return hipstr::upload_on(ctx, batch, seed_base, st, st);
}

}  // extern "C"

// SURVEY.md §8(d) algorithmic traffic and flank-cell work of one pass (hipstr_hmm_workload; computed when asked for)
static void workload_of(hipstr_dev_batch_t* dev){
  const hipstr::Prepared& P = dev->prep;
  int64_t bytes = 0, cells = 0;
  for (const hs_locus_t& loc : P.loci){
    int64_t hap_bytes = 0, flank_rows = 0; int n_re = 0;
    for (int k = 0; k < loc.n_alleles; k++){
      const hs_allele_t& al = P.alleles[loc.hap_begin + k];
      if (!al.realign) continue;
      const int B = P.stropts[al.str_opt[0]].B;
      hap_bytes += 2*(int64_t)(al.n_flank + B) + 8*13 + 2*6*(int64_t)B + 16;
      flank_rows += al.n_flank; n_re++;
    }
    bytes += hap_bytes;
    for (int i = 0; i < loc.n_reads; i++){
      const hs_read_t& rd = P.reads[loc.read_begin + i];
      if (!P.realign_read[loc.read_begin + i] || rd.seed < 0) continue;
      bytes += 2*(int64_t)rd.len + 8 + 4 + 8*(int64_t)n_re;
      cells += (int64_t)(rd.len - 1) * flank_rows;      // (n_L + n_R) x flank rows, leading flank counted per allele as the reference recomputes it without reuse
    }
  }
  dev->algo_bytes = bytes; dev->dp_cells = cells;
}

// The upload with the copy on a stream of the caller's choice (the pipelined path copies on its own stream so that the next
// batch's tables travel while the previous batch's kernels run); hipstr_hmm_align waits for it through an event.
hipstr_dev_batch_t* hipstr::upload_on(Ctx* ctx, const hipstr_batch_t* batch, const int32_t* seed_base, hipStream_t copy_stream, hipStream_t compute_stream, bool reads_pinned){
  if (bind(ctx)) return NULL;
  hipstr_dev_batch_t* dev = new hipstr_dev_batch_t();
  dev->ctx = ctx;
  dev->h2d_stream = copy_stream;
  dev->stream = compute_stream;
  std::string err;
  // workspace budget (doubles per workspace; there are two large ones): HIPSTR_WS_GIB, else a fifth of the free HBM, at most 24 GiB
  int64_t budget = (int64_t)3 << 30;
  if (getenv("HIPSTR_WS_GIB")) budget = (int64_t)(atof(getenv("HIPSTR_WS_GIB"))*134217728.0);
  if (budget < 1024) budget = 1024;
  const auto t_prep0 = std::chrono::steady_clock::now();
  hipstr::adopt_recycled(dev->prep);
  if (hipstr::prepare_batch(batch, dev->prep, err, budget, seed_base)){ g_err = err; hipstr::recycle_prepared(dev->prep); delete dev; return NULL; }
  dev->t_prepare = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_prep0).count();
  if (getenv("HIPSTR_TIMING")) fprintf(stderr, "hipstr_hmm_upload: prepare_batch %.3f ms\n", 1e3*dev->t_prepare);
  hipstr::Prepared& P = dev->prep;
  hs_dev_t& h = dev->h;
  memset(&h, 0, sizeof h);
  // ---- one device block for everything the host fills, one pinned block to stage it, one copy
  const auto t_stage0 = std::chrono::steady_clock::now();
  struct Piece { const void* src; size_t bytes, off; };
  std::vector<Piece> pieces;        // what is copied where; a pool gathered from fragments contributes one piece per fragment
  size_t total = 0;
  auto place = [&](const void* src, size_t bytes){ total = (total + 255) & ~(size_t)255; pieces.push_back(Piece{src, bytes, total}); total += bytes ? bytes : 1; return pieces.size() - 1; };
  // a large pool: the batch's own vector (single-threaded preparation) followed by the fragments' (threaded), back to back
  // (tail_bytes: room behind the pool that the host does not fill — the generated part of the f64 pool)
  auto place_pool = [&](const void* own, size_t own_bytes, auto frag_ptr, size_t elem, size_t tail_bytes){
    total = (total + 255) & ~(size_t)255;
    const size_t first = pieces.size();
    pieces.push_back(Piece{own, own_bytes, total}); total += own_bytes;
    for (const hipstr::Prepared& f : P.frags){ const auto& v = f.*frag_ptr; pieces.push_back(Piece{v.data(), v.size()*elem, total}); total += v.size()*elem; }
    total += tail_bytes + 1;
    return first;
  };
  dev->n_lead_items = (int)P.lead_items.size();
  dev->n_trail_items = (int)P.trail_items.size();
#define PL(vec) place((vec).data(), (vec).size()*sizeof((vec)[0]))
  const size_t i_loci = PL(P.loci), i_alleles = PL(P.alleles), i_stropts = PL(P.stropts), i_rowsets = PL(P.rowsets),
    i_rows = place_pool(P.rows.data(), P.rows.size()*sizeof(hs_row_t), &hipstr::Prepared::rows, sizeof(hs_row_t), 0),
    i_visits = place_pool(P.visits.data(), P.visits.size()*sizeof(hs_visit_t), &hipstr::Prepared::visits, sizeof(hs_visit_t), 0),
    i_f64 = place_pool(P.f64pool.data(), P.f64pool.size()*sizeof(double), &hipstr::Prepared::f64pool, sizeof(double), sizeof(double)*(size_t)P.gen_f64),
    i_chars = place_pool(P.chars.data(), P.chars.size(), &hipstr::Prepared::chars, 1, 0),
    i_recd = PL(P.rec_descs), i_pmf = PL(P.pmf13),
    i_reads = PL(P.reads), i_active = PL(P.active),
    i_ws = PL(P.ws), i_tg = PL(P.tgroups), i_tm = PL(P.tmembers), i_tp = PL(P.tpack), i_ord = PL(P.str_order), i_ndr = PL(P.nd_rows);
#undef PL
  // the work items of the three phases, one array on the device: lead | trail | STR
  total = (total + 255) & ~(size_t)255;
  const size_t i_items = pieces.size();
  pieces.push_back(Piece{P.lead_items.data(), P.lead_items.size()*sizeof(hs_item_t), total}); total += P.lead_items.size()*sizeof(hs_item_t);
  pieces.push_back(Piece{P.trail_items.data(), P.trail_items.size()*sizeof(hs_item_t), total}); total += P.trail_items.size()*sizeof(hs_item_t);
  pieces.push_back(Piece{P.str_items.data(), P.str_items.size()*sizeof(hs_item_t), total}); total += P.str_items.size()*sizeof(hs_item_t) + 1;
  const size_t i_args = place(&h, sizeof h);
  // the reads' bases and qualities come last: a caller whose arrays are pinned (the stream's batches) has them copied from where they lie,
  // without a pass through the staging block
  const size_t n_bases = P.reads.empty() ? 0 : (size_t)batch->base_off[P.reads.size()];
  const size_t packed_total = reads_pinned ? ((total + 255) & ~(size_t)255) : 0;
  const size_t i_bases = place(reads_pinned ? NULL : batch->bases, n_bases), i_quals = place(reads_pinned ? NULL : batch->quals, n_bases);
  const size_t stage_total = reads_pinned ? packed_total : total;
  const bool up_prof = hipstr::api_profile_on();
  double up_blocks = 0, up_t = up_prof ? hipstr::ApiTimer::now() : 0;
  char* dblk = (char*)ctx->dev_cache.get(total);
  if (!dblk){ hipstr_hmm_free(dev); return NULL; }
  dev->dev_blocks.push_back(dblk);
  char* stage = (char*)ctx->pin_cache.get(stage_total);
  if (!stage){ hipstr_hmm_free(dev); return NULL; }
  dev->pin_blocks.push_back(stage);
  if (up_prof) up_blocks += hipstr::ApiTimer::now() - up_t;
  auto at = [&](size_t i){ return dblk + pieces[i].off; };
  h.loci = (const hs_locus_t*)at(i_loci); h.alleles = (const hs_allele_t*)at(i_alleles); h.stropts = (const hs_stropt_t*)at(i_stropts);
  h.rowsets = (const hs_rowset_t*)at(i_rowsets); h.rows = (const hs_row_t*)at(i_rows); h.visits = (const hs_visit_t*)at(i_visits);
  h.f64pool = (const double*)at(i_f64); h.chars = (const char*)at(i_chars); h.reads = (const hs_read_t*)at(i_reads);
  h.active = (const int32_t*)at(i_active); h.items = (const hs_item_t*)at(i_items); h.ws = (const hs_ws_t*)at(i_ws);
  h.tgroups = (const hs_tgroup_t*)at(i_tg); h.tmembers = (const int32_t*)at(i_tm); h.tpack = (const int32_t*)at(i_tp);
  h.str_order = (const int32_t*)at(i_ord); h.rec_descs = (const hs_recdesc_t*)at(i_recd); h.pmf13 = (const double*)at(i_pmf); h.nd_rows = (const hs_ndrow_t*)at(i_ndr); h.bases = at(i_bases); h.quals = at(i_quals);
  dev->d_args = (hs_dev_t*)at(i_args);
  // ---- output + workspaces (device only)
  auto dalloc = [&](size_t bytes) -> void* { void* p = ctx->dev_cache.get(bytes ? bytes : 1); if (p) dev->dev_blocks.push_back(p); return p; };
  if (up_prof) up_t = hipstr::ApiTimer::now();
  const size_t out_bytes = (size_t)(P.n_out ? P.n_out : 1) * sizeof(double);
  h.aln_probs = (double*)dalloc(out_bytes);
  h.ws_mr = (double*)dalloc(sizeof(double)*(size_t)P.ws_mr_size); h.ws_lt = (double*)dalloc(sizeof(double)*(size_t)P.ws_lt_size);
  h.ws_lead = (double*)dalloc(sizeof(double)*(size_t)P.ws_lead_size); h.ws_col = (double*)dalloc(sizeof(double)*(size_t)P.ws_col_size);
  h.ws_nd = (double*)dalloc(sizeof(double)*(size_t)P.ws_nd_size);
  h.grp_recs = (const int32_t*)dalloc(sizeof(int32_t)*HS_GRP_REC_DWORDS*P.rec_descs.size());        // assembled on the device (hs_expand_recs_kernel)
  h.n_stropts = (int32_t)P.stropts.size(); h.n_recs = (int32_t)P.rec_descs.size(); h.f64_gen_base = (int64_t)P.n_f64();
  // trailing-flank kernel: persistent wavefronts, each with two band-boundary rows of [max side columns][64 lanes][M,D]
  h.band_cols = P.max_side_len > 0 ? P.max_side_len : 1;
  dev->trail_waves = (int)std::min<size_t>(P.trail_items.size() ? P.trail_items.size() : 1, 256 * 16);
  dev->max_rows = 0;
  for (const hs_rowset_t& rs : P.rowsets) dev->max_rows = std::max(dev->max_rows, (int)rs.len);
  h.ws_band = (double*)dalloc(sizeof(double)*(size_t)dev->trail_waves*h.band_cols*64*2);
  h.lts_rows = 1; h.ws_lts = NULL;          // (fields of the removed fused trailing-flank item: the argument block keeps its layout)
  h.n_active = (int32_t)P.active.size();
  // [n_active] re-do flags of hs_str_kernel | [2 x chunks] work counters of hs_lead_kernel and hs_trail_kernel
  h.redo = (int32_t*)dalloc(sizeof(int32_t)*((size_t)h.n_active + 2*P.chunks.size() + 2));
  if (!h.aln_probs || !h.ws_mr || !h.ws_lt || !h.ws_lead || !h.ws_col || !h.ws_nd || !h.ws_band || !h.redo || !h.grp_recs){ hipstr_hmm_free(dev); return NULL; }
  if (up_prof){ up_blocks += hipstr::ApiTimer::now() - up_t; hipstr::api_profile_add(hipstr::PB_UP_BLOCKS, up_blocks); }
  const hipstr::HostTables& T = hipstr::host_tables();
  h.int_log = ctx->int_log; h.qual_correct = ctx->qc; h.qual_error = ctx->qe; h.m2m = ctx->m2m; h.m2i = ctx->m2i;
  h.log_thresh = T.log_thresh; h.log_half = T.log_half;
  h.lds_len = P.max_read_len;
  h.debug_redo = getenv("HIPSTR_DEBUG_REDO") ? atoi(getenv("HIPSTR_DEBUG_REDO")) : 0;
  // Workgroups: one per (active read[, side]), times enough allele chunks to put >= ~8192 wavefronts on the 256 CUs
  int maxA = 1;
  for (const hs_locus_t& l : P.loci) maxA = l.n_alleles > maxA ? l.n_alleles : maxA;
  int gy = 1;
  if (h.n_active > 0 && h.n_active < 4096){ gy = (4096 + h.n_active - 1) / h.n_active; if (gy > maxA) gy = maxA; }
  h.allele_chunk = (maxA + gy - 1) / gy;
  dev->grid_y = (maxA + h.allele_chunk - 1) / h.allele_chunk;
  dev->max_alleles = maxA;
  h.max_B = P.max_B;
  dev->lds_bytes = hs_str_lds_bytes(h.lds_len, h.max_B);
  if (dev->lds_bytes > 160*1024){ g_err = "batch needs more than 160 KiB of LDS per workgroup"; hipstr_hmm_free(dev); return NULL; }
  h.grp_nd_cap = std::max(2, (P.grp_nd_cap + 1) & ~1);
  dev->any_short = false;
  for (const hs_locus_t& l : P.loci) dev->any_short |= (l.n_short[0] > 0 || l.n_short[1] > 0);
  dev->any_pw = false;
  for (const hs_locus_t& l : P.loci) dev->any_pw |= (l.n_pw[0] > l.n_tab[0] || l.n_pw[1] > l.n_tab[1]);
  dev->any_rp = false;
  for (const hs_locus_t& l : P.loci) dev->any_rp |= (l.n_rp[0] > l.n_pw[0] || l.n_rp[1] > l.n_pw[1]);
  dev->grp_lds_bytes = hs_str_group_lds_bytes(h.max_B, h.grp_nd_cap, 0);
  dev->grp_pw_lds_bytes = hs_str_group_lds_bytes(h.max_B, h.grp_nd_cap, 1);
  if (dev->grp_pw_lds_bytes > 48*1024){
    HS_HIP_DEV(hipFuncSetAttribute((const void*)hs_str_group_kernel_pw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev->grp_pw_lds_bytes));
    HS_HIP_DEV(hipFuncSetAttribute((const void*)hs_str_group_kernel_rp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev->grp_pw_lds_bytes));
  }
  if (getenv("HIPSTR_TIMING")) fprintf(stderr, "hipstr_hmm_upload: STR group kernel LDS %zu bytes (max block %d, read-end table %d doubles, %zu groups)\n", dev->grp_lds_bytes, h.max_B, h.grp_nd_cap, P.str_items.size());
  if (dev->grp_lds_bytes > 48*1024){
    HS_HIP_DEV(hipFuncSetAttribute((const void*)hs_str_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev->grp_lds_bytes));
  }
  if (dev->lds_bytes > 48*1024){
    HS_HIP_DEV(hipFuncSetAttribute((const void*)hs_str_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev->lds_bytes));
    HS_HIP_DEV(hipFuncSetAttribute((const void*)hs_str_kernel_generic, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dev->lds_bytes));
  }
  // ---- pack (large pieces cut into 4 MiB spans so that the copy is shared by the host threads) and send
  {
    struct Span { char* dst; const char* src; size_t n; };
    std::vector<Span> spans;
    for (const Piece& pc : pieces)
      for (size_t o = 0; pc.src && o < pc.bytes; o += (size_t)4 << 20)
        spans.push_back(Span{stage + pc.off + o, (const char*)pc.src + o, std::min<size_t>((size_t)4 << 20, pc.bytes - o)});
    hipstr::parallel_for((int)spans.size(), stage_total > ((size_t)8 << 20) ? hipstr::host_threads() : 1, [&](int i){ memcpy(spans[i].dst, spans[i].src, spans[i].n); });
  }
  { hipstr::Prepared only_frags; only_frags.frags.swap(P.frags); hipstr::recycle_prepared(only_frags); }      // the fragments' pools are packed: their storage can serve the next batch already
  dev->t_stage = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_stage0).count();
  if (up_prof) up_t = hipstr::ApiTimer::now();
  HS_HIP_DEV(hipMemcpyAsync(dblk, stage, stage_total, hipMemcpyHostToDevice, copy_stream));
  if (reads_pinned && n_bases){
    HS_HIP_DEV(hipMemcpyAsync(dblk + pieces[i_bases].off, batch->bases, n_bases, hipMemcpyHostToDevice, copy_stream));
    HS_HIP_DEV(hipMemcpyAsync(dblk + pieces[i_quals].off, batch->quals, n_bases, hipMemcpyHostToDevice, copy_stream));
  }
  if (up_prof){ const double n_ = hipstr::ApiTimer::now(); hipstr::api_profile_add(hipstr::PB_UP_MEMCPY, n_ - up_t); up_t = n_; }
  // what the device builds for itself (expand_kernels.hip), behind the copies on the same stream and in front of everything else
  // One-shot path (copies and kernels on ONE stream): aside, on the thread's second stream, beside the column tables and the leading flanks
  hipStream_t exp_stream = copy_stream;
  if (copy_stream == compute_stream && (P.gen_f64 > 0 || !P.rec_descs.empty()) && (dev->aux_stream = thread_aux_stream(ctx)) != NULL){
    hipEvent_t fork = ctx->get_event(false); dev->ev_expand = ctx->get_event(false);
    if (fork && dev->ev_expand && hipEventRecord(fork, copy_stream) == hipSuccess && hipStreamWaitEvent(dev->aux_stream, fork, 0) == hipSuccess) exp_stream = dev->aux_stream;
    else { ctx->put_event(dev->ev_expand, false); dev->ev_expand = NULL; }
    ctx->put_event(fork, false);          // (its wait is queued: the event may be recorded again by its next user)
  }
  if (P.gen_f64 > 0) hipLaunchKernelGGL(hs_expand_stropts_kernel, dim3((unsigned)((P.stropts.size() + 3)/4)), dim3(256), 0, exp_stream, (const hs_dev_t*)dev->d_args);
  if (!P.rec_descs.empty()) hipLaunchKernelGGL(hs_expand_recs_kernel, dim3((unsigned)((P.rec_descs.size()*HS_GRP_REC_DWORDS + 255)/256)), dim3(256), 0, exp_stream, (const hs_dev_t*)dev->d_args);
  HS_HIP_DEV(hipGetLastError());
  if (dev->ev_expand) HS_HIP_DEV(hipEventRecord(dev->ev_expand, dev->aux_stream));
  // entries no kernel writes (reads or alleles that are not realigned, reads without a seed) read as 0 for whoever takes the device pointer
  // (hipstr_hmm_dev_aln_probs); a batch where every read is active and every allele realigned — the usual one — has none: no fill launch
  bool all_written = P.active.size() == P.reads.size();
  for (size_t li = 0; all_written && li < P.loci.size(); li++) all_written = P.loci[li].n_re == P.loci[li].n_alleles;
  if (!all_written) HS_HIP_DEV(hipMemsetAsync(h.aln_probs, 0, out_bytes, copy_stream));
  if (up_prof){ const double n_ = hipstr::ApiTimer::now(); hipstr::api_profile_add(hipstr::PB_UP_EXPAND, n_ - up_t); up_t = n_; }
  dev->ev0 = ctx->get_event(true); dev->ev1 = ctx->get_event(true);
  // The stream's batches take tens of milliseconds and their collectors must not burn a core waiting: hipEventSynchronize spins at 100 %
  // of a CPU on this stack whatever the event's flags (tools/wait_probe.hip: 41.7 ms of thread CPU per 41.7 ms of waiting, also with
  // hipEventBlockingSync), a query + usleep loop costs 0.6 ms (results_wait).
  constexpr bool spin = false;
  dev->sleepy_wait = reads_pinned && !spin;
  dev->ev_h2d = ctx->get_event(false); dev->ev_done = ctx->get_event(false); dev->ev_d2h = ctx->get_event(false);
  if (!dev->ev0 || !dev->ev1 || !dev->ev_h2d || !dev->ev_done || !dev->ev_d2h){ g_err = "hipEventCreate failed"; hipstr_hmm_free(dev); return NULL; }
  HS_HIP_DEV(hipEventRecord(dev->ev_h2d, copy_stream));
  if (up_prof) hipstr::api_profile_add(hipstr::PB_UP_EVENTS, hipstr::ApiTimer::now() - up_t);
  if (getenv("HIPSTR_TIMING"))
    fprintf(stderr, "hipstr_hmm_upload: total %.3f ms (prepare %.3f, blocks + staging %.3f), %zu B of tables, %lld alignments\n",
            1e3*std::chrono::duration<double>(std::chrono::steady_clock::now() - t_prep0).count(), 1e3*dev->t_prepare, 1e3*dev->t_stage, total, (long long)P.n_alignments);
  return dev;
}

// Pipelined fetch: the device-to-host copy of aln_probs queued on `copy_stream` behind the batch's last pass; results_wait blocks
// until it has landed in the batch's pinned buffer; scatter_loci then applies the reference's output contract for a range of loci.
// Waiting for an event without a core AND without a sleep's granularity where the wait is short (a one-locus batch of a latency-bound caller goes
// through three of these: upload, kernels, copy back): yield for the first 300 us, then sleep 100 us at a time.
static hipError_t event_wait_sleepy(hipEvent_t ev){
  const auto t0 = std::chrono::steady_clock::now();
  hipError_t e;
  for (unsigned n = 0; (e = hipEventQuery(ev)) == hipErrorNotReady; n++){
    if (std::chrono::steady_clock::now() - t0 < std::chrono::microseconds(300)) sched_yield(); else usleep(100);
  }
  return e;
}
int hipstr::fetch_begin(hipstr_dev_batch_t* dev, hipStream_t compute_stream, hipStream_t copy_stream){
  if (bind(dev->ctx)) return 1;
  const hipstr::Prepared& P = dev->prep;
  HS_HIP(hipEventRecord(dev->ev_done, compute_stream));
  if (P.n_out && !dev->host_out){
    dev->host_out = (double*)dev->ctx->pin_cache.get((size_t)P.n_out*sizeof(double));
    if (!dev->host_out) return 1;
    dev->pin_blocks.push_back(dev->host_out);
  }
  dev->d2h_stream = copy_stream;
  // Round 6: the copy back is NOT queued behind a hipStreamWaitEvent(copy_stream, ev_done).  A cross-stream wait for an event that tens of
  // milliseconds of kernels stand in front of is resolved by the HSA runtime's event thread ON THE CPU: it spins in the KFD's wait call until
  // the event fires (tools/r06_rt_probe.py: 2.65 s of CPU per 3 s of such waits, nothing for the same kernels without the wait) — with a
  // stream running that was a core in use for as long as the device was busy, a third of a rank's two-CPU allowance.  The copy is queued by
  // whoever finds ev_done fired first: a worker passing by (fetch_poll) or the collector that waits for the results (results_wait).
  std::lock_guard<std::mutex> lg(dev->d2h_m);
  dev->d2h_pending = true;
  return 0;
}
// ev_done has fired (the caller saw it): the copy back and its event, on the copy stream, with nothing to wait for
static int fetch_issue_locked(hipstr_dev_batch_t* dev){
  const hipstr::Prepared& P = dev->prep;
  if (P.n_out) HS_HIP(hipMemcpyAsync(dev->host_out, dev->h.aln_probs, (size_t)P.n_out*sizeof(double), hipMemcpyDeviceToHost, dev->d2h_stream));
  HS_HIP(hipEventRecord(dev->ev_d2h, dev->d2h_stream));
  dev->d2h_pending = false;
  return 0;
}
// a look without waiting: queues the copy back if the batch's kernels are done.  0 = nothing pending any more, 1 = still running, -1 = error
int hipstr::fetch_poll(hipstr_dev_batch_t* dev){
  std::unique_lock<std::mutex> lg(dev->d2h_m, std::try_to_lock);
  if (!lg.owns_lock()) return 1;
  if (!dev->d2h_pending) return 0;
  if (bind(dev->ctx)) return -1;
  const hipError_t e = hipEventQuery(dev->ev_done);
  if (e == hipErrorNotReady) return 1;
  if (e != hipSuccess){ g_err = std::string("hipEventQuery: ") + hipGetErrorString(e); return -1; }
  return fetch_issue_locked(dev) == 0 ? 0 : -1;
}
int hipstr::results_wait(hipstr_dev_batch_t* dev){
  if (bind(dev->ctx)) return 1;
  {
    std::lock_guard<std::mutex> lg(dev->d2h_m);
    if (dev->d2h_pending){
      if (dev->sleepy_wait){
        const hipError_t e = event_wait_sleepy(dev->ev_done);
        if (e != hipSuccess){ g_err = std::string("hipEventQuery: ") + hipGetErrorString(e); return 1; }
      } else HS_HIP(hipEventSynchronize(dev->ev_done));
      if (fetch_issue_locked(dev)) return 1;
    }
  }
  if (dev->sleepy_wait){
    const hipError_t e = event_wait_sleepy(dev->ev_d2h);
    if (e != hipSuccess){ g_err = std::string("hipEventQuery: ") + hipGetErrorString(e); return 1; }
    return 0;
  }
  HS_HIP(hipEventSynchronize(dev->ev_d2h));
  return 0;
}
void hipstr::scatter_loci(const hipstr_dev_batch_t* dev, int l0, int l1, double* aln_probs, int32_t* seeds){
  const hipstr::Prepared& P = dev->prep;
  if (l1 <= l0) return;
  const int64_t out0 = P.loci[l0].out_off; const int r0 = P.loci[l0].read_begin;
  for (int li = l0; li < l1; li++){
    const hs_locus_t& loc = P.loci[li];
    const int A = loc.n_alleles;
    const bool all_haps = loc.n_re == A;
    // the usual locus — every read realigned and seeded, every haplotype realigned — is one block in both layouts
    bool plain = all_haps;
    for (int i = 0; plain && i < loc.n_reads; i++) plain = P.realign_read[loc.read_begin + i] && P.seeds[loc.read_begin + i] >= 0;
    if (plain){
      memcpy(aln_probs + (loc.out_off - out0), dev->host_out + loc.out_off, sizeof(double)*(size_t)loc.n_reads*(size_t)A);
      memcpy(seeds + (loc.read_begin - r0), P.seeds.data() + loc.read_begin, sizeof(int32_t)*(size_t)loc.n_reads);
      continue;
    }
    for (int i = 0; i < loc.n_reads; i++){
      const int r = loc.read_begin + i;
      if (!P.realign_read[r]) continue;                       // HapAligner.cpp:326-329
      seeds[r - r0] = P.seeds[r];
      double* dst = aln_probs + (loc.out_off - out0) + (int64_t)i*A;
      const double* src = dev->host_out + loc.out_off + (int64_t)i*A;
      if (P.seeds[r] == -1){ for (int k = 0; k < A; k++) dst[k] = 0; continue; }   // HapAligner.cpp:333-337
      if (all_haps) memcpy(dst, src, sizeof(double)*(size_t)A);
      else for (int k = 0; k < A; k++) if (P.realign_hap[loc.hap_begin + k]) dst[k] = src[k];   // HapAligner.cpp:615-619
    }
  }
}
// Releases a batch whose results have landed on the host (its D2H event completed, hence every kernel that touched its blocks):
// the blocks can go back to the cache without synchronising the stream, which would stall the batches queued behind it.
void hipstr::free_landed(hipstr_dev_batch_t* dev, bool landed){
  if (!dev) return;
  if (landed && dev->ctx && bind(dev->ctx) == 0){
    for (void* p : dev->dev_blocks) dev->ctx->dev_cache.put(p);
    for (void* p : dev->pin_blocks) dev->ctx->pin_cache.put(p);
    dev->dev_blocks.clear(); dev->pin_blocks.clear();
  }
  hipstr_hmm_free(dev);
}
double hipstr::batch_prepare_seconds(const hipstr_dev_batch_t* dev){ return dev ? dev->t_prepare : 0.0; }
hipStream_t hipstr::ctx_stream(Ctx* c){ return c->stream; }
int hipstr::ctx_device(Ctx* c){ return c->device; }

extern "C" {

int hipstr_hmm_align(hipstr_dev_batch_t* dev, void* hip_stream){
  if (!dev) return fail("null device batch");
  if (bind(dev->ctx)) return 1;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : dev->stream;
  if (hip_stream && (hipStream_t)hip_stream != dev->stream) dev->foreign_stream = true;
  // the tables were sent on another stream.  Also with nothing to align: what follows on `st` (the copy back, the events that release the
  // batch's blocks to the cache) must come after the upload that is still writing into those blocks
  // (round 6, the stream's batches — the caller is a worker thread with a batch's time to spare: the two events a batch's kernels depend on, the
  //  upload and the table expansion, are waited for HERE, asleep, instead of by hipStreamWaitEvent on a stream that is busy with the previous
  //  batch: a cross-stream wait keeps the HSA runtime's event thread spinning on a CPU while it is pending, fetch_begin.  One-shot calls keep the
  //  device-side waits: theirs are microseconds long and a host round trip would be their latency.
  //  Measured with the two side by side, profiles/r06_host_side_dependencies.txt: the event thread's 0.65-1.04 s per second of streaming gone,
  //  p30 under a rank's two CPUs 8.0-8.5 -> 5.5-6.3 us of CPU per locus, NS 185 -> 80.)
  const bool host_deps = dev->sleepy_wait && !hip_stream;
  auto host_wait = [&](hipEvent_t ev) -> int {
    const hipError_t e = event_wait_sleepy(ev);
    if (e != hipSuccess) return fail(std::string("hipEventQuery: ") + hipGetErrorString(e));
    return 0;
  };
  if (dev->h2d_stream != st){ if (host_deps){ if (host_wait(dev->ev_h2d)) return 1; } else HS_HIP(hipStreamWaitEvent(st, dev->ev_h2d, 0)); }
  if (host_deps && dev->ev_expand && !dev->expand_joined){ if (host_wait(dev->ev_expand)) return 1; dev->expand_joined = true; dev->expand_joined_on = st; }
  if (dev->h.n_active == 0) return 0;
  const hs_dev_t* dp = dev->d_args;
  auto mark = [&]() -> int {
    if (!dev->profiling) return 0;
    if (dev->prof_used == dev->prof_pool.size()){ hipEvent_t e; HS_HIP(hipEventCreate(&e)); dev->prof_pool.push_back(e); }
    HS_HIP(hipEventRecord(dev->prof_pool[dev->prof_used++], st));
    return 0;
  };
  const int n_clear = (int)((size_t)dev->h.n_active + 2*dev->prep.chunks.size() + 2);        // re-do flags + item counters: cleared by hs_col_kernel
  int chunk_no = 0;
  for (const hipstr::Prepared::Chunk& ch : dev->prep.chunks){
    const unsigned nact = ch.active_end - ch.active_begin;
    if (mark()) return 1;
    // leading flanks: persistent wavefronts striding over (locus side, distinct flank, 64 reads) items
    hs_launch_lead2(nact, (unsigned)std::max(1, std::min(dev->trail_waves, ch.lead_end - ch.lead_begin)), st, dp, ch.active_begin, ch.lead_begin, ch.lead_end, 2*chunk_no, dev->h.band_cols, n_clear);
    if (mark()) return 1;
    // (once per stream: a later pass on ANOTHER stream of the caller's is not ordered behind the first one's wait)
    if (dev->ev_expand && (!dev->expand_joined || dev->expand_joined_on != st)){ HS_HIP(hipStreamWaitEvent(st, dev->ev_expand, 0)); dev->expand_joined = true; dev->expand_joined_on = st; }
    // tabulated alleles: reads of a locus side packed into workgroups (HIPSTR_STR_GROUP=0: one workgroup per read, for comparison)
    const bool str_group = !(getenv("HIPSTR_STR_GROUP") && atoi(getenv("HIPSTR_STR_GROUP")) == 0);
    if (!str_group) hipLaunchKernelGGL(hs_str_kernel, dim3(nact, dev->grid_y), dim3(128), dev->lds_bytes, st, dp, ch.active_begin, 0);
    else {
      // blocks of at least six repeat units (nearly all) through the kernel with a compile-time period, the shorter ones as before
      constexpr bool group_p = true;
      // A call of a locus or two: the interrupted alleles' kernels beside the tabulated alleles' (different alleles of the same reads: disjoint
      // outputs; the re-do marks both may set are the same value) on the thread's second stream — 18 us of a 180 us call
      hipStream_t st_pw = st; hipEvent_t ev_pw = NULL, ev_fork = NULL;
      if (ch.str_end > ch.str_begin && ch.str_end - ch.str_begin <= 256 && group_p && (dev->any_pw || dev->any_rp)){
        hipStream_t aux = thread_aux_stream(dev->ctx);
        ev_fork = aux ? dev->ctx->get_event(false) : NULL;
        ev_pw = ev_fork ? dev->ctx->get_event(false) : NULL;
        if (ev_pw) st_pw = aux; else { dev->ctx->put_event(ev_fork, false); ev_fork = NULL; }
      }
      if (ch.str_end > ch.str_begin){
        if (group_p && dev->prep.ws_nd_size > 0)       // read-end deletion sums of the tabulated alleles, every (row, column) a lane
          hipLaunchKernelGGL(hs_nd_kernel, dim3(nact, 2), dim3(256), 0, st, dp, ch.active_begin);
        if (ev_pw){      // (back to the pool once its wait is queued: the next user records it again)
          const bool ok = hipEventRecord(ev_fork, st) == hipSuccess && hipStreamWaitEvent(st_pw, ev_fork, 0) == hipSuccess;
          dev->ctx->put_event(ev_fork, false);
          if (!ok){ dev->ctx->put_event(ev_pw, false); return fail("hipEventRecord / hipStreamWaitEvent failed"); }      // (nothing launched on the side stream yet)
        }
        if (group_p) hipLaunchKernelGGL(hs_str_group_kernel_p, dim3(ch.str_end - ch.str_begin, dev->grid_y), dim3(HS_GRP_COLS), hs_str_group_p_lds_bytes(), st, dp,
                                        dev->n_lead_items + dev->n_trail_items + ch.str_begin);
        if (!group_p || dev->any_short)      // (periods above HS_GRP_MAXP only, once hs_str_group_kernel_p is on)
          hipLaunchKernelGGL(hs_str_group_kernel, dim3(ch.str_end - ch.str_begin, dev->grid_y), dim3(HS_GRP_COLS), dev->grp_lds_bytes, st, dp,
                             dev->n_lead_items + dev->n_trail_items + ch.str_begin, group_p ? 1 : 0);
      }
      if (dev->any_pw && ch.str_end > ch.str_begin)       // interrupted repeats: the piecewise simple lists' closed forms, grouped like the tabulated ones
        hipLaunchKernelGGL(hs_str_group_kernel_pw, dim3(ch.str_end - ch.str_begin, dev->grid_y), dim3(HS_GRP_COLS), dev->grp_pw_lds_bytes, st_pw, dp,
                           dev->n_lead_items + dev->n_trail_items + ch.str_begin);
      if (dev->any_rp && ch.str_end > ch.str_begin)       // three and more interruptions: lists without a closed form, replayed in the grouped layout
        hipLaunchKernelGGL(hs_str_group_kernel_rp, dim3(ch.str_end - ch.str_begin, dev->grid_y), dim3(HS_GRP_COLS), dev->grp_pw_lds_bytes, st_pw, dp,
                           dev->n_lead_items + dev->n_trail_items + ch.str_begin);
      if (ev_pw){
        // the join: whatever happens the event goes back to the pool, and a join that could not be queued waits for the side stream here —
        // its kernels read and write this batch's blocks, which the caller is about to return to the cache
        dev->aux_pw_stream = st_pw;
        const bool ok = hipEventRecord(ev_pw, st_pw) == hipSuccess && hipStreamWaitEvent(st, ev_pw, 0) == hipSuccess;
        dev->ctx->put_event(ev_pw, false);
        if (!ok){ hipStreamSynchronize(st_pw); return fail("hipEventRecord / hipStreamWaitEvent failed (side stream of the interrupted alleles' kernels)"); }
      }
      if (ch.n_long_sides > 0)        // sides with more columns than a group holds: one workgroup per read as before
        hipLaunchKernelGGL(hs_str_kernel, dim3(nact, dev->grid_y), dim3(128), dev->lds_bytes, st, dp, ch.active_begin, 1);
    }
    // alleles without a tabulated closed form (interrupted repeats, very long blocks) and whatever hs_str_kernel marked HS_REDO
    hipLaunchKernelGGL(hs_str_kernel_generic, dim3(nact, dev->grid_y), dim3(128), dev->lds_bytes, st, dp, ch.active_begin, str_group ? 1 : 0);
    if (mark()) return 1;
    if (ch.trail_end > ch.trail_begin)     // trailing flanks: persistent wavefronts striding over (read side, allele group) items
      hs_launch_trail((unsigned)std::min(dev->trail_waves, ch.trail_end - ch.trail_begin), st, dp,
                      dev->n_lead_items + ch.trail_begin, dev->n_lead_items + ch.trail_end, 2*chunk_no + 1, dev->h.band_cols, dev->max_rows);
    if (mark()) return 1;
    hipLaunchKernelGGL(hs_combine_kernel, dim3(nact), dim3(64*hs_combine_waves()), 0, st, dp, ch.active_begin);       // compute_aln_logprob
    if (mark()) return 1;
    HS_HIP(hipGetLastError());
    chunk_no++;
  }
  return 0;
}

int hipstr_hmm_profile(hipstr_dev_batch_t* dev, int enable){
  if (!dev) return fail("null device batch");
  dev->profiling = enable != 0;
  return 0;
}

// ms[4*i + {0,1,2,3}] = {leading flanks, STR block, trailing flanks, combine} of pass i (summed over chunks)
int hipstr_hmm_profile_read(hipstr_dev_batch_t* dev, float* ms, int cap){
  if (!dev || !ms) return -1;
  const size_t per_pass = 5 * dev->prep.chunks.size();
  if (per_pass == 0) return 0;
  int n = 0;
  for (size_t base = 0; base + per_pass <= dev->prof_used && n < cap; base += per_pass, n++){
    float acc[4] = {0, 0, 0, 0};
    for (size_t c = 0; c < dev->prep.chunks.size(); c++)
      for (int ph = 0; ph < 4; ph++){
        hipEvent_t a = dev->prof_pool[base + 5*c + ph], b = dev->prof_pool[base + 5*c + ph + 1];
        float t = 0;
        if (hipEventSynchronize(b) != hipSuccess || hipEventElapsedTime(&t, a, b) != hipSuccess) return -1;
        acc[ph] += t;
      }
    for (int ph = 0; ph < 4; ph++) ms[4*n + ph] = acc[ph];
  }
  dev->prof_used = 0;
  return n;
}

int hipstr_hmm_workload(hipstr_dev_batch_t* dev, int64_t* n_alignments, int64_t* algorithmic_bytes, int64_t* dp_cells){
  if (!dev) return fail("null device batch");
  if (n_alignments) *n_alignments = dev->prep.n_alignments;
  if (dev->algo_bytes == 0 && dev->dp_cells == 0) workload_of(dev);
  if (algorithmic_bytes) *algorithmic_bytes = dev->algo_bytes;
  if (dp_cells) *dp_cells = dev->dp_cells;
  return 0;
}

int hipstr_hmm_align_timed(hipstr_dev_batch_t* dev, int reps, float* ms_total, float* ms_kernel){
  if (!dev || reps < 1 || !ms_total) return fail("bad argument");
  if (bind(dev->ctx)) return 1;
  hipStream_t st = dev->stream;
  HS_HIP(hipEventRecord(dev->ev0, st));
  for (int i = 0; i < reps; i++) if (hipstr_hmm_align(dev, st)) return 1;
  HS_HIP(hipEventRecord(dev->ev1, st));
  HS_HIP(hipEventSynchronize(dev->ev1));
  HS_HIP(hipEventElapsedTime(ms_total, dev->ev0, dev->ev1));
  if (ms_kernel) *ms_kernel = *ms_total;
  return 0;
}

double* hipstr_hmm_dev_aln_probs(hipstr_dev_batch_t* dev){ return dev ? dev->h.aln_probs : NULL; }

int hipstr_hmm_fetch(hipstr_dev_batch_t* dev, double* aln_probs, int32_t* seeds){
  if (!dev || !aln_probs || !seeds) return fail("null argument");
  if (bind(dev->ctx)) return 1;
  const hipstr::Prepared& P = dev->prep;
  Ctx* ctx = dev->ctx;
  if (dev->foreign_stream) HS_HIP(hipDeviceSynchronize());         // a pass was launched on a stream of the caller's
  double* tmp = NULL;
  if (P.n_out){
    tmp = (double*)ctx->pin_cache.get((size_t)P.n_out*sizeof(double));
    if (!tmp) return 1;
    if (hipMemcpyAsync(tmp, dev->h.aln_probs, (size_t)P.n_out*sizeof(double), hipMemcpyDeviceToHost, dev->stream) != hipSuccess){
      ctx->pin_cache.put(tmp); return fail("hipMemcpyAsync (device to host) failed"); }
  }
  if (hipstr::wait_stream(dev->stream) != hipSuccess){ ctx->pin_cache.put(tmp); return fail("hipStreamSynchronize failed"); }
  // the reference's output contract, locus by locus (loci are independent: shared among the host threads)
  auto scatter = [&](int li){
    const hs_locus_t& loc = P.loci[li];
    const int A = loc.n_alleles;
    const bool all_haps = loc.n_re == A;
    for (int i = 0; i < loc.n_reads; i++){
      const int r = loc.read_begin + i;
      if (!P.realign_read[r]) continue;                       // HapAligner.cpp:326-329
      seeds[r] = P.seeds[r];
      double* dst = aln_probs + loc.out_off + (int64_t)i*A;
      const double* src = tmp + loc.out_off + (int64_t)i*A;
      if (P.seeds[r] == -1){ for (int k = 0; k < A; k++) dst[k] = 0; continue; }   // HapAligner.cpp:333-337
      if (all_haps) memcpy(dst, src, sizeof(double)*(size_t)A);
      else for (int k = 0; k < A; k++) if (P.realign_hap[loc.hap_begin + k]) dst[k] = src[k];   // HapAligner.cpp:615-619
    }
  };
  hipstr::parallel_for((int)P.loci.size(), P.n_out > (1 << 20) ? hipstr::host_threads() : 1, scatter);
  ctx->pin_cache.put(tmp);
  return 0;
}

int hipstr_hmm_process_reads(const hipstr_batch_t* batch, double* aln_probs, int32_t* seeds){
  return hipstr_hmm_process_reads_seeded(batch, NULL, aln_probs, seeds);
}

int hipstr_hmm_process_reads_seeded(const hipstr_batch_t* batch, const int32_t* seed_base, double* aln_probs, int32_t* seeds){
  hipstr::ApiTimer whole(hipstr::PB_PROCESS_READS);
  const bool prof = hipstr::api_profile_on();
  const double t0 = prof ? hipstr::ApiTimer::now() : 0;
  hipstr_dev_batch_t* dev = hipstr_hmm_upload_seeded(batch, seed_base);
  if (!dev) return 1;
  const double t1 = prof ? hipstr::ApiTimer::now() : 0;
  int rc = hipstr_hmm_align(dev, NULL);
  const double t2 = prof ? hipstr::ApiTimer::now() : 0;
  if (!rc) rc = hipstr_hmm_fetch(dev, aln_probs, seeds);
  const double t3 = prof ? hipstr::ApiTimer::now() : 0;
  if (prof){
    hipstr::api_profile_add(hipstr::PB_PR_PREPARE, dev->t_prepare); hipstr::api_profile_add(hipstr::PB_PR_STAGE, dev->t_stage);
    hipstr::api_profile_add(hipstr::PB_PR_UPLOAD_REST, (t1 - t0) - dev->t_prepare - dev->t_stage);
    hipstr::api_profile_add(hipstr::PB_PR_LAUNCH, t2 - t1); hipstr::api_profile_add(hipstr::PB_PR_FETCH, t3 - t2);
  }
  hipstr_hmm_free(dev);
  if (prof) hipstr::api_profile_add(hipstr::PB_PR_FREE, hipstr::ApiTimer::now() - t3);
  return rc;
}

// ---- hipstr_debug_api_profile
extern "C++" {
namespace hipstr {
static std::atomic<bool> g_prof_on(false);
static std::atomic<int64_t> g_prof_ns[PB_COUNT], g_prof_calls[PB_COUNT];
bool api_profile_on(){ return g_prof_on.load(std::memory_order_relaxed); }
void api_profile_add(int bucket, double seconds, int calls){
  g_prof_ns[bucket].fetch_add((int64_t)(seconds*1e9), std::memory_order_relaxed); g_prof_calls[bucket].fetch_add(calls, std::memory_order_relaxed);
}
double ApiTimer::now(){ return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}
}
#ifndef HIPSTR_NO_DEBUG_ABI      // diagnostics (include/hipstr_hmm_debug.h): tests, fuzzers, bench.py — not the drop-in ABI
int64_t hipstr_debug_driver_allocs(void){ return g_driver_allocs.load(); }
int hipstr_debug_cr_math(int which, const double* x, double* y, int64_t n){
  if (!x || !y || n < 0) return fail("null argument");
  if (n == 0) return 0;
  Ctx* c = current_ctx();
  if (!c || bind(c)) return 1;
  double* dx = (double*)c->dev_cache.get((size_t)n*8); double* dy = (double*)c->dev_cache.get((size_t)n*8);
  int rc = 1;
  if (dx && dy && hipMemcpy(dx, x, (size_t)n*8, hipMemcpyHostToDevice) == hipSuccess){
    const auto t0 = std::chrono::steady_clock::now();
    const int reps = getenv("HIPSTR_TIMING") ? 10 : 1;
    for (int r = 0; r < reps; r++)
      hipLaunchKernelGGL(hs_cr_math_kernel, dim3((unsigned)((n + 255)/256)), dim3(256), 0, c->stream, which, (const double*)dx, dy, n);
    if (reps > 1 && hipstr::wait_stream(c->stream) == hipSuccess)
      fprintf(stderr, "hipstr_debug_cr_math: %s of %lld arguments x %d launches: %.3f ms per launch\n", which ? "log" : "exp", (long long)n, reps,
              1e3*std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()/reps);
    if (hipstr::wait_stream(c->stream) == hipSuccess && hipMemcpy(y, dy, (size_t)n*8, hipMemcpyDeviceToHost) == hipSuccess) rc = 0;
  }
  c->dev_cache.put(dx); c->dev_cache.put(dy);
  return rc ? fail("hipstr_debug_cr_math: device call failed") : 0;
}
void* hipstr_debug_cache_get(int64_t bytes){ Ctx* c = current_ctx(); if (!c || bind(c) || bytes < 0) return NULL; return c->dev_cache.get((size_t)bytes); }
void hipstr_debug_cache_put(void* p){ Ctx* c = current_ctx(); if (c && p) c->dev_cache.put(p); }
int hipstr_debug_cache_stats(int64_t out[12]){
  Ctx* c = current_ctx();
  if (!c) return 1;
  BlockCache* bc[2] = { &c->dev_cache, &c->pin_cache };
  for (int k = 0; k < 2; k++){
    std::lock_guard<std::mutex> g(bc[k]->m);
    size_t held = 0; for (const BlockCache::Chunk& ch : bc[k]->chunks) held += ch.size;
    out[4*k] = (int64_t)held; out[4*k + 1] = (int64_t)bc[k]->cached; out[4*k + 2] = (int64_t)bc[k]->in_use; out[4*k + 3] = (int64_t)bc[k]->cap;
    out[8 + 2*k] = bc[k]->n_fallback; out[9 + 2*k] = bc[k]->n_trim_retry;
  }
  return 0;
}

// Diagnostics: how the (realigned allele, side) pairs of a batch split over the STR kernels — counts[1]: every visiting list tabulated
// (periodic blocks: hs_str_group_kernel_p), counts[2]: simple or piecewise-simple lists (one or two interruptions: hs_str_group_kernel_pw),
// counts[3]: at least one list replayed inside the grouped layout (three and more interruptions: hs_str_group_kernel_rp),
// counts[0]: anything else (hs_str_kernel_generic).
int hipstr_debug_allele_kinds(hipstr_dev_batch_t* dev, int64_t counts[4]){
  if (!dev || !counts) return fail("null argument");
  counts[0] = counts[1] = counts[2] = counts[3] = 0;
  const hipstr::Prepared& P = dev->prep;
  for (const hs_allele_t& al : P.alleles)
    if (al.realign) for (int side = 0; side < 2; side++){ const int k = P.stropts[al.str_opt[side]].kind; if (k >= 0 && k <= 3) counts[k]++; }
  return 0;
}

// Diagnostics (tests): a non-blocking HIP stream of the calling thread's context, as a caller of its own would create one, and its release.
void* hipstr_debug_stream_create(void){
  Ctx* ctx = hipstr::api_current_ctx();
  if (!ctx) return NULL;
  hipStream_t st = NULL;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess){ fail("hipStreamCreate failed"); return NULL; }
  return (void*)st;
}
void hipstr_debug_stream_destroy(void* st){ if (st){ hipStreamSynchronize((hipStream_t)st); hipStreamDestroy((hipStream_t)st); } }

// Diagnostics (tests/test_expand_gpu.py): the device's copy of a batch's option records (what = 0), f64 pool with its generated part (1)
// or per-allele records (2), after everything queued by the upload has run.  Returns the bytes the table has (-1 on failure); copies at most cap.
int64_t hipstr_debug_fetch_table(hipstr_dev_batch_t* dev, int what, void* buf, int64_t cap){
  if (!dev || bind(dev->ctx)) return -1;
  const hipstr::Prepared& P = dev->prep;
  const void* src = NULL; int64_t n = 0;
  if (what == 0){ src = dev->h.stropts; n = (int64_t)P.stropts.size()*sizeof(hs_stropt_t); }
  else if (what == 1){ src = dev->h.f64pool; n = (int64_t)(dev->h.f64_gen_base + P.gen_f64)*sizeof(double); }
  else if (what == 2){ src = dev->h.grp_recs; n = (int64_t)P.rec_descs.size()*HS_GRP_REC_DWORDS*sizeof(int32_t); }
  else { fail("unknown table"); return -1; }
  if (hipStreamSynchronize(dev->h2d_stream ? dev->h2d_stream : dev->stream) != hipSuccess){ fail("hipStreamSynchronize failed"); return -1; }
  if (dev->ev_expand && hipStreamSynchronize(dev->aux_stream) != hipSuccess){ fail("hipStreamSynchronize failed"); return -1; }
  const int64_t m = std::min(n, cap);
  if (m > 0 && buf && hipMemcpy(buf, src, (size_t)m, hipMemcpyDeviceToHost) != hipSuccess){ fail("hipMemcpy failed"); return -1; }
  return n;
}

int hipstr_debug_api_profile(int mode, int cap, const char** names, double* seconds, int64_t* calls){
  static const char* const kNames[hipstr::PB_COUNT] = {
    "hipstr_hmm_process_reads[_seeded]", "  prepare_batch", "  pack staging buffer", "  blocks + H2D enqueue", "  kernel launches", "  wait + D2H + scatter", "  release",
    "hipstr_hmm_trace[_seeded]", "  replay + string work (host)", "hipstr_post_run", "hipstr_post_extract", "hipstr_em_train", "hipstr_nw_align",
    "hipstr_stream_submit", "hipstr_stream_take", "hipstr_calc_seed_bases",
    "    upload: device + pinned blocks from the caches", "    upload: hipMemcpyAsync (tables, reads)", "    upload: expansion kernels + output memset", "    upload: events" };
  if (mode == 1){ for (int i = 0; i < hipstr::PB_COUNT; i++){ hipstr::g_prof_ns[i] = 0; hipstr::g_prof_calls[i] = 0; } hipstr::g_prof_on = true; }
  else if (mode == 0) hipstr::g_prof_on = false;
  for (int i = 0; i < hipstr::PB_COUNT && i < cap; i++){
    if (names) names[i] = kNames[i];
    if (seconds) seconds[i] = 1e-9*(double)hipstr::g_prof_ns[i].load();
    if (calls) calls[i] = hipstr::g_prof_calls[i].load();
  }
  return hipstr::PB_COUNT;
}

// Diagnostics for the host preparation (no device needed): the haplotype rows, as the flank sweeps
// consume them, of allele k / side / block (0 = leading flank, 1 = trailing flank) of a ONE-locus batch.
int hipstr_debug_rows(const hipstr_batch_t* batch, int k, int side, int which, uint32_t* rows, int cap){
  hipstr::Prepared P; std::string err;
  if (hipstr::prepare_batch(batch, P, err)){ g_err = err; return -1; }
  if (k < 0 || k >= (int)P.alleles.size() || side < 0 || side > 1) return -1;
  const hs_allele_t& al = P.alleles[k];
  if (!al.realign) return 0;
  const hs_rowset_t rs = P.rowsets[which ? al.trail_rows[side] : al.lead_rows[side]];
  for (int i = 0; i < rs.len && i < cap; i++) rows[i] = P.rows[rs.off + i];
  return rs.len;
}

// Diagnostics (host only): runs the host preparation of a batch with `threads` host threads (0 = default) and returns its wall time
// and a digest of everything it produced — the digest must not depend on the thread count.
int hipstr_debug_prepare(const hipstr_batch_t* batch, int threads, double* seconds, uint64_t* digest){
  hipstr::Prepared P; std::string err;
  hipstr::set_host_threads(threads);
  const auto t0 = std::chrono::steady_clock::now();
  const int rc = hipstr::prepare_batch(batch, P, err);
  const auto t1 = std::chrono::steady_clock::now();
  hipstr::set_host_threads(0);
  if (rc){ g_err = err; return 1; }
  if (seconds) *seconds = std::chrono::duration<double>(t1 - t0).count();
  hipstr::prep_profile_print();
  if (digest){
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void* p, size_t n){ const uint8_t* q = (const uint8_t*)p; for (size_t i = 0; i < n; i++){ h ^= q[i]; h *= 1099511628211ull; } };
#define HS_MIX(v) mix((v).data(), (v).size()*sizeof((v)[0]))
    HS_MIX(P.loci); HS_MIX(P.alleles); HS_MIX(P.stropts); HS_MIX(P.rowsets);
    HS_MIX(P.rows); for (const hipstr::Prepared& f : P.frags) HS_MIX(f.rows);
    HS_MIX(P.visits); for (const hipstr::Prepared& f : P.frags) HS_MIX(f.visits);
    HS_MIX(P.f64pool); for (const hipstr::Prepared& f : P.frags) HS_MIX(f.f64pool);
    HS_MIX(P.chars); for (const hipstr::Prepared& f : P.frags) HS_MIX(f.chars);
    HS_MIX(P.rec_descs); HS_MIX(P.pmf13);
    HS_MIX(P.reads); HS_MIX(P.active); HS_MIX(P.seeds); HS_MIX(P.realign_read); HS_MIX(P.realign_hap); HS_MIX(P.ws); HS_MIX(P.lead_items);
    HS_MIX(P.trail_items); HS_MIX(P.str_items); HS_MIX(P.tpack); HS_MIX(P.str_order); HS_MIX(P.tgroups); HS_MIX(P.tmembers); HS_MIX(P.chunks); HS_MIX(P.nd_rows);
#undef HS_MIX
    const int64_t tail[9] = { P.ws_mr_size, P.ws_lt_size, P.ws_lead_size, P.ws_col_size, P.max_side_len, P.max_B, P.n_out, P.n_alignments, P.gen_f64 };
    mix(tail, sizeof tail);
    *digest = h;
  }
  return 0;
}

int hipstr_debug_str_groups(const hipstr_batch_t* batch, int32_t* side, int32_t* columns, int32_t* read_off, int cap_groups,
                            int32_t* reads, int cap_reads, int32_t* max_columns){
  if (!batch || !side || !columns || !read_off || !reads) return -1;
  hipstr::Prepared P; std::string err;
  if (hipstr::prepare_batch(batch, P, err)){ g_err = err; return -1; }
  if (max_columns) *max_columns = HS_GRP_COLS;
  const int ng = (int)P.str_items.size();
  if (ng > cap_groups) return -1;
  int nr = 0;
  read_off[0] = 0;
  for (int g = 0; g < ng; g++){
    const hs_item_t& it = P.str_items[g];
    side[g] = it.side; columns[g] = it.rowset;
    if (nr + it.slot > cap_reads) return -1;
    for (int k = 0; k < it.slot; k++) reads[nr++] = P.active[P.tpack[it.active + k]];
    read_off[g + 1] = nr;
  }
  return ng;
}

int hipstr_debug_simple_table(int bound, int U0, int tail, double entry[3]){
  if (!entry || bound < 0 || U0 < 0 || tail < 0 || tail >= 10000 || U0 >= 10000) return fail("bad argument");
  hipstr::debug_simple_table(bound, U0, tail, entry);
  return 0;
}
#endif  // HIPSTR_NO_DEBUG_ABI

// ----------------------------------------------------------------------------- posteriors
int hipstr_post_offsets(const hipstr_post_batch_t* pb, int64_t* post_off, int64_t* samp_off){
  if (!pb || !post_off || !samp_off) return fail("null argument");
  int64_t po = 0, so = 0;
  for (int l = 0; l < pb->n_loci; l++){
    post_off[l] = po; samp_off[l] = so;
    po += (int64_t)pb->n_samples[l]*pb->n_alleles[l]*pb->n_alleles[l];
    so += pb->n_samples[l];
  }
  post_off[pb->n_loci] = po; samp_off[pb->n_loci] = so;
  return 0;
}

namespace {
struct PostRun {
  Ctx* ctx = NULL;
  hipStream_t stream = NULL;        // the creating thread's stream
  std::vector<hs_post_unit_t> units;
  std::vector<void*> allocs;        // device blocks from the context's cache
  hs_post_dev_t h;
  hs_post_dev_t* d_args = NULL;
  int64_t n_post = 0, n_samp = 0, n_ll = 0;
  int n_reads = 0;
  // small runs (a locus or a few: everything under 4 MiB): inputs and argument block travel in ONE pinned block with one copy, the three
  // results sit next to each other in the same device block and come back with one copy
  void* pin_in = NULL; size_t res_bytes = 0, res_total_off = 0, res_map_off = 0;
  hipEvent_t ev_up = NULL;          // small runs: recorded on `stream` behind the asynchronous upload; a launch on another stream waits for it
  ~PostRun(){
    if (ctx && ev_up) ctx->put_event(ev_up, false);
    if (!ctx || (allocs.empty() && !pin_in) || bind(ctx)) return;
    hipStreamSynchronize(stream);           // the blocks are handed to the next user
    for (void* p : allocs) ctx->dev_cache.put(p);
    if (pin_in) ctx->pin_cache.put(pin_in);
  }
};

int post_setup(const hipstr_post_batch_t* pb, const double* dev_ll, PostRun& R){
  const hipstr::HostTables& T = hipstr::host_tables();
  Ctx* ctx = R.ctx;
  int64_t po = 0, so = 0, lo = 0;
  for (int l = 0; l < pb->n_loci; l++){
    const int A = pb->n_alleles[l], S = pb->n_samples[l];
    const int r0 = pb->read_off[l], r1 = pb->read_off[l+1];
    if (A < 1 || A+1 >= 10000) return fail("allele count out of range");
    const bool hap = pb->haploid && pb->haploid[l];
    const double hom = hap ? -T.int_log[A] : T.int_log[2] - T.int_log[A] - T.int_log[A+1];     // genotyper.cpp:20-25
    const double het = hap ? -DBL_MAX/2 : -T.int_log[A] - T.int_log[A+1];                     // genotyper.cpp:27-32
    int r = r0;
    for (int s = 0; s < S; s++){
      hs_post_unit_t u;
      u.post_off = u.prior_off = po + (int64_t)s*A*A; u.n_alleles = A; u.samp_index = (int32_t)(so + s);
      u.log_hom_prior = hom; u.log_het_prior = het;
      u.read_begin = r; u.ll_off = lo + (int64_t)(r-r0)*A;
      while (r < r1 && pb->sample_label[r] == s) r++;
      u.n_reads = r - u.read_begin;
      R.units.push_back(u);
    }
    if (r != r1) return fail("reads of a locus must be grouped by ascending sample label (genotyper.h:112-119)");
    po += (int64_t)S*A*A; so += S; lo += (int64_t)(r1-r0)*A;
  }
  R.n_post = po; R.n_samp = so; R.n_ll = lo; R.n_reads = pb->n_loci ? pb->read_off[pb->n_loci] : 0;
  memset(&R.h, 0, sizeof R.h);
  if (!dev_ll && !pb->log_aln_probs) return fail("no log_aln_probs given");
  {
    struct Piece { const void* src; size_t bytes, off; };
    std::vector<Piece> in; size_t tot = 0;
    auto add = [&](const void* src, size_t bytes){ const size_t off = tot; in.push_back(Piece{src, bytes, off}); tot = (tot + (bytes ? bytes : 1) + 255) & ~(size_t)255; return off; };
    const size_t o_units = add(R.units.data(), R.units.size()*sizeof(hs_post_unit_t)), o_p1 = add(pb->log_p1, sizeof(double)*R.n_reads),
                 o_p2 = add(pb->log_p2, sizeof(double)*R.n_reads), o_w = add(pb->read_weight, sizeof(int32_t)*R.n_reads),
                 o_prior = pb->log_prior ? add(pb->log_prior, sizeof(double)*R.n_post) : 0,
                 o_ll = dev_ll ? 0 : add(pb->log_aln_probs, sizeof(double)*R.n_ll), o_args = add(&R.h, sizeof R.h);
    const size_t in_total = tot;
    const size_t o_post = tot; tot += ((size_t)sizeof(double)*R.n_post + 255) & ~(size_t)255;
    const size_t o_tot = tot;  tot += ((size_t)sizeof(double)*R.n_samp + 255) & ~(size_t)255;
    const size_t o_map = tot;  tot += ((size_t)sizeof(int32_t)*2*R.n_samp + 255) & ~(size_t)255;
    if (tot < ((size_t)4 << 20)){
      char* dblk = (char*)ctx->dev_cache.get(tot);
      if (!dblk) return 1;
      R.allocs.push_back(dblk);
      char* pin = (char*)ctx->pin_cache.get(in_total);
      if (!pin) return 1;
      R.pin_in = pin;
      R.h.units = (const hs_post_unit_t*)(dblk + o_units); R.h.log_p1 = (const double*)(dblk + o_p1); R.h.log_p2 = (const double*)(dblk + o_p2);
      R.h.read_weight = (const int32_t*)(dblk + o_w);
      if (pb->log_prior) R.h.log_prior = (const double*)(dblk + o_prior);
      R.h.log_aln_probs = dev_ll ? dev_ll : (const double*)(dblk + o_ll);
      R.h.log_post = (double*)(dblk + o_post); R.h.sample_total = (double*)(dblk + o_tot); R.h.map_gt = (int32_t*)(dblk + o_map);
      R.h.log_thresh = T.log_thresh; R.h.log_half = T.log_half;
      R.d_args = (hs_post_dev_t*)(dblk + o_args);
      for (const Piece& pc : in) if (pc.bytes && pc.src) memcpy(pin + pc.off, pc.src, pc.bytes);      // (the argument block last in the list: filled in by now)
      HS_HIP(hipMemcpyAsync(dblk, pin, in_total, hipMemcpyHostToDevice, R.stream));
      R.ev_up = ctx->get_event(false);
      if (!R.ev_up) return fail("hipEventCreate failed");
      HS_HIP(hipEventRecord(R.ev_up, R.stream));
      R.res_bytes = tot - o_post; R.res_total_off = o_tot - o_post; R.res_map_off = o_map - o_post;
      return 0;
    }
  }
  auto up = [&](const void* src, size_t bytes, void** out){
    *out = ctx->dev_cache.get(bytes ? bytes : 1);
    if (!*out) return 1;
    R.allocs.push_back(*out);
    if (src && bytes && hipMemcpyAsync(*out, src, bytes, hipMemcpyHostToDevice, R.stream) != hipSuccess) return fail("hipMemcpy failed");
    return 0;
  };
  void* p;
  if (up(R.units.data(), R.units.size()*sizeof(hs_post_unit_t), &p)) return 1; R.h.units = (const hs_post_unit_t*)p;
  if (up(pb->log_p1, sizeof(double)*R.n_reads, &p)) return 1; R.h.log_p1 = (const double*)p;
  if (up(pb->log_p2, sizeof(double)*R.n_reads, &p)) return 1; R.h.log_p2 = (const double*)p;
  if (up(pb->read_weight, sizeof(int32_t)*R.n_reads, &p)) return 1; R.h.read_weight = (const int32_t*)p;
  if (pb->log_prior){ if (up(pb->log_prior, sizeof(double)*R.n_post, &p)) return 1; R.h.log_prior = (const double*)p; }
  if (dev_ll) R.h.log_aln_probs = dev_ll;
  else {
    if (!pb->log_aln_probs) return fail("no log_aln_probs given");
    if (up(pb->log_aln_probs, sizeof(double)*R.n_ll, &p)) return 1; R.h.log_aln_probs = (const double*)p;
  }
  if (up(NULL, sizeof(double)*R.n_post, &p)) return 1; R.h.log_post = (double*)p;
  if (up(NULL, sizeof(double)*R.n_samp, &p)) return 1; R.h.sample_total = (double*)p;
  if (up(NULL, sizeof(int32_t)*2*R.n_samp, &p)) return 1; R.h.map_gt = (int32_t*)p;
  R.h.log_thresh = T.log_thresh; R.h.log_half = T.log_half;
  if (up(&R.h, sizeof R.h, &p)) return 1; R.d_args = (hs_post_dev_t*)p;
  HS_HIP(hipstr::wait_stream(R.stream));       // the sources are the caller's (pageable) arrays: done with them before returning
  return 0;
}
}  // namespace

struct hipstr_post_dev { PostRun R; std::vector<int32_t> n_samples, n_alleles; std::vector<uint8_t> haploid; bool foreign_stream = false; };

hipstr_post_dev_t* hipstr_post_upload(const hipstr_post_batch_t* pb, const double* dev_log_aln_probs){
  if (!pb){ g_err = "null argument"; return NULL; }
  Ctx* ctx = hipstr::api_current_ctx();
  if (!ctx) return NULL;
  hipstr_post_dev_t* pd = new hipstr_post_dev_t();
  pd->R.ctx = ctx;
  pd->R.stream = thread_stream(ctx);
  if (post_setup(pb, dev_log_aln_probs, pd->R)){ delete pd; return NULL; }
  pd->n_samples.assign(pb->n_samples, pb->n_samples + pb->n_loci);
  pd->n_alleles.assign(pb->n_alleles, pb->n_alleles + pb->n_loci);
  pd->haploid.assign(pb->n_loci, 0);
  if (pb->haploid) pd->haploid.assign(pb->haploid, pb->haploid + pb->n_loci);
  return pd;
}

int hipstr_post_launch(hipstr_post_dev_t* pd, void* hip_stream){
  if (!pd) return fail("null argument");
  if (pd->R.units.empty()) return 0;
  if (bind(pd->R.ctx)) return 1;
  hipStream_t st = hip_stream ? (hipStream_t)hip_stream : pd->R.stream;
  if (hip_stream && st != pd->R.stream) pd->foreign_stream = true;
  // a small run's inputs and argument block were sent asynchronously on the run's own stream: a launch elsewhere comes after them
  if (st != pd->R.stream && pd->R.ev_up) HS_HIP(hipStreamWaitEvent(st, pd->R.ev_up, 0));
  // HIPSTR_DEBUG_HOST_LIBM=1 (tests/test_genotypes_gpu.py): the three places where the device's exp / log enter — the per-sample
  // log-sum-exp over the diplotypes, the streaming log-sum-exps per genotype and the exact pair log-sum-exp of the unphased posterior —
  // are evaluated on the host with its libm, in the reference's order, on the device's accumulated values.  It shows where the
  // one-float-step differences of GL / GLDIFF / PL against the reference come from (they vanish); it is not a way to run the path.
  const bool host_libm = getenv("HIPSTR_DEBUG_HOST_LIBM") && atoi(getenv("HIPSTR_DEBUG_HOST_LIBM")) != 0;
  if (host_libm != (pd->R.h.raw != 0)){
    pd->R.h.raw = host_libm ? 1 : 0;
    HS_HIP(hipMemcpy(pd->R.d_args, &pd->R.h, sizeof pd->R.h, hipMemcpyHostToDevice));
  }
  {
    // few units with many diplotypes each (one sample per locus, 128 haplotypes): the accumulation of a unit is shared by several
    // workgroups, to ~2048 in all (post_kernels.hip; bit-identical either way)
    const size_t n_units = pd->R.units.size();
    int max_nd = 1;
    for (const hs_post_unit_t& u : pd->R.units) max_nd = std::max(max_nd, u.n_alleles*u.n_alleles);
    const int split = (int)std::min<size_t>((size_t)(max_nd + 255)/256, n_units < 1024 ? (2048 + n_units - 1)/n_units : 1);
    constexpr bool no_split = false;
    if (split > 1 && !no_split){
      hipLaunchKernelGGL(hs_posterior_accumulate_kernel, dim3((unsigned)n_units, (unsigned)split), dim3(256), 0, st, (const hs_post_dev_t*)pd->R.d_args);
      hipLaunchKernelGGL(hs_posterior_finish_kernel, dim3((unsigned)n_units), dim3(256), 0, st, (const hs_post_dev_t*)pd->R.d_args);
    } else
      hipLaunchKernelGGL(hs_posterior_kernel, dim3((unsigned)n_units), dim3(256), 0, st, (const hs_post_dev_t*)pd->R.d_args);
  }
  HS_HIP(hipGetLastError());
  if (host_libm){
    PostRun& R = pd->R;
    HS_HIP(hipstr::wait_stream(st));
    std::vector<double> post((size_t)R.n_post), total((size_t)R.n_samp, 0.0);
    std::vector<int32_t> mapgt(2*(size_t)R.n_samp, -1);
    HS_HIP(hipMemcpy(post.data(), R.h.log_post, sizeof(double)*R.n_post, hipMemcpyDeviceToHost));
    for (const hs_post_unit_t& u : R.units){
      double* v = post.data() + u.post_off;
      const int nd = u.n_alleles*u.n_alleles;
      double mx = v[0];                                   // log_sum_exp (mathops.cpp:44-50)
      for (int i = 1; i < nd; i++) mx = std::max(mx, v[i]);
      double tot = 0.0;
      for (int i = 0; i < nd; i++) tot += exp(v[i] - mx);
      const double lse = mx + log(tot);
      double bv = -1.7976931348623157e308; int bi = -1;   // first maximum, a1-major (genotyper.cpp:88-95)
      for (int i = 0; i < nd; i++){ v[i] -= lse; if (v[i] > bv){ bv = v[i]; bi = i; } }
      total[u.samp_index] = lse;
      if (bi >= 0){ mapgt[2*u.samp_index] = bi / u.n_alleles; mapgt[2*u.samp_index+1] = bi % u.n_alleles; }
    }
    HS_HIP(hipMemcpy(R.h.log_post, post.data(), sizeof(double)*R.n_post, hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(R.h.sample_total, total.data(), sizeof(double)*R.n_samp, hipMemcpyHostToDevice));
    HS_HIP(hipMemcpy(R.h.map_gt, mapgt.data(), sizeof(int32_t)*2*R.n_samp, hipMemcpyHostToDevice));
  }
  return 0;
}

int hipstr_post_fetch(hipstr_post_dev_t* pd, double* log_post, double* sample_total_ll, int32_t* map_gt, double* locus_total_ll){
  if (!pd || !log_post || !sample_total_ll || !map_gt || !locus_total_ll) return fail("null argument");
  PostRun& R = pd->R;
  if (bind(R.ctx)) return 1;
  if (pd->foreign_stream) HS_HIP(hipDeviceSynchronize());
  if (!(R.res_bytes && !pd->foreign_stream)) HS_HIP(hipstr::wait_stream(R.stream));      // (the small form's one copy is ordered behind the kernel on the same stream)
  if (!R.units.empty() && R.res_bytes){
    char* pin = (char*)R.ctx->pin_cache.get(R.res_bytes);
    if (!pin) return 1;
    if (hipMemcpyAsync(pin, R.h.log_post, R.res_bytes, hipMemcpyDeviceToHost, R.stream) != hipSuccess || hipstr::wait_stream(R.stream) != hipSuccess){
      R.ctx->pin_cache.put(pin); return fail("device-to-host copy failed"); }
    memcpy(log_post, pin, sizeof(double)*R.n_post); memcpy(sample_total_ll, pin + R.res_total_off, sizeof(double)*R.n_samp);
    memcpy(map_gt, pin + R.res_map_off, sizeof(int32_t)*2*R.n_samp);
    R.ctx->pin_cache.put(pin);
  } else if (!R.units.empty()){
    if (fetch_array(R.ctx, R.stream, log_post, R.h.log_post, sizeof(double)*R.n_post) ||
        fetch_array(R.ctx, R.stream, sample_total_ll, R.h.sample_total, sizeof(double)*R.n_samp) ||
        fetch_array(R.ctx, R.stream, map_gt, R.h.map_gt, sizeof(int32_t)*2*R.n_samp)) return 1;
  }
  int64_t so = 0;
  for (size_t l = 0; l < pd->n_samples.size(); l++){     // sum(sample_total_LLs_) in sample order (genotyper.cpp:75)
    double tot = 0.0;
    for (int s = 0; s < pd->n_samples[l]; s++) tot += sample_total_ll[so+s];
    locus_total_ll[l] = tot;
    so += pd->n_samples[l];
  }
  return 0;
}

void hipstr_post_free(hipstr_post_dev_t* pd){ delete pd; }

int hipstr_gt_offsets(const hipstr_post_batch_t* pb, const hipstr_gt_request_t* rq, int64_t* gl_off, int64_t* pgl_off){
  if (!pb || !rq || !gl_off || !pgl_off) return fail("null argument");
  int64_t g = 0, pg = 0, so = 0;
  for (int l = 0; l < pb->n_loci; l++){
    const int64_t V = rq->n_variants[l];
    const bool hap = pb->haploid && pb->haploid[l];
    for (int s = 0; s < pb->n_samples[l]; s++, so++){
      gl_off[so] = g; pgl_off[so] = pg;
      g += hap ? V : V*(V+1)/2; pg += hap ? V : V*V;
    }
  }
  gl_off[so] = g; pgl_off[so] = pg;
  return 0;
}

int hipstr_post_extract(hipstr_post_dev_t* pd, const hipstr_gt_request_t* rq, hipstr_gt_out_t* out){
  hipstr::ApiTimer prof_t(hipstr::PB_POST_EXTRACT);
  if (!pd || !rq || !out || !rq->n_variants || !rq->hap_to_allele) return fail("null argument");
  if (!out->best_hap || !out->best_gt || !out->log_phased_post || !out->log_unphased_post || !out->hap_log_phased_post || !out->hap_log_unphased_post)
    return fail("null output array");
  const bool any = rq->calc_gls || rq->calc_pls || rq->calc_phased_gls;
  if (any && !out->gl_diff) return fail("gl_diff output missing");
  if ((rq->calc_gls && !out->gls) || (rq->calc_pls && !out->pls) || (rq->calc_phased_gls && !out->phased_gls)) return fail("requested output array missing");
  PostRun& R = pd->R;
  const hipstr::HostTables& T = hipstr::host_tables();
  const size_t n_loci = pd->n_samples.size();
  const bool timing = getenv("HIPSTR_TIMING") != NULL;
  auto t_lap = std::chrono::steady_clock::now();
  auto lap = [&](const char* what){ if (timing){ const auto t2 = std::chrono::steady_clock::now(); fprintf(stderr, "hipstr_post_extract: %-10s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(t2 - t_lap).count()); t_lap = t2; } };
  std::vector<hs_gt_unit_t> units;
  std::vector<int32_t> gmem, goff;
  int64_t po = 0, tot = 0, g = 0, pg = 0; int so = 0, map_off = 0;
  for (size_t l = 0; l < n_loci; l++){
    const int A = pd->n_alleles[l], S = pd->n_samples[l], V = rq->n_variants[l];
    const bool hap = pd->haploid[l] != 0;
    if (V < 1 || V > A) return fail("n_variants must be in [1, n_alleles]");
    const int32_t* h2a = rq->hap_to_allele + map_off;
    std::vector<int> count(V, 0);
    for (int a = 0; a < A; a++){ if (h2a[a] < 0 || h2a[a] >= V) return fail("hap_to_allele entry out of range"); count[h2a[a]]++; }
    const int goff_off = (int)goff.size();
    int run = 0;
    for (int v = 0; v < V; v++){ if (count[v] == 0) return fail("every variant must be carried by a haplotype"); goff.push_back(run); run += count[v]; }
    goff.push_back(run);
    for (int v = 0; v < V; v++) for (int a = 0; a < A; a++) if (h2a[a] == v) gmem.push_back(a);
    const double hom = hap ? -T.int_log[A] : T.int_log[2] - T.int_log[A] - T.int_log[A+1];
    const double het = hap ? 0.0 : -T.int_log[A] - T.int_log[A+1];                           // genotyper.cpp:198
    const double gl_ncfg  = hap ? T.int_log[2] + T.int_log[A] - T.int_log[V] : T.int_log[2] + 2*(T.int_log[A] - T.int_log[V]);
    const double pgl_ncfg = hap ? T.int_log[A] - T.int_log[V] : 2*(T.int_log[A] - T.int_log[V]);
    for (int s = 0; s < S; s++, so++){
      hs_gt_unit_t u; memset(&u, 0, sizeof u);
      u.post_off = po + (int64_t)s*A*A; u.tot_off = tot; tot += (int64_t)V*V;
      u.gl_off = g; u.pgl_off = pg; g += hap ? V : (int64_t)V*(V+1)/2; pg += hap ? V : (int64_t)V*V;
      u.n_alleles = A; u.n_variants = V; u.samp_index = so; u.haploid = hap ? 1 : 0;
      u.map_off = map_off; u.goff_off = goff_off;
      u.hom_corr = hom; u.het_corr = het; u.gl_ncfg = gl_ncfg; u.pgl_ncfg = pgl_ncfg;
      units.push_back(u);
    }
    po += (int64_t)S*A*A; map_off += A;
  }
  if (units.empty()) return 0;
  lap("units");
  if (bind(R.ctx)) return 1;
  Ctx* ctx = R.ctx;
  std::vector<void*> tmp;
  struct Free { Ctx* c; hipStream_t st; std::vector<void*>& v; ~Free(){ hipStreamSynchronize(st); for (void* p : v) c->dev_cache.put(p); } } guard{ctx, R.stream, tmp};
  auto dalloc = [&](size_t bytes, void** outp) -> int { *outp = ctx->dev_cache.get(bytes ? bytes : 1); if (!*outp) return 1; tmp.push_back(*outp); return 0; };
  hs_gt_dev_t h; memset(&h, 0, sizeof h);
  void* p;
  if (dalloc(units.size()*sizeof(hs_gt_unit_t), &p)) return 1; HS_HIP(hipMemcpy(p, units.data(), units.size()*sizeof(hs_gt_unit_t), hipMemcpyHostToDevice)); h.units = (const hs_gt_unit_t*)p;
  if (dalloc((size_t)map_off*4, &p)) return 1; HS_HIP(hipMemcpy(p, rq->hap_to_allele, (size_t)map_off*4, hipMemcpyHostToDevice)); h.h2a = (const int32_t*)p;
  if (dalloc(gmem.size()*4, &p)) return 1; HS_HIP(hipMemcpy(p, gmem.data(), gmem.size()*4, hipMemcpyHostToDevice)); h.gmem = (const int32_t*)p;
  if (dalloc(goff.size()*4, &p)) return 1; HS_HIP(hipMemcpy(p, goff.data(), goff.size()*4, hipMemcpyHostToDevice)); h.goff = (const int32_t*)p;
  h.log_post = R.h.log_post; h.sample_total = R.h.sample_total; h.map_gt = R.h.map_gt;
  if (dalloc((size_t)tot*8, &p)) return 1; h.tot = (double*)p;
  if (dalloc((size_t)so*2*4, &p)) return 1; h.best_gt = (int32_t*)p;
  double* five = NULL;
  if (dalloc((size_t)so*5*8, &p)) return 1; five = (double*)p;
  h.log_phased = five; h.log_unphased = five + so; h.hap_log_phased = five + 2*(size_t)so; h.hap_log_unphased = five + 3*(size_t)so; h.gl_diff = five + 4*(size_t)so;
  if (any){ if (dalloc((size_t)g*8, &p)) return 1; h.gls = (double*)p; }
  if (rq->calc_pls){ if (dalloc((size_t)g*4, &p)) return 1; h.pls = (int32_t*)p; }
  if (rq->calc_phased_gls){ if (dalloc((size_t)pg*8, &p)) return 1; h.pgls = (double*)p; }
  h.calc_any = any; h.calc_gls = rq->calc_gls; h.calc_pls = rq->calc_pls; h.calc_pgls = rq->calc_phased_gls;
  h.log_thresh = T.log_thresh;
  if (dalloc(sizeof h, &p)) return 1; HS_HIP(hipMemcpy(p, &h, sizeof h, hipMemcpyHostToDevice));
  const bool host_libm = getenv("HIPSTR_DEBUG_HOST_LIBM") && atoi(getenv("HIPSTR_DEBUG_HOST_LIBM")) != 0;      // see hipstr_post_launch
  std::vector<double> h_tot;
  if (host_libm){
    HS_HIP(hipDeviceSynchronize());
    std::vector<double> post((size_t)R.n_post);
    HS_HIP(hipMemcpy(post.data(), R.h.log_post, sizeof(double)*R.n_post, hipMemcpyDeviceToHost));
    h_tot.resize((size_t)tot);
    for (const hs_gt_unit_t& u : units){                  // total_log_phased_posteriors, streamed as genotyper.cpp:148-170 does (mathops.cpp:72-80)
      const int A = u.n_alleles, V = u.n_variants;
      const int32_t* gm = gmem.data() + u.map_off; const int32_t* go = goff.data() + u.goff_off;
      for (int v1 = 0; v1 < V; v1++) for (int v2 = 0; v2 < V; v2++){
        double mx = -1.7976931348623157e308/2, t = 0.0;
        for (int x = go[v1]; x < go[v1+1]; x++) for (int y = go[v2]; y < go[v2+1]; y++){
          const double lv = post[u.post_off + (int64_t)gm[x]*A + gm[y]];
          if (lv <= mx) t += exp(lv - mx); else { t *= exp(mx - lv); t += 1.0; mx = lv; }
        }
        h_tot[u.tot_off + (int64_t)v1*V + v2] = mx + log(t);
      }
    }
    HS_HIP(hipMemcpy(h.tot, h_tot.data(), sizeof(double)*(size_t)tot, hipMemcpyHostToDevice));
    h.tot_given = 1;
    HS_HIP(hipMemcpy(p, &h, sizeof h, hipMemcpyHostToDevice));
  }
  if (pd->foreign_stream) HS_HIP(hipDeviceSynchronize());          // the posterior kernel may still be running on a stream of the caller's
  hipLaunchKernelGGL(hs_genotype_kernel, dim3((unsigned)units.size()), dim3(256), 0, R.stream, (const hs_gt_dev_t*)p);
  HS_HIP(hipGetLastError());
  lap("setup");
  HS_HIP(hipstr::wait_stream(R.stream));
  lap("kernel");
  HS_HIP(hipMemcpy(out->best_hap, R.h.map_gt, (size_t)so*2*4, hipMemcpyDeviceToHost));
  HS_HIP(hipMemcpy(out->best_gt, h.best_gt, (size_t)so*2*4, hipMemcpyDeviceToHost));
  HS_HIP(hipMemcpy(out->log_phased_post, h.log_phased, (size_t)so*8, hipMemcpyDeviceToHost));
  HS_HIP(hipMemcpy(out->log_unphased_post, h.log_unphased, (size_t)so*8, hipMemcpyDeviceToHost));
  if (host_libm)                                          // the exact pair log_sum_exp of the two phasings (mathops.cpp:52-57) with the host libm
    for (const hs_gt_unit_t& u : units){
      const int ga = out->best_gt[2*u.samp_index], gb = out->best_gt[2*u.samp_index+1];
      if (ga < 0 || gb < 0 || ga == gb) continue;
      const double lp = h_tot[u.tot_off + (int64_t)u.n_variants*ga + gb], alt = h_tot[u.tot_off + (int64_t)u.n_variants*gb + ga];
      out->log_unphased_post[u.samp_index] = (lp > alt) ? lp + log(1 + exp(alt - lp)) : alt + log(1 + exp(lp - alt));
    }
  HS_HIP(hipMemcpy(out->hap_log_phased_post, h.hap_log_phased, (size_t)so*8, hipMemcpyDeviceToHost));
  HS_HIP(hipMemcpy(out->hap_log_unphased_post, h.hap_log_unphased, (size_t)so*8, hipMemcpyDeviceToHost));
  if (any) HS_HIP(hipMemcpy(out->gl_diff, h.gl_diff, (size_t)so*8, hipMemcpyDeviceToHost));
  if (rq->calc_gls && fetch_array(ctx, R.stream, out->gls, h.gls, (size_t)g*8)) return 1;
  if (rq->calc_pls && fetch_array(ctx, R.stream, out->pls, h.pls, (size_t)g*4)) return 1;
  if (rq->calc_phased_gls && fetch_array(ctx, R.stream, out->phased_gls, h.pgls, (size_t)pg*8)) return 1;
  lap("fetch");
  return 0;
}

int hipstr_post_run(const hipstr_post_batch_t* pb, const double* dev_log_aln_probs,
                    double* log_post, double* sample_total_ll, int32_t* map_gt, double* locus_total_ll){
  hipstr::ApiTimer prof_t(hipstr::PB_POST_RUN);
  if (!pb || !log_post || !sample_total_ll || !map_gt || !locus_total_ll) return fail("null argument");
  hipstr_post_dev_t* pd = hipstr_post_upload(pb, dev_log_aln_probs);
  if (!pd) return 1;
  int rc = hipstr_post_launch(pd, NULL);
  if (!rc) rc = hipstr_post_fetch(pd, log_post, sample_total_ll, map_gt, locus_total_ll);
  hipstr_post_free(pd);
  return rc;
}

}  // extern "C"
