/* hipstr_hmm_debug.h — diagnostics entry points of libhipstr_hmm.so (round 6: split from hipstr_hmm.h).
 *
 * NOT part of the drop-in boundary: nothing on the reference side binds these.  They exist for the test suite (tests/), the fuzzers
 * (tools/fuzz_*.py) and the measurement harness (bench.py) — host-only views of the preparation, the block caches' counters, the
 * device's copies of tables, a correctly-rounded-math probe.  A deployment build compiles the library with -DHIPSTR_NO_DEBUG_ABI and
 * none of them is defined or exported (tests/test_prep.py::test_library_exports_every_declared_symbol checks the default build, which
 * has them). */
#ifndef HIPSTR_HMM_DEBUG_H
#define HIPSTR_HMM_DEBUG_H
#include "hipstr_hmm.h"
#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* Diagnostics (host only, no device): the haplotype rows of allele k of a ONE-locus batch as the
 * device sweep consumes them — side 0 = forward/left problem, 1 = reversed/right problem; which 0 =
 * leading flank block, 1 = trailing flank block.  Row encoding: bits 0-7 base, 8-11 homopolymer index
 * min(15, ..) of HapAligner.cpp:119-120, 12-23 compact row index, bit 31 valid.  Returns the row
 * count (0 if the allele is not realigned, -1 on error).  Used by tests/test_prep.py. */
int hipstr_debug_rows(const hipstr_batch_t* batch, int k, int side, int which, uint32_t* rows, int cap);

/* Diagnostics (host only): the host preparation of a batch (flatten, reuse replay, visiting lists, closed-form tables, launch plan
 * — what hipstr_hmm_upload does before any byte moves) run with `threads` host threads (0 = the library default,
 * HIPSTR_HOST_THREADS or min(hardware threads, 32)); *seconds = its wall time, *digest = a hash of everything it produced, which
 * must not depend on the thread count.  Used by tests/test_prep.py and bench.py. */
int hipstr_debug_prepare(const hipstr_batch_t* batch, int threads, double* seconds, uint64_t* digest);

/* Diagnostics (host only): the STR-block groups of the launch plan — reads of one locus and side whose columns are laid end to end over
 * one workgroup's lanes (hs_str_group_kernel).  Group g: side[g], reads[read_off[g] .. read_off[g+1]) (read indices of the batch),
 * columns[g] = the sum of their side lengths.  Returns the number of groups (-1 on error or if a capacity is too small); *max_columns =
 * lanes of a workgroup.  Used by tests/test_prep.py. */
int hipstr_debug_str_groups(const hipstr_batch_t* batch, int32_t* side, int32_t* columns, int32_t* read_off, int cap_groups,
                            int32_t* reads, int cap_reads, int32_t* max_columns);

/* Diagnostics (host only): one entry {A, G, Bnd} of the tabulated closed form the STR kernel uses for a "simple" visiting
 * list (StutterAlignerClass.cpp:59-150 for a periodic block): with `bound` columns of the block in reach, a run of U0 equal
 * configurations at the block's right end and `tail` configurations in total, fast_log_sum_exp over the pushed values is
 * (lp0 + A) + G bit for bit whenever |lp0| < Bnd.  Used by tests/test_prep.py to check exactly that against the oracle. */
int hipstr_debug_simple_table(int bound, int U0, int tail, double entry[3]);

/* Diagnostics: where the calling process' host time inside the library goes.  mode 1 = reset and start, 0 = stop, anything else =
 * read only.  Fills up to `cap` entries of names / seconds (wall clock, summed over threads) / calls and returns the number of
 * buckets; names indented by two spaces are parts of the entry point above them.  Used by integration/genotype_flow.cpp --profile. */
int hipstr_debug_api_profile(int mode, int cap, const char** names, double* seconds, int64_t* calls);
/* Diagnostics: how many blocks the library has taken from the driver so far (hipMalloc / hipHostMalloc: misses of its block caches, 0.1 ms to
 * 1 s each).  A stream is in its steady state once this stops growing from pass to pass. */
int64_t hipstr_debug_driver_allocs(void);
/* Diagnostics (tests): the calling thread's device context's block caches — out[0..3] device: bytes held from the driver, bytes in free
 * blocks, blocks in use, the cap on idle bytes (HIPSTR_DEV_CACHE_GIB, default 70 % of the device's memory); out[4..7] the same for pinned
 * host memory (HIPSTR_PIN_CACHE_GIB, default 24).  A release that leaves more idle bytes than the cap gives the chunks without a block in
 * use back to the driver; when the driver refuses a new chunk a request is served from any free block that is large enough, then after
 * trimming idle chunks; only then does it fail.  out[8], out[9]: driver refusals the device cache survived by the first / the second way;
 * out[10], out[11]: the pinned cache's. */
int hipstr_debug_cache_stats(int64_t out[12]);
/* Diagnostics (tests): one block from / back to the calling thread's device block cache — what every upload does dozens of times. */
void* hipstr_debug_cache_get(int64_t bytes);
void hipstr_debug_cache_put(void* block);
/* Diagnostics (tests): the correctly rounded exp (which = 0) / log (1) of hipstr_amd/csrc/cr_math.h evaluated ON THE DEVICE, element by
 * element — the functions the posterior, genotype and EM kernels use in place of the device's own exp / log so that they reproduce
 * the host libm's bits. */
int hipstr_debug_cr_math(int which, const double* x, double* y, int64_t n);
/* Diagnostics: (realigned allele, side) pairs of a batch by the STR kernel that takes them: counts[1] periodic blocks (tabulated closed form),
 * counts[2] blocks with one or two interruptions (piecewise closed form), counts[3] more interruptions (lists replayed in the grouped layout),
 * counts[0] the rest (per-read kernel). */
int hipstr_debug_allele_kinds(hipstr_dev_batch_t* dev, int64_t counts[4]);
/* Diagnostics (tests): a non-blocking HIP stream made by the library's own HIP runtime — what a caller passes as `hip_stream` — and its release. */
void* hipstr_debug_stream_create(void);
void hipstr_debug_stream_destroy(void* hip_stream);
/* Diagnostics: the device's copy of a batch's STR-option records (what = 0: hs_stropt_t of hipstr_amd/csrc/layout.h), its f64 pool incl. the
 * part the device generated (1) or the per-allele records the device assembled (2).  Returns the table's size in bytes (-1 on failure). */
int64_t hipstr_debug_fetch_table(hipstr_dev_batch_t* dev, int what, void* buf, int64_t cap_bytes);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
