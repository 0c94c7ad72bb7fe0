// prep.h — host-side preparation of a hipstr_batch_t into the flat HBM layout of layout.h.
#pragma once
#include <stdint.h>
#include <functional>
#include <string>
#include <vector>

#include "../../include/hipstr_hmm.h"
#include "layout.h"

namespace hipstr {

// Host-side work sharing: the number of threads the library may use for host work of one call (HIPSTR_HOST_THREADS, default
// min(hardware threads, 32)) and a loop that hands indices [0, n) to up to max_threads threads through a counter.
int host_threads();
void set_host_threads(int n);      // 0 = back to the default
void set_thread_budget(int n);     // the calling thread's own share (what host_threads() returns on it); 0 = none: a stream's workers split the host threads among themselves
void parallel_for(int n, int max_threads, const std::function<void(int)>& fn);

// Constant tables the reference keeps in globals; computed once with the host libm so they
// carry exactly the bits the reference computes (mathops.cpp:13-21, AlignmentModel.cpp:20-32,
// base_quality.h:29-38 and its clamps :44-75, mathops.h:36, mathops.cpp:9).
struct HostTables {
  std::vector<double> int_log;        // [10000]
  std::vector<double> qual_correct;   // [256] by raw quality char
  std::vector<double> qual_error;     // [256]
  double m2m[16], m2i[16];
  double log_thresh, log_half;
};
const HostTables& host_tables();

struct Prepared {
  std::vector<hs_locus_t>  loci;
  std::vector<hs_allele_t> alleles;
  std::vector<hs_stropt_t> stropts;
  std::vector<hs_rowset_t> rowsets;
  std::vector<hs_row_t>    rows;
  std::vector<hs_visit_t>  visits;
  std::vector<double>      f64pool;
  std::vector<char>        chars;
  std::vector<hs_read_t>   reads;
  std::vector<int32_t>     active;
  std::vector<int32_t>     seeds;          // per read (valid only where realign_read)
  std::vector<uint8_t>     realign_read;   // per read
  std::vector<uint8_t>     realign_hap;    // per global allele
  // ---- launch plan: the batch is cut into chunks of consecutive active reads whose workspaces fit the budget
  struct Chunk {
    int32_t active_begin, active_end;      // range in `active`
    int32_t lead_begin, lead_end;          // lead_items range
    int32_t trail_begin, trail_end;        // trail_items range
    int32_t str_begin, str_end;            // str_items range
    int32_t n_long_sides;                  // read sides with more than HS_GRP_COLS columns (hs_str_kernel takes them)
    int64_t n_alignments;
  };
  std::vector<Chunk>      chunks;
  std::vector<hs_ws_t>    ws;              // per active read, offsets inside its chunk's workspaces
  std::vector<hs_item_t>  lead_items;      // (first entry in tpack, side | slot << 1, rowset, number of reads): leading flank of one distinct
                                           // flank for up to 64 reads of one locus and side
  std::vector<hs_item_t>  trail_items;     // (first entry in tpack, side, number of packed reads, group): trailing flank of one
                                           // allele group for up to 64/npad reads of one locus and side
  std::vector<hs_item_t>  str_items;       // (first entry in tpack, side, columns, number of reads): STR block of the tabulated alleles for a
                                           // group of reads of one locus and side whose columns fill a workgroup (hs_str_group_kernel)
  int32_t                 grp_nd_cap = 0;  // largest (reads x 36 period) of the str_items
  std::vector<int32_t>    tpack;           // active-read indices of the packed reads
  std::vector<int32_t>    str_order;       // per locus and side: realigned alleles sorted so that nested STR blocks follow each other
 std::vector<hs_ndrow_t>  nd_rows;         // row descriptors of the read-end deletion sums per locus side (layout.h); a small pool
  std::vector<hs_recdesc_t> rec_descs;     // one per tabulated position of a side's order (layout.h): what the device assembles a grp_recs[] record from
  std::vector<double>     pmf13;           // 13 doubles per locus: log_stutter_pmf of the artifact sizes (what the device-generated constants start from)
  int64_t                 gen_f64 = 0;     // doubles of the f64 pool the device fills in (constants + tables of the options marked hs_stropt_t::gen)
  std::vector<hs_tgroup_t> tgroups;
  std::vector<int32_t>    lead_off, lead_ids;   // host only: the distinct leading-flank rowsets of locus l, side s, by slot = lead_ids[lead_off[2 l + s] ...], hs_locus_t::n_lead[s] of them
  std::vector<int32_t>    tmembers;
  int64_t ws_mr_size = 0, ws_lt_size = 0, ws_lead_size = 0, ws_col_size = 0, ws_nd_size = 0;   // doubles, max over chunks
  int32_t max_side_len = 0;
  int32_t max_B = 1;          // longest STR allele
  int64_t n_out        = 0;
  int64_t n_alignments = 0;   // (active read) x (realigned allele) pairs = HMM alignments per pass
  int32_t max_read_len = 0;
  int32_t max_flank    = 0;   // max n_flank over alleles
  // Threaded preparation (prep.cpp): the four LARGE pools — rows, visits, f64pool, chars — are not merged on the host; they stay in
  // the fragments that built them (offsets inside the small pools are already batch-wide) and the upload gathers them straight into
  // the staging block.  Empty when the batch was prepared by one thread: then the pools above hold everything.
  std::vector<Prepared> frags;
  size_t n_rows() const { size_t n = rows.size(); for (const Prepared& f : frags) n += f.rows.size(); return n; }
  size_t n_visits() const { size_t n = visits.size(); for (const Prepared& f : frags) n += f.visits.size(); return n; }
  size_t n_f64() const { size_t n = f64pool.size(); for (const Prepared& f : frags) n += f.f64pool.size(); return n; }
  size_t n_chars() const { size_t n = chars.size(); for (const Prepared& f : frags) n += f.chars.size(); return n; }
};

// The tables of a batch are tens to hundreds of megabytes of std::vector storage that every batch would otherwise take fresh from the
// kernel (page faults: a quarter to a half of the preparation's CPU time).  recycle_prepared hands the storage of a finished batch —
// the object's own vectors and those of its fragments — to a pool (bounded: HIPSTR_PREP_POOL_MIB, default 2048); prepare_batch draws
// its fragments from it and adopt_recycled lets a fresh top-level object start with pooled capacity.  `p` is left empty.
void recycle_prepared(Prepared& p);
void adopt_recycled(Prepared& p);

// Returns 0 on success; otherwise fills err.
// seed_in: optional [n_reads] seeds chosen by the caller (HapAligner::process_read's seed_base argument, HapAligner.h:83);
// HIPSTR_SEED_AUTO entries are computed with calc_seed_base.
int prepare_batch(const hipstr_batch_t* b, Prepared& out, std::string& err, int64_t ws_budget_doubles = (int64_t)3 << 30, const int32_t* seed_in = NULL);

// Returns 0 if prepare_batch would accept the batch; otherwise its error message in err.
int check_batch(const hipstr_batch_t* b, std::string& err);
// The offset tables of a batch are consistent with each other (counts in range, offsets non-negative and never decreasing, CIGAR runs positive):
// called by every entry point that takes a caller's batch, before anything indexes with the tables.
int validate_tables(const hipstr_batch_t* b, std::string& err);
void prep_profile_print();
// one locus of it; the cursor into opt_off is advanced.  seeds_out (optional, indexed by the batch's read index): the seed bases the check computed anyway
// (HIPSTR_SEED_AUTO for reads that are not realigned), so that prepare_batch does not walk the CIGARs a second time
// Estimate of the device work of locus l (first block option opt0 in opt_off), in units of one north-star (read, allele) pair: what the
// shards of a region list are balanced by (sum of P A L H) — reads x realigned alleles x (mean read length / 150) x
// [flank sweeps ~ flank bases / 60 x 1.4  +  STR block ~ 1 + 2.4 interruptions of the repeat per allele] (the measured ratios of a pass:
// profiles/r04_ns_kernel_stats.txt, profiles/r04_bench_ns_inherit*.json).  Cheap: one scan of the locus' STR options.
double locus_cost(const hipstr_batch_t* b, int l, int opt0);
// dims_out (optional): [0] = the longest realigned read, [1] = the longest STR allele of the locus — what the per-read STR kernels' LDS is
// sized by, batch-wide (layout.h hs_str_kernel_lds_bytes): whoever puts loci into one batch keeps the combined figure within HS_LDS_LIMIT
int check_locus(const hipstr_batch_t* b, int locus, int* opt_cursor_io, std::string& err, int32_t* seeds_out = NULL, int* dims_out = NULL);

// HapAligner::calc_seed_base (HapAligner.cpp:238-318).  Returns -2 on the inputs the reference dies on.
int calc_seed_base(const hipstr_batch_t* b, int locus, int read);

// Option index per block of allele k in Haplotype::next() order (Haplotype.cpp:123-196).
void allele_options(const int32_t nopts[3], int k, int32_t opts[3]);

// Rows of the leading and trailing flank of one allele in one orientation under the allele's OWN homopolymer context
// (what a non-reusing alignment such as trace_optimal_aln computes); side_seqs = the three block sequences in side order.
void fresh_flank_rows(const std::string side_seqs[3], std::vector<hs_row_t>& lead, std::vector<hs_row_t>& trail);
// Appends one hs_stropt_t (+ visiting lists, f64 constants, block bytes) for a block sequence given in side orientation.
void append_stropt(const std::string& blk, int period, const double* stutter, Prepared& out);        // (constants and table written by the host: hs_stropt_t::gen = 0)

// One {A, G, Bnd} entry of the tabulated closed form of a simple visiting list (diagnostics / tests).
void debug_simple_table(int lim, int U0, int tail, double ent[3]);

// StutterModel::log_stutter_pmf (stutter_model.cpp:29-53) from the six constructor parameters.
double log_stutter_pmf(const double* sp, int period, int sample_bps, int read_bps);

}  // namespace hipstr
