// layout.h — HBM layout of a prepared (device-resident) batch, shared by the host
// preparation code (prep.cpp) and the HIP kernels (hmm_kernels.hip).
//
// Everything the kernels touch is a flat pool + small POD tables of offsets; nothing
// is pointer-chased.  Terminology follows the reference: a locus has candidate
// haplotypes ("alleles") built from [left flank | STR block | right flank]; each
// pooled read is split at its seed base into a LEFT problem (read prefix vs the
// forward haplotype) and a RIGHT problem (reversed read suffix vs the reversed
// haplotype) — HapAligner.cpp:606-628.  "side" 0 = left/forward, 1 = right/reversed.
#pragma once
#include <stdint.h>
#include <stddef.h>

#define HS_PW_SLOTS      10        // 64-bit descriptor slots per piecewise-simple visiting list (prep.cpp emit_stropt)
#define HS_SHAPE_PIECEWISE (-2)
#define HS_SHAPE_PWK     (-3)      // piecewise with three to HS_PWK_MAX breaks (round 4): closed form from HS_PWK_SLOTS slots per list, behind the ten-slot descriptors
#define HS_PWK_MAX       6         // breaks per list the K-level closed form takes (three interruptions of the repeat)
#define HS_PWK_SLOTS     24        // 64-bit slots per list: 0 nseg | terminal ni   1 first plain ni | one past the last   2+3s run ni | run U   3+3s ln(run U)   4+3s break ni | ca + 256 cb   (s = 0..HS_PWK_MAX)
#define HS_TAB_CAP       48        // closed-form table entries per STR option the STR kernel keeps in LDS (hs_stropt_t::tab_*)
#ifndef HS_GRP_COLS
#define HS_GRP_COLS      256       // lanes (= read columns) of one hs_str_group_kernel workgroup; prep.cpp packs reads of a locus side up to this many columns
#endif
#define HS_GRP_MAX_BLOCK (4*HS_GRP_COLS)   // longest STR block hs_str_group_kernel / _pw / _rp take: a workgroup fetches an allele's block four bases per lane (hs_str_group_kernel_p never fetches it: any length); longer ones go to the per-read kernels (prep.cpp, round 6: tools/fuzz_align.py "big" found alleles of 1026+ bp mis-scored)
#define HS_GRP_MAXP      6         // hs_str_group_kernel_p is instantiated for the periods 1..HS_GRP_MAXP
#define HS_NART          13        // artifact sizes -6p..+6p (RepeatStutterInfo.h:10-11)
#define HS_MAXREP        6
#define HS_MAX_COLS      16        // max read columns per lane in the systolic sweep of the traceback fill (trace.hip): sides of up to 1024 bases, like the forward pass
                                   // (1-6 columns per lane: static LDS; 8 / 12 / 16 for the rare longer sides: dynamic LDS up to 146 KiB)
#define HS_MAX_SIDE_LEN  256       // columns of a read side the grouped STR kernels and hs_nd_kernel hold (longer sides: hs_str_kernel, a workgroup per read)
#define HS_MAX_SIDE_FWD  1024      // longest read side the forward pass takes (HapAligner sizes its matrices by the read, HapAligner.cpp:593-602)
#define HS_MAX_STR_BP    2047      // longest STR allele (11-bit block length fields)
#define HS_IMPOSSIBLE    (-1000000000.0)   // HapAligner.cpp:20
#define HS_REDO          (1.0e300)         // mark in the MR workspace: "this chunk of columns is left to hs_str_kernel_generic"
#define HS_ND_TOTAL 192      // sum over the six deletion sizes of min(|D|, n) <= 21 p <= 189
#define HS_WAVE_LDS (HS_ND_TOTAL + 24 + 2*HS_TAB_CAP)   // doubles of per-wavefront LDS of hs_str_kernel: nd | cstl | tab
#define HS_LDS_LIMIT (160*1024)
// LDS bytes of one hs_str_kernel workgroup (both sides of a read) when the longest read has lds_len bases and the longest STR allele
// max_B: the host refuses a locus whose reads and alleles would not fit (check_locus), before it shares a batch with others.
static inline size_t hs_str_kernel_lds_bytes(int lds_len, int max_B){
  const size_t Lc = ((size_t)lds_len + 3) & ~(size_t)1;
  const size_t ilog_len = ((size_t)max_B + 9) & ~(size_t)1, blk_len = ((size_t)max_B + 19) & ~(size_t)15;
  return Lc*16 + Lc*8*2 + Lc*8*6 + ilog_len*8 + 2*HS_WAVE_LDS*8 + 2*blk_len + ((Lc + 15) & ~(size_t)15);
}

// One haplotype row of a flank block, as the flank sweeps consume it.
//   bits  0..7   haplotype base (raw char, compared with read chars for equality)
//   bits  8..11  min(15, homopolymer length) -> index into LOG_MATCH_TO_* (HapAligner.cpp:119-120)
//   bits 12..23  compact row index u (see lastcol layout below)
//   bit  31      valid
typedef uint32_t hs_row_t;
#define HS_ROW_VALID 0x80000000u

// A contiguous run of rows in the row pool: the rows of one flank block option under
// one homopolymer context.  Row 0 of a lead rowset is matrix row 0 (HapAligner.cpp:36-42);
// row 0 of a trail rowset is the "stutter block must be followed by a match" row
// (HapAligner.cpp:130-139).
struct hs_rowset_t { int32_t off, len; };

// Visiting list entry for StutterAlignerClass's artifact-position loops
// (StutterAlignerClass.cpp:74-96 and :123-142): which block offsets the loop visits is
// a function of the block sequence only, so the host enumerates them once per STR
// option and the device replays the list uniformly across lanes.
//   meta bits  0..15  ni = -i, the (non-positive) loop variable negated
//        bits 16..31  U  = upstream match run length at this position (0 = none)
//        bits 32..39  ca = block char whose emission is SUBTRACTED when U == 0
//        bits 40..47  cb = block char whose emission is ADDED when U == 0
//        bit  48      plain: push the running value unchanged (insertion branch `-i+p >= B`)
//   logU = int_log(U) (mathops.cpp:13-21), valid when U > 0
struct hs_visit_t { uint64_t meta; double logU; };

// Per (STR option, side): block sequence in side orientation + everything derived from it.
struct hs_stropt_t {
  int32_t seq_off;           // into char pool
  int32_t B;                 // block length
  int32_t nd;                // num_deletions_ (StutterAlignerClass.h:64-69)
  int32_t period;
  int32_t f64_off;           // into f64 pool: pmf[13] | -int_log(B+1) | -int_log(B+D+1) for D=-p..-6p | [7 x HS_PW_SLOTS descriptor slots, only if some list's shape is HS_SHAPE_PIECEWISE or the option is of kind 3] | [7 x HS_PWK_SLOTS, only if some list's shape is HS_SHAPE_PWK]
  int32_t ins_off, ins_len;  // visiting list shared by all insertion sizes
  int32_t del_off[HS_MAXREP], del_len[HS_MAXREP];
  // Shape of each visiting list (index 0..5: deletion lists, 6: insertion list).  Periodic blocks give "simple" lists —
  // at most one run-skip entry at offset 0 (covering offsets 0..U0-1) followed only by plain entries at consecutive
  // offsets — whose log-sum-exp has a closed form in (lp0, bound); everything else replays the list.
  //   shape = -1: generic list;  shape = U0 >= 0: simple list, U0 = 0 means "no skip entry, plain entries from offset 0";
  //   shape = HS_SHAPE_PIECEWISE: one or two interruptions, closed form from the ten descriptor slots behind the constants
  //   shape = HS_SHAPE_PWK: three to HS_PWK_MAX breaks, closed form from the K-level slots (hs_str_group_kernel_rp); every other consumer
  //                         treats it like -1 and replays the list, which is kept
  int32_t shape[HS_MAXREP + 1];
  // Tabulated closed form of the simple lists (prep.cpp simple_table): for a simple list the log-sum-exp over the artifact
  // positions is lp0 + A + G with A, G functions of the lane's bound only — as long as the float conversions inside
  // fast_log_sum_exp round the way they do for an exact lp0, which holds whenever |lp0| < Bnd (else the kernel evaluates
  // the closed form the long way).  Entry e of list k sits at f64 pool [tab_off + 3*(tab_base[k] + e)] = {A, G, Bnd}, the smallest
  // Bnd of the table (what the kernel actually compares with) at [tab_off + 3*tab_len];
  // e = [bound > 0] + max(bound - U0, 0).  tab_len = 0: no table (a list of the option is not simple, or too many entries).
  int32_t tab_off, tab_len;
  int32_t tab_base[HS_MAXREP + 1];
  // ins_probs_ (StutterAlignerClass.cpp:40-51) cycles through the block's last `period` bases; where the block's right end is
  // periodic, its first nd_eq repeat units add exactly the terms of del_probs_ / match_probs_, in the same order: the STR kernels
  // take those sums from the tables they already hold and only loop over the remaining (6 - nd_eq) units.  nd_eq <= nd.
  int32_t nd_eq;
  // Base codes ((char >> 1) & 3: A, C, T, G -> 0, 1, 2, 3) of the block's last 16 bases, two bits each, the last base in bits 0-1: a
  // tabulated block is periodic, so base t from its right end is code (t mod period) of this word (hs_str_group_kernel_p).
  int32_t tail_codes;
  // Which forward kernel evaluates the option.  1: every list simple and tabulated (tab_len > 0): the grouped kernels with the table;
  // 2: every list simple (tabulated: tab_len entries) or piecewise simple (descriptor slots), block of A/C/G/T: hs_str_group_kernel_pw;
  // 3: like 2 with at least one list that is neither (replayed entry by entry, still in the grouped layout): hs_str_group_kernel_rp;
  // 0: none of these (blocks with other characters than A/C/G/T, tables too large: hs_str_kernel_generic).  A kind-2/3 option's table holds
  // the entries of its simple lists only.
  int32_t kind;
  // Round 4: the constants and the closed-form table of an option whose whole block repeats with the period (nearly all) are not built
  // on the host any more.  gen = 1: f64_off / tab_off count from the start of the GENERATED region of the f64 pool (hs_dev_t::f64_gen_base),
  // pmf_off = the locus' 13 stutter-pmf values in hs_dev_t::pmf13; hs_expand_stropts_kernel writes the 20 constants and the table there,
  // makes the two offsets pool-wide and clears the flag before any other kernel runs.  gen = 0: as before (host-written, pool-wide).
  int32_t gen;
  int32_t pmf_off;
};

// What the host says about one record of grp_recs[] (below): hs_expand_recs_kernel assembles the 256-byte record from it, the option
// record and the option's constants (host-written or generated).  flags = lead slot (10 bits) | bit 29 / bit 30 of the processing order.
struct hs_recdesc_t { int32_t stropt; int32_t flags; int32_t re_ord; int32_t nd_row; };

struct hs_allele_t {
  int32_t lead_rows[2];      // rowset id per side
  int32_t trail_rows[2];
  int32_t str_opt[2];        // hs_stropt_t index per side
  int32_t n_flank;           // F0 + F2: number of non-STR haplotype bases (compute_aln_logprob's num_seeds)
  int32_t realign;           // realign_to_haplotype flag
  int32_t lead_slot[2];      // which of the locus' distinct leading-flank rowsets (per side) this allele uses
  int32_t re_ord;            // ordinal among the realigned alleles of the locus (workspace row)
  int32_t pad;
};

struct hs_locus_t {
  int64_t out_off;           // offset of this locus' [P x A] block in aln_probs
  int32_t hap_begin, n_alleles;
  int32_t read_begin, n_reads;
  int32_t n_re;              // realigned alleles
  int32_t lt_stride;         // max n_flank over realigned alleles: stride of the trailing last-column workspace
  int32_t lead_flank[2];     // max leading-flank length per side (lead workspace record = n_side + lead_flank + 1 doubles)
  int32_t n_lead[2];         // distinct leading-flank rowsets per side
  int32_t tg_begin[2];       // trailing-flank allele groups of this locus per side: range in tgroups[]
  int32_t tg_count[2];
  int32_t order_off[2];      // STR-kernel processing order of the realigned alleles per side: range [order_off, +n_re) in str_order[]
  int32_t n_tab[2];          // the first n_tab positions of that order are the alleles with a tabulated closed form (hs_stropt_t::tab_len > 0)
  int32_t n_short[2];        // hs_str_group_kernel_p takes the positions [n_short, n_tab) of that order, hs_str_group_kernel the first n_short:
                             // all n_tab of them where the period is above HS_GRP_MAXP, none otherwise
  int32_t rec_off[2];        // first record (of n_tab, in the side's order) in grp_recs[], in records
  int32_t ndrow_off[2];      // read-end deletion sums of the side's alleles in [n_short, n_tab) (hs_nd_kernel): first row descriptor in nd_rows[] ...
  int32_t n_ndrows[2];       // ... and their number; a read side's block in the ws_nd workspace is n_ndrows x 6 period doubles
  int32_t n_pw[2];           // positions [n_tab, n_pw) of the order: alleles whose lists are simple or piecewise simple (hs_stropt_t::kind 2),
                             // hs_str_group_kernel_pw's; the rest, [n_pw, n_re), is hs_str_kernel_generic's
  int32_t period;            // the locus' STR period (the same for all of its alleles)
  int32_t fused;             // always 0 (round 5's fused trailing-flank item was measured slower and removed in round 6; the field keeps the record's layout)
                             // (hs_trail_kernel_coop, side == 2; prep.cpp trail_fusable): the two sides' allele groups are the same lists,
                             // every allele of a group has the same flank configuration; hs_combine_kernel leaves this locus' reads alone
  int32_t n_rp[2];           // positions [n_pw, n_rp) of the order: alleles with a list that has no closed form (three and more interruptions;
                             // hs_stropt_t::kind 3): hs_str_group_kernel_rp replays it inside the grouped layout; [n_rp, n_re) is hs_str_kernel_generic's
};

// One row of read-end deletion sums (hs_nd_kernel): the start values of StutterAlignerClass.cpp:117-120 for the columns within six repeat
// units of the read end against a block remainder of `len` bases (block length minus deletion size) whose bases, from the right end, are
// the tail codes repeated.  A family of alleles whose blocks grow by one repeat unit shares rows: size q of the family's allele k is row
// (k - q + 5) of the family, which is how the sums are inherited from allele to allele (a row is computed once).  len < 0: unused row.
struct hs_ndrow_t { int32_t len; int32_t tail_codes; };

// hs_str_group_kernel_p reads everything an allele needs that is the same for all lanes — block length, flags, table indices, the 20
// constants — with scalar loads from ONE record per (locus, side, position in the side's order), HS_GRP_REC_DWORDS dwords:
//   [0] lead slot (10 bits) | tab_len << 10 (8 bits) | bit 29: block = the previous position's plus one repeat unit | bit 30: ... ends with it
//   [1] re_ord   [2] tail_codes (12 bits: six bases) | B << 12   [3] tab_off (f64 pool)   [4] row of the allele's size-0 read-end sums (size q: [4] - q)
//   [8..14] per visiting list k (0..5 deletion sizes, 6 insertions): shape U0 | tab_base << 16        [5..7, 15] unused
//   [16..55] 20 doubles: pmf[13] | prior_ins | prior_del[6]    [56..57] the table's smallest Bnd    [58..63] unused
#define HS_GRP_REC_DWORDS 64

struct hs_read_t {
  int32_t base_off;          // into bases/quals pools
  int32_t len;
  int32_t seed;              // calc_seed_base result (-1 = none)
  int32_t locus;
};

// Workspace offsets (in doubles) of one active read inside the current chunk's workspaces.
//   mr   : [n_re][len-1]   M of the STR block's last row, left side columns then right side columns
//   lt   : [n_re][lt_stride] last read column of the trailing-flank rows: left side (F2 rows) then right side (F0 rows)
//   lead : per side [n_lead][n_side + lead_flank + 1]: rowP (M of the row before the STR block) | last column of the
//          leading-flank rows | side_prob
//   col  : [len-1][3] per read column, left side then right side: log P(correct), log P(error), base (as a double)
struct hs_ws_t { int64_t mr, lt, lead[2], col, nd[2]; };       // nd: per side [n_ndrows][6 period] read-end deletion sums (hs_nd_kernel -> hs_str_group_kernel_p)

// Alleles of one locus and side that share a trailing-flank rowset (identical rows incl. homopolymer context):
// the trailing-flank kernel runs them as the 64 lanes of one wavefront.
struct hs_tgroup_t { int32_t rowset; int32_t member_off; int32_t n_members; int32_t pad; };

// Work items of the phase kernels (sorted by columns-per-lane class where the kernel is templated on it).
struct hs_item_t { int32_t active; int32_t side; int32_t rowset; int32_t slot; };

// Kernel argument block (all device pointers).  HS_P(T) is `T*`; the kernel translation unit defines it as a pointer into the global
// address space before including this header, so that loads and stores through these fields are global_* instead of flat_*
// instructions (a flat access also counts against the LDS wait counter).
#ifndef HS_P
#define HS_P(T) T*
#endif
struct hs_dev_t {
  HS_P(const hs_locus_t) loci;
  HS_P(const hs_allele_t) alleles;
  HS_P(const hs_stropt_t) stropts;
  HS_P(const hs_rowset_t) rowsets;
  HS_P(const hs_row_t) rows;
  HS_P(const hs_visit_t) visits;
  HS_P(const double) f64pool;
  HS_P(const char) chars;      // STR block sequences
  HS_P(const hs_read_t) reads;
  HS_P(const char) bases;
  HS_P(const char) quals;
  HS_P(const int32_t) active;     // read indices that need alignment (realign && seed >= 0)
  HS_P(const hs_ws_t) ws;         // [n_active] workspace offsets
  HS_P(const hs_item_t) items;      // lead items and trail items, grouped (see api.hip)
  HS_P(const hs_tgroup_t) tgroups;
  HS_P(const int32_t) tmembers;   // allele indices (within the locus) of the trail groups
  HS_P(const int32_t) tpack;      // active-read indices of the reads packed into one trail item
  HS_P(const int32_t) grp_recs;   // hs_str_group_kernel_p: HS_GRP_REC_DWORDS dwords per tabulated position of a locus side's order (hs_locus_t::rec_off);
                                  // device only: assembled by hs_expand_recs_kernel from rec_descs[]
  HS_P(const hs_recdesc_t) rec_descs;
  HS_P(const double) pmf13;       // per locus: log_stutter_pmf of the 13 artifact sizes (host libm), what the generated constants start from
  HS_P(const hs_ndrow_t) nd_rows; // row descriptors of the read-end deletion sums, per locus side (hs_locus_t::ndrow_off)
  HS_P(const int32_t) str_order;  // allele index (within the locus) per processing position; bit 30 set = this allele's STR block,
                                 // in side orientation, ends with the previous position's block (its tables are continued); bit 29
                                 // set = ... and is that block plus one repeat unit, periodic, with all six deletion sizes
  HS_P(double) ws_col;
  HS_P(double) ws_band;    // per persistent wavefront: 2 x [band_cols][64 lanes][2] band-boundary rows (M, D)
  HS_P(double) ws_lts;     // unused (NULL): scratch of the removed fused trailing-flank item; kept for the argument block's layout
  HS_P(double) ws_mr;
  HS_P(double) ws_lt;
  HS_P(double) ws_lead;
  HS_P(double) ws_nd;
  HS_P(double) aln_probs;
  HS_P(int32_t) redo;       // [n_active] set by hs_str_kernel when it left HS_REDO marks for a read; cleared before every pass
  // constant tables
  HS_P(const double) int_log;    // [10000]
  HS_P(const double) qual_correct; // [256] indexed by raw quality char (clamps applied)
  HS_P(const double) qual_error;   // [256]
  HS_P(const double) m2m;        // [16] LOG_MATCH_TO_MATCH
  HS_P(const double) m2i;        // [16] LOG_MATCH_TO_INS (== LOG_MATCH_TO_DEL, AlignmentModel.cpp:27-28)
  double             log_thresh;   // LOG_THRESH = log(0.001), mathops.h:36 (host libm bits)
  double             log_half;     // LOG_ONE_HALF, mathops.cpp:9
  int32_t            n_active;
  int32_t            allele_chunk;   // alleles per workgroup
  int32_t            grp_nd_cap;     // hs_str_group_kernel: doubles of the read-end deletion table = max over the groups of reads x 36 period
  int32_t            lds_len;        // max read length in the batch (LDS carve of the STR kernel)
  int32_t            band_cols;      // max columns of one read side (rows of a band-boundary buffer)
  int32_t            lts_rows;       // rows of a workgroup's ws_lts block: the batch's largest lt_stride
  int32_t            max_B;          // longest STR allele of the batch (LDS carve of the STR kernel)
  int32_t            debug_redo;     // tests (HIPSTR_DEBUG_REDO=k): hs_str_kernel leaves every k-th chunk to the re-do path
  int32_t            n_stropts, n_recs;
  int64_t            f64_gen_base;   // doubles: the f64 pool is [host-written part | generated part]
};
