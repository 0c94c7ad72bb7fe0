/*
 * hipstr_oracle.c — TEST INFRASTRUCTURE ONLY.  CPU restatement (plain C11,
 * double precision, single thread) of HipSTR v0.7's read-to-haplotype HMM
 * (forward score and Viterbi traceback), diplotype posteriors and genotype calls,
 * de novo stutter EM and Needleman-Wunsch.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this file's shared object; the
 * product library never does.
 *
 * PARITY STATUS: pinned.  tests/test_oracle_vs_ref.py and the *_oracle.py tests of
 * the later stages (traceback, genotype calls, EM, Needleman-Wunsch, haplotype
 * alignment strings) compare every entry point with the compiled reference (oracle/_ref/libhipstr_ref.so, built by
 * oracle/Makefile from the sources under /root/reference) on seeded random loci
 * and on the two known-answer vectors of SURVEY.md §8(c); the committed fixtures
 * under tests/golden/ were produced by that reference build
 * (tests/golden/make_golden.py).  Agreement is bit-exact (max |diff| == 0).
 *
 * Each function cites the reference code (HipSTR v0.7 paths) whose arithmetic,
 * operation order and tie rules it restates.  The restatement works on the flat
 * hipstr_batch_t, keeps the reference's evaluation order for every double
 * operation (so results are bit-identical, which matters because the float
 * log-sum-exp approximations are discontinuous in their inputs), and simulates
 * the reference's allele loop literally — including its reuse of leading-flank
 * rows computed under an earlier allele's homopolymer context
 * (HapAligner.cpp:54-60, 612-634).
 */
#include <float.h>
#include <math.h>
#include "../hipstr_amd/csrc/cr_math.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/hipstr_hmm.h"

#define IMPOSSIBLE     (-1000000000.0)   /* HapAligner.cpp:20 */
#define LARGE_NEGATIVE (-10e6)           /* RepeatStutterInfo.h:12 */
#define MIN_SEED_DIST  5                 /* HapAligner.cpp:17 */
#define INT_LOG_N      10000             /* mathops.cpp:13 */

/* ------------------------------------------------------------------ tables */
static double g_int_log[INT_LOG_N];                 /* mathops.cpp:13-21 */
static double g_m2m[16], g_m2i[16], g_m2d[16];      /* AlignmentModel.cpp:20-32 */
static double g_q_correct[256], g_q_error[256];     /* base_quality.h:29-38 */
static double g_log_thresh, g_log_half;             /* mathops.h:36, mathops.cpp:9 */
static const double I2I = -1.0, I2M = -0.4586751453870818910216436;   /* AlignmentModel.h:7-10 */
static const double D2D = -1.0, D2M = -0.4586751453870818910216436;
static int g_ready = 0;

static void oracle_init(void){
  if (g_ready) return;
  g_int_log[0] = -1000;
  for (int i = 1; i < INT_LOG_N; i++) g_int_log[i] = log((double)i);
  static const double dindel[10] = {2.9e-5, 2.9e-5, 2.9e-5, 2.9e-5, 4.3e-5, 1.1e-4, 2.4e-4, 5.7e-4, 1.0e-3, 1.4e-3};
  g_m2m[0] = g_m2i[0] = g_m2d[0] = 0;
  for (unsigned int i = 1; i <= 15; i++){
    g_m2i[i] = (i <= 10 ? log(dindel[i-1]) : log(dindel[9]+(4.3e-4)*(i-10)));
    g_m2d[i] = g_m2i[i];
    g_m2m[i] = log(1.0 - exp(g_m2i[i]) - exp(g_m2d[i]));
  }
  g_q_correct[0] = -100000;
  g_q_error[0]   = -log(3);
  for (int i = 1; i <= 'J'-'!'; i++){
    g_q_correct[i] = log(1.0 - pow(10.0, i/(-10.0)));
    g_q_error[i]   = log(pow(10.0, i/(-10.0))/3.0);
  }
  g_log_thresh = log(0.001);
  g_log_half   = log(0.5);
  g_ready = 1;
}

static int qual_index(char q){            /* base_quality.h:44-75 clamps */
  if (q < '!') return 0;
  if (q > 'J') return 'J'-'!';
  return q - '!';
}

/* ------------------------------------------- float approximations (A.5) */
static float bits_to_float(uint32_t u){ float f; memcpy(&f, &u, 4); return f; }
static uint32_t float_to_bits(float f){ uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp / log of the posterior, genotype-call and EM stages: the host libm's (what the reference calls; the default), or — oracle_set_cr_math(1) —
 * the correctly rounded ones of hipstr_amd/csrc/cr_math.h, the functions the device kernels evaluate.  With the switch on this file is an
 * operation-for-operation CPU restatement of the device path, and the two settings agree wherever the host libm is itself correctly
 * rounded (tests/test_cr_math.py: all but ~8 in 10^4 exp and ~1 in 10^5 log arguments with glibc).  The model tables of oracle_init are
 * the host libm's in either setting (the library builds them on the host as well). */
static int g_cr_math = 0;
void oracle_set_cr_math(int on){ g_cr_math = on; }
#define X_EXP(x) (g_cr_math ? cr_exp(x) : exp(x))
#define X_LOG(x) (g_cr_math ? cr_log(x) : log(x))

static float o_fasterexp(float p){                   /* fastonebigheader.h:206-218 */
  float y = 1.442695040f * p;
  float clipp = (y < -126) ? -126.0f : y;
  return bits_to_float((uint32_t)((float)(1 << 23) * (clipp + 126.94269504f)));
}
static float o_fasterlog(float x){                   /* fastonebigheader.h:348-358 */
  float y = (float)float_to_bits(x);
  y *= 8.2629582881927490e-8f;
  return y - 87.989971088f;
}
static float o_fastexp(float p){                     /* fastonebigheader.h:188-204 */
  float q = 1.442695040f * p;
  float offset = (q < 0) ? 1.0f : 0.0f;
  float clipp = (q < -126) ? -126.0f : q;
  int w = (int)clipp;
  float z = clipp - w + offset;
  return bits_to_float((uint32_t)((float)(1 << 23) * (clipp + 121.2740575f + 27.7280233f / (4.84252568f - z) - 1.49012907f * z)));
}
static float o_fastlog(float x){                     /* fastonebigheader.h:320-338 */
  uint32_t vi = float_to_bits(x);
  float mx = bits_to_float((vi & 0x007FFFFF) | 0x3f000000);
  float y = (float)vi;
  y *= 1.1920928955078125e-7f;
  float l2 = y - 124.22551499f - 1.498030302f * mx - 1.72587999f / (0.3520887068f + mx);
  return 0.69314718f * l2;
}

double oracle_fast_lse_vec(const double* v, int n){  /* mathops.cpp:97-106 */
  oracle_init();
  double m = v[0];
  for (int i = 1; i < n; i++) if (v[i] > m) m = v[i];
  double total = 0;
  for (int i = 0; i < n; i++){
    double diff = v[i] - m;
    if (diff > g_log_thresh) total += o_fasterexp((float)diff);
  }
  return m + o_fasterlog((float)total);
}

double oracle_fast_lse2(double a, double b){         /* mathops.cpp:86-95 */
  oracle_init();
  if (a > b){
    double diff = b - a;
    return diff < g_log_thresh ? a : a + o_fastlog(1 + o_fastexp((float)diff));
  } else {
    double diff = a - b;
    return diff < g_log_thresh ? b : b + o_fastlog(1 + o_fastexp((float)diff));
  }
}

double oracle_log_sum_exp(const double* v, int n){   /* mathops.cpp:44-50 (exact) */
  double m = v[0];
  for (int i = 1; i < n; i++) if (v[i] > m) m = v[i];
  double total = 0.0;
  for (int i = 0; i < n; i++) total += X_EXP(v[i] - m);
  return m + X_LOG(total);
}

double oracle_int_log(int v){ oracle_init(); return g_int_log[v]; }
double oracle_transition(int which, int h){
  oracle_init();
  switch (which){ case 0: return g_m2m[h]; case 1: return g_m2i[h]; case 2: return g_m2d[h];
                  case 3: return I2I; case 4: return I2M; case 5: return D2D; default: return D2M; }
}
double oracle_base_quality(int q, int correct){ oracle_init(); return correct ? g_q_correct[qual_index((char)q)] : g_q_error[qual_index((char)q)]; }

/* stutter_model.cpp:29-53 + ctor logs (stutter_model.h:31-60) */
double oracle_stutter_pmf(const double* sp, int period, int sample_bps, int read_bps){
  double in_step = X_LOG(1-sp[0]), in_nostep = X_LOG(sp[0]), in_up = X_LOG(sp[1]), in_down = X_LOG(sp[2]);
  double out_step = X_LOG(1-sp[3]), out_nostep = X_LOG(sp[3]), out_up = X_LOG(sp[4]), out_down = X_LOG(sp[5]);
  double log_equal = X_LOG(1-sp[1]-sp[2]-sp[4]-sp[5]);
  int bp_diff = read_bps - sample_bps;
  if (bp_diff % period != 0){
    int eff = bp_diff - (bp_diff/period);
    if (eff < 0) return out_down + out_nostep + out_step*(-eff-1);
    return out_up + out_nostep + out_step*(eff-1);
  }
  int rep = bp_diff/period;
  if (rep == 0) return log_equal;
  if (rep < 0) return in_down + in_nostep + in_step*(-rep-1);
  return in_up + in_nostep + in_step*(rep-1);
}

/* ------------------------------------------------- haplotype side model */
typedef struct {
  int nopts;
  char** seq;     /* [nopts] in this side's orientation */
  int*   len;
  int**  llen;    /* HapBlock.cpp:7-30 homopolymer run tables */
  int**  rlen;
} OBlock;

typedef struct {
  OBlock blk[3];          /* in side order (rev: blocks reversed, sequences reversed) */
  int    str_blk;         /* index of the STR block (always 1) */
  int    counts[3];       /* current option per block */
  int    max_size;
  /* per STR option: upstream-match tables U[d-1][x] for shifts (d*period), StutterAlignerClass.h:35-42,70-74 */
  int*** ups;             /* [nopts][ntab][len] */
  int*   ntab;            /* tables per option */
  int*   ndel;            /* num_deletions_ per option (StutterAlignerClass.h:64-69) */
} OSide;

static void run_tables(const char* s, int n, int** ll, int** rl){
  if (n == 0){ *ll = NULL; *rl = NULL; return; }
  int* l = malloc(sizeof(int)*n); int* r = malloc(sizeof(int)*n);
  l[0] = 0; int count = 0;
  for (int j = 1; j < n; j++){ count = (s[j-1] == s[j] ? count+1 : 0); l[j] = count; }
  r[n-1] = 0;
  for (int j = n-2; j >= 0; j--){ count = (s[j+1] == s[j] ? count+1 : 0); r[j] = count; }
  *ll = l; *rl = r;
}

static int* upstream_matches(const char* s, int n, int shift){   /* StutterAlignerClass.h:35-42 */
  int* m = malloc(sizeof(int)*(n > 0 ? n : 1));
  for (int i = 0; i < (shift < n ? shift : n); i++) m[i] = 0;
  for (int i = shift; i < n; i++) m[i] = (s[i-shift] != s[i] ? 0 : 1 + m[i-1]);
  return m;
}

static void side_build(OSide* sd, const hipstr_batch_t* b, int l, int opt_base, int reversed){
  int period = b->period[l];
  int cursor = opt_base;
  const char* src[3][1024]; int slen[3][1024]; int nopt[3];
  for (int k = 0; k < 3; k++){
    nopt[k] = b->blk_nopts[3*l+k];
    for (int o = 0; o < nopt[k]; o++, cursor++){ src[k][o] = b->seq + b->opt_off[cursor]; slen[k][o] = b->opt_off[cursor+1]-b->opt_off[cursor]; }
  }
  sd->max_size = 0;
  for (int k = 0; k < 3; k++){
    int from = reversed ? 2-k : k;
    OBlock* ob = &sd->blk[k];
    ob->nopts = nopt[from];
    ob->seq = malloc(sizeof(char*)*ob->nopts); ob->len = malloc(sizeof(int)*ob->nopts);
    ob->llen = malloc(sizeof(int*)*ob->nopts); ob->rlen = malloc(sizeof(int*)*ob->nopts);
    int mx = 0;
    for (int o = 0; o < ob->nopts; o++){
      int n = slen[from][o];
      ob->len[o] = n; ob->seq[o] = malloc(n+1);
      for (int i = 0; i < n; i++) ob->seq[o][i] = reversed ? src[from][o][n-1-i] : src[from][o][i];
      ob->seq[o][n] = 0;
      run_tables(ob->seq[o], n, &ob->llen[o], &ob->rlen[o]);
      if (n > mx) mx = n;
    }
    sd->max_size += mx;
  }
  sd->str_blk = 1;
  OBlock* sb = &sd->blk[1];
  sd->ups = malloc(sizeof(int**)*sb->nopts); sd->ntab = malloc(sizeof(int)*sb->nopts); sd->ndel = malloc(sizeof(int)*sb->nopts);
  for (int o = 0; o < sb->nopts; o++){
    int B = sb->len[o];
    int nd = HIPSTR_MAX_STUTTER_REPS;               /* -(max_deletion/period) */
    while (nd*period > B) nd--;
    sd->ndel[o] = nd;
    int nt = nd > 0 ? nd : 1;                       /* "required for insertion calculations" */
    sd->ntab[o] = nt;
    sd->ups[o] = malloc(sizeof(int*)*nt);
    for (int t = 0; t < nt; t++) sd->ups[o][t] = upstream_matches(sb->seq[o], B, (t+1)*period);
  }
}

static void side_free(OSide* sd){
  for (int k = 0; k < 3; k++){
    OBlock* ob = &sd->blk[k];
    for (int o = 0; o < ob->nopts; o++){ free(ob->seq[o]); free(ob->llen[o]); free(ob->rlen[o]); }
    free(ob->seq); free(ob->len); free(ob->llen); free(ob->rlen);
  }
  OBlock* sb = &sd->blk[1];
  for (int o = 0; o < sb->nopts; o++){ for (int t = 0; t < sd->ntab[o]; t++) free(sd->ups[o][t]); free(sd->ups[o]); }
  free(sd->ups); free(sd->ntab); free(sd->ndel);
}

/* Haplotype.cpp:239-287, including the fact that the cross-block extension never looks past one neighbour */
static int nb_left(const OSide* sd, char c, int bi){
  int total = 0;
  while (bi >= 0){
    const OBlock* ob = &sd->blk[bi]; int o = sd->counts[bi]; int n = ob->len[o];
    if (n > 0){
      if (ob->seq[o][n-1] == c){
        int ll = ob->llen[o][n-1];
        total += 1 + ll;
        if (ll != n) break;
      } else break;
    }
    bi--;
  }
  return total;
}
static int nb_right(const OSide* sd, char c, int bi){
  int total = 0;
  while (bi < 3){
    const OBlock* ob = &sd->blk[bi]; int o = sd->counts[bi]; int n = ob->len[o];
    if (n > 0){
      if (ob->seq[o][0] == c){
        int rl = ob->rlen[o][0];
        total += 1 + rl;
        if (rl != n) break;
      } else break;
    }
    bi++;
  }
  return total;
}
static int hom_len(const OSide* sd, int bi, int base){
  const OBlock* ob = &sd->blk[bi]; int o = sd->counts[bi]; int n = ob->len[o];
  int ll = ob->llen[o][base], rl = ob->rlen[o][base];
  if (base - ll == 0)   ll += nb_left(sd, ob->seq[o][base], bi-1);
  if (base + rl == n-1) rl += nb_right(sd, ob->seq[o][base], bi+1);
  return ll + rl + 1;
}

/* ---------------------------------------------- stutter block (A.3) */
typedef struct {
  int n, B, p, nins, ndel;
  const char* blk;          /* block sequence (side orientation), blk[B-1] = rightmost */
  const char* rd; const double* blc; const double* blw;   /* read side arrays, index 0..n-1 */
  double* Mt; double* Dl; double* In;   /* indexed by from-the-right offset */
  int** ups;
  double* scratch; int nscratch;
  int left_align;           /* RepeatBlock.h:29,42,51: true for the forward haplotype, false for the reversed one */
  int best_pos;             /* arg-max artifact position of the last stut_ins/stut_del call (traceback only) */
} OStut;

static double emit1(char r, char c, double lc, double lw){ return r == c ? lc : lw; }

/* StutterAlignerClass.cpp:12-53 */
static void stut_load(OStut* s){
  int n = s->n, B = s->B, p = s->p, maxdel = s->ndel*p, maxins = s->nins*p;
  int ins_i = 0, del_i = 0;
  for (int i = 0; i < n; i++){
    int e = n-1-i;          /* read index the sums end at */
    int j; double lp = 0.0;
    int lim = (n-i < maxdel ? n-i : maxdel);
    for (j = 0; j < lim; j++){
      lp += emit1(s->rd[e-j], s->blk[B-1-j], s->blc[e-j], s->blw[e-j]);
      if ((j+1) % p == 0) s->Dl[del_i++] = lp;
    }
    for (; j < maxdel; j++) if ((j+1) % p == 0) del_i++;
    int lim2 = (n-i < B ? n-i : B);
    for (; j < lim2; j++) lp += emit1(s->rd[e-j], s->blk[B-1-j], s->blc[e-j], s->blw[e-j]);
    s->Mt[i] = lp;
    double li = 0.0;
    int lim3 = (maxins < n-i ? maxins : n-i);
    for (j = 0; j < lim3; j++){
      if (j % p < B) li += emit1(s->rd[e-j], s->blk[B-1-(j%p)], s->blc[e-j], s->blw[e-j]);
      else           li += s->blc[e-j];
      if ((j+1) % p == 0) s->In[ins_i++] = li;
    }
    for (; j < maxins; j++) if ((j+1) % p == 0) s->In[ins_i++] = li;
  }
}

/* StutterAlignerClass.cpp:59-104; j = read index of the segment's right end, len = base_seq_len, off = n-1-j */
static double stut_ins(OStut* s, int len, int j, int D){
  int B = s->B, p = s->p, off = s->n-1-j, cnt = 0;
  double* v = s->scratch;
  const int* up = s->ups[0];
  double lp = -g_int_log[B+1] + s->In[s->nins*off + D/p - 1] + (len > D ? s->Mt[off+D] : 0);
  v[cnt++] = lp;
  double best = lp; s->best_pos = 0;
  int lim = len-D; if (lim < 0) lim = 0; if (lim > B) lim = B;
  int i = 0;
  for (; i > -lim; i--){
    if (-i + p < B){
      int U = up[B-1+i];
      if (U == 0){
        for (int idx = i-p; idx >= i-D; idx -= p){
          lp -= emit1(s->rd[j+idx], s->blk[B-1+i],   s->blc[j+idx], s->blw[j+idx]);
          lp += emit1(s->rd[j+idx], s->blk[B-1+i-p], s->blc[j+idx], s->blw[j+idx]);
        }
        v[cnt++] = lp;
      } else {
        v[cnt++] = g_int_log[U] + lp;
        i -= (U-1);
      }
    } else v[cnt++] = lp;
    if (lp > best || (s->left_align && lp == best)){ s->best_pos = 1-i; best = lp; }   /* StutterAlignerClass.cpp:92-95 */
  }
  if (i > -B) v[cnt++] = g_int_log[B+i] + lp;
  return oracle_fast_lse_vec(v, cnt);
}

/* StutterAlignerClass.cpp:106-150 */
static double stut_del(OStut* s, int len, int j, int D){
  int B = s->B, p = s->p, off = s->n-1-j, cnt = 0;
  double* v = s->scratch;
  const int* up = s->ups[-D/p - 1];
  double lp = -g_int_log[B+D+1];
  if (off + D >= 0)
    lp += s->Mt[off+D] - s->Dl[(off+D)*s->ndel - D/p - 1];
  else
    for (int k = 0; k > -len; k--) lp += emit1(s->rd[j+k], s->blk[B-1+k+D], s->blc[j+k], s->blw[j+k]);
  v[cnt++] = lp;
  double best = lp; s->best_pos = 0;
  int i;
  for (i = 0; i > -len; i--){
    int U = up[B-1+i];
    if (U == 0){
      lp -= emit1(s->rd[j+i], s->blk[B-1+i+D], s->blc[j+i], s->blw[j+i]);
      lp += emit1(s->rd[j+i], s->blk[B-1+i],   s->blc[j+i], s->blw[j+i]);
      v[cnt++] = lp;
    } else {
      v[cnt++] = g_int_log[U] + lp;
      i -= (U-1);
    }
    if (lp > best || (s->left_align && lp == best)){ s->best_pos = 1-i; best = lp; }   /* StutterAlignerClass.cpp:138-141 */
  }
  if (-i < B+D) v[cnt++] = g_int_log[B+D+i] + lp;
  return oracle_fast_lse_vec(v, cnt);
}

/* ---------------------------------------------- the DP (A.2) */
typedef struct {
  int n;
  const char* rd; const double* blc; const double* blw;
  double *M, *I, *D;        /* [max_size * n] row = haplotype position */
  double side_prob;
  int *art_size, *art_pos;  /* [n] best artifact size / position per read column of the STR block (traceback), or NULL */
  int left_align;
} OAln;

/* test hook: when set, align_side records the homopolymer index used for every flank row */
static int* g_dbg_h[2] = {NULL, NULL};
static int g_dbg_stop = -1;
static int g_dbg_side = 0;

/* HapAligner.cpp:26-161.  pmf = stutter pmf per STR option [nopts][13]. */
static void align_side(OSide* sd, int reuse, int last_changed, OAln* a, const double* pmf, int period){
  int n = a->n;
  double* M = a->M; double* I = a->I; double* D = a->D;
  double run = 0.0;
  char first = sd->blk[0].seq[sd->counts[0]][0];
  for (int j = 0; j < n; j++){
    M[j] = emit1(a->rd[j], first, a->blc[j], a->blw[j]) + run;
    I[j] = a->blc[j] + run;
    D[j] = IMPOSSIBLE;
    run += a->blc[j];
  }
  a->side_prob = run;
  int hi = 1, stutR = -1;
  for (int bi = 0; bi < 3; bi++){
    const OBlock* ob = &sd->blk[bi]; int o = sd->counts[bi];
    const char* bs = ob->seq[o]; int blen = ob->len[o];
    if (reuse && bi < last_changed){
      hi += blen + (bi == 0 ? -1 : 0);
      if (bi == 1) stutR = hi-1;
      continue;
    }
    if (bi == 1){
      OStut st;
      st.n = n; st.B = blen; st.p = period; st.nins = HIPSTR_MAX_STUTTER_REPS; st.ndel = sd->ndel[o];
      st.blk = bs; st.rd = a->rd; st.blc = a->blc; st.blw = a->blw; st.ups = sd->ups[o];
      st.left_align = a->left_align; st.best_pos = -1;
      st.Mt = malloc(sizeof(double)*n);
      st.Dl = malloc(sizeof(double)*n*(st.ndel > 0 ? st.ndel : 1));
      st.In = malloc(sizeof(double)*n*st.nins);
      st.scratch = malloc(sizeof(double)*(blen+8));
      stut_load(&st);
      const double* prevM = M + (size_t)n*(hi-1);
      double* rowM = M + (size_t)n*(hi+blen-1); double* rowI = I + (size_t)n*(hi+blen-1); double* rowD = D + (size_t)n*(hi+blen-1);
      for (int j = 0; j < n; j++){
        double terms[HIPSTR_NUM_ARTIFACTS]; int t = 0;
        double best_ll = IMPOSSIBLE;                      /* HapAligner.cpp:81-97 */
        if (a->art_size) a->art_size[j] = -10000;
        for (int art = -HIPSTR_MAX_STUTTER_REPS*period; art <= HIPSTR_MAX_STUTTER_REPS*period; art += period, t++){
          int len = blen+art < j+1 ? blen+art : j+1;
          int apos = -1;
          if (len >= 0){
            double pr;
            if (art == 0) pr = st.Mt[n-1-j];
            else { pr = art > 0 ? stut_ins(&st, len, j, art) : stut_del(&st, len, j, art); apos = st.best_pos; }
            double pre = (j-len < 0 ? 0 : prevM[j-len]);
            terms[t] = pmf[o*HIPSTR_NUM_ARTIFACTS + t] + pr + pre;
          } else terms[t] = IMPOSSIBLE;
          if (a->art_size && terms[t] > best_ll){ a->art_size[j] = art; a->art_pos[j] = apos; best_ll = terms[t]; }
        }
        rowM[j] = oracle_fast_lse_vec(terms, HIPSTR_NUM_ARTIFACTS);
        rowI[j] = IMPOSSIBLE; rowD[j] = IMPOSSIBLE;
      }
      free(st.Mt); free(st.Dl); free(st.In); free(st.scratch);
      stutR = hi + blen - 1;
      hi += blen;
    } else {
      for (int ci = (bi == 0 ? 1 : 0); ci < blen; ci++, hi++){
        char hc = bs[ci];
        int h1 = hom_len(sd, bi, ci), h2 = hom_len(sd, bi, ci-1 > 0 ? ci-1 : 0);
        int h = h1 > h2 ? h1 : h2; if (h > HIPSTR_MAX_HOMOP_LEN) h = HIPSTR_MAX_HOMOP_LEN;
        if (g_dbg_h[g_dbg_side]) g_dbg_h[g_dbg_side][hi] = h;
        double* m = M + (size_t)n*hi; double* ins = I + (size_t)n*hi; double* del = D + (size_t)n*hi;
        const double* pm = m - n; const double* pd = del - n;
        int after_str = (hi == stutR+1);
        m[0]   = emit1(a->rd[0], hc, a->blc[0], a->blw[0]);
        ins[0] = after_str ? IMPOSSIBLE : a->blc[0];
        del[0] = after_str ? IMPOSSIBLE : fmax(pd[0]+D2D, pm[0]+D2M);
        if (after_str){
          for (int j = 1; j < n; j++){
            m[j] = emit1(a->rd[j], hc, a->blc[j], a->blw[j]) + pm[j-1];
            ins[j] = IMPOSSIBLE; del[j] = IMPOSSIBLE;
          }
          continue;
        }
        for (int j = 1; j < n; j++){
          double c0 = ins[j-1] + g_m2i[h], c1 = pm[j-1] + g_m2m[h], c2 = pd[j-1] + g_m2d[h];
          m[j]   = emit1(a->rd[j], hc, a->blc[j], a->blw[j]) + fmax(c0, fmax(c1, c2));
          ins[j] = a->blc[j] + fmax(pm[j-1] + I2M, ins[j-1] + I2I);
          del[j] = fmax(pm[j] + D2M, pd[j] + D2D);
        }
      }
    }
  }
}

/* HapAligner.cpp:163-231 (forward haplotype = L side) */
static double combine(const OSide* fw, const OAln* L, const OAln* R, char seed_c, double seed_lw, double seed_lc, double* scratch, int* max_index){
  int nL = L->n, nR = R->n, H = 0, nseeds = 0, cnt = 0;
  for (int bi = 0; bi < 3; bi++){ int len = fw->blk[bi].len[fw->counts[bi]]; H += len; if (bi != 1) nseeds += len; }
  double prior = -g_int_log[nseeds];
  const char* b0 = fw->blk[0].seq[fw->counts[0]];
  const char* b2 = fw->blk[2].seq[fw->counts[2]]; int l2 = fw->blk[2].len[fw->counts[2]];
  scratch[cnt++] = prior + (seed_c == b0[0] ? seed_lc : seed_lw) + L->side_prob + R->M[(size_t)nR*(H-1)-1];
  scratch[cnt++] = prior + (seed_c == b2[l2-1] ? seed_lc : seed_lw) + R->side_prob + L->M[(size_t)nL*(H-1)-1];
  int mi = 0; double mll = scratch[0];                    /* HapAligner.cpp:184-193 */
  if (scratch[1] > mll){ mi = H-1; mll = scratch[1]; }
  int hap_index = 1;
  const double* lp = L->M + (nL-1);
  const double* rp = R->M + ((size_t)nR*(H-2) - 1);
  for (int bi = 0; bi < 3; bi++){
    const char* bs = fw->blk[bi].seq[fw->counts[bi]]; int blen = fw->blk[bi].len[fw->counts[bi]];
    if (bi == 1){ lp += (size_t)nL*blen; rp -= (size_t)nR*blen; hap_index += blen; continue; }
    int ci = (bi == 0 ? 1 : 0), ce = (bi == 2 ? blen-1 : blen);
    for (; ci < ce; ci++, hap_index++){
      scratch[cnt++] = prior + (seed_c == bs[ci] ? seed_lc : seed_lw) + *lp + *rp;
      if (scratch[cnt-1] > mll){ mi = hap_index; mll = scratch[cnt-1]; }   /* HapAligner.cpp:219-222 */
      lp += nL; rp -= nR;
    }
  }
  if (max_index) *max_index = mi;
  return oracle_fast_lse_vec(scratch, cnt);
}

/* ---------------------------------------------- seed (A.6) */
static void best_seed_in(int32_t rs, int32_t re, int32_t rep_s, int32_t rep_e, int32_t* best_dist, int32_t* best_pos){
  /* HapAligner.cpp:238-264 with a single repeat interval [rep_s, rep_e) */
  *best_dist = *best_pos = -1;
  int32_t pos = rs; int ri = 0;
  while (ri < 1 && pos <= re){
    if (pos < rep_s){
      int32_t e = re < rep_s-1 ? re : rep_s-1;
      int32_t dist = 1 + (e-pos)/2;
      if (dist >= *best_dist){ *best_dist = dist; *best_pos = dist-1+pos; }
      pos = rep_e; ri++;
    } else if (pos < rep_e){ pos = rep_e; ri++; }
    else ri++;
  }
  if (pos <= re){
    int32_t dist = 1 + (re-pos)/2;
    if (dist >= *best_dist){ *best_dist = dist; *best_pos = dist-1+pos; }
  }
}

/* HapAligner.cpp:270-318; returns -2 on the inputs the reference dies on */
static int seed_base(const hipstr_batch_t* b, int l, int r){
  int32_t pos = b->read_start[r];
  int best = -1, cur = 0, max_dist = MIN_SEED_DIST;
  int32_t first_start = b->blk_start[3*l], last_end = b->blk_end[3*l+2];
  int32_t rep_s = b->blk_start[3*l+1], rep_e = b->blk_end[3*l+1];
  for (int c = b->cigar_off[r]; c < b->cigar_off[r+1]; c++){
    int num = b->cigar_len[c];
    switch (b->cigar_op[c]){
      case '=': {
        int32_t lo = pos, hi = pos+num-1;
        if (lo < first_start) lo = first_start;
        if (hi > last_end-1)  hi = last_end-1;
        if (lo <= hi){
          int32_t dist, dpos;
          best_seed_in(lo, hi, rep_s, rep_e, &dist, &dpos);
          if (dist >= max_dist){ max_dist = dist; best = cur + (dpos-pos); }
        }
        pos += num; cur += num; break;
      }
      case 'I': cur += num; break;
      case 'X': pos += num; cur += num; break;
      case 'D': pos += num; break;
      default: return -2;
    }
  }
  int len = b->base_off[r+1]-b->base_off[r];
  if (best < -1 || best == 0 || best >= len-1) return -2;
  return best;
}

int oracle_calc_seed_bases(const hipstr_batch_t* b, int32_t* seeds){
  for (int l = 0; l < b->n_loci; l++)
    for (int r = b->read_off[l]; r < b->read_off[l+1]; r++){
      seeds[r] = seed_base(b, l, r);
      if (seeds[r] == -2) return 1;
    }
  return 0;
}

/* ---------------------------------------------- allele iterator (A.7) */
typedef struct { int n[3], f[3], dir[3], cnt[3], counter, ncombs, last_changed; } OIter;   /* Haplotype.cpp:123-196 */
static void iter_reset(OIter* it){
  it->ncombs = 1;
  for (int i = 0; i < 3; i++){ it->f[i] = it->ncombs; it->ncombs *= it->n[i]; it->dir[i] = 1; it->cnt[i] = 0; }
  it->counter = 0; it->last_changed = -1;
}
static int iter_next(OIter* it){
  if (it->counter == it->ncombs-1) return 0;
  int t = it->counter+1, idx = -1;
  for (int j = 2; j >= 0; j--){ t %= it->f[j]; if (t == 0){ idx = j; break; } }
  it->last_changed = idx;
  it->cnt[idx] += it->dir[idx];
  if (it->cnt[idx] == 0 || it->cnt[idx] == it->n[idx]-1) it->dir[idx] *= -1;
  it->counter++;
  return 1;
}

/* option index per block of allele k (fw block order), for tests of the Gray code */
int oracle_allele_options(const int32_t* nopts, int k, int32_t* opts){
  OIter it; for (int i = 0; i < 3; i++) it.n[i] = nopts[i];
  iter_reset(&it);
  if (k < 0 || k >= it.ncombs) return 1;
  while (it.counter < k) iter_next(&it);
  for (int i = 0; i < 3; i++) opts[i] = it.cnt[i];
  return 0;
}

/* ---------------------------------------------- process_reads (A.1) */
static int process_reads_impl(const hipstr_batch_t* b, const int32_t* seed_in, double* aln_probs, int32_t* seeds);
int oracle_process_reads(const hipstr_batch_t* b, double* aln_probs, int32_t* seeds){ return process_reads_impl(b, NULL, aln_probs, seeds); }
/* HapAligner::process_read with the caller's seed_base (HapAligner.h:83, HapAligner.cpp:573-709): seed_in[r] >= 0 replaces
 * calc_seed_base for read r; HIPSTR_SEED_AUTO (-2) computes it. */
int oracle_process_reads_seeded(const hipstr_batch_t* b, const int32_t* seed_in, double* aln_probs, int32_t* seeds){
  return process_reads_impl(b, seed_in, aln_probs, seeds);
}
static int process_reads_impl(const hipstr_batch_t* b, const int32_t* seed_in, double* aln_probs, int32_t* seeds){
  oracle_init();
  int opt_cursor = 0;
  int64_t out_off = 0;
  for (int l = 0; l < b->n_loci; l++){
    int period = b->period[l];
    OSide fw, rv;
    side_build(&fw, b, l, opt_cursor, 0);
    side_build(&rv, b, l, opt_cursor, 1);
    OIter it;
    for (int k = 0; k < 3; k++){ it.n[k] = b->blk_nopts[3*l+k]; opt_cursor += it.n[k]; }
    iter_reset(&it);
    int A = it.ncombs;
    if (A != b->hap_off[l+1]-b->hap_off[l]){ side_free(&fw); side_free(&rv); return 2; }
    /* stutter pmf per STR option (RepeatStutterInfo.h:53-61): identical for both orientations */
    int nso = fw.blk[1].nopts;
    double* pmf = malloc(sizeof(double)*nso*HIPSTR_NUM_ARTIFACTS);
    for (int o = 0; o < nso; o++){
      int size = fw.blk[1].len[o], t = 0;
      for (int art = -HIPSTR_MAX_STUTTER_REPS*period; art <= HIPSTR_MAX_STUTTER_REPS*period; art += period, t++)
        pmf[o*HIPSTR_NUM_ARTIFACTS+t] = (size+art < 0) ? LARGE_NEGATIVE : oracle_stutter_pmf(b->stutter+6*l, period, size, size+art);
    }
    int Hmax = fw.max_size;
    for (int r = b->read_off[l]; r < b->read_off[l+1]; r++){
      double* out = aln_probs + out_off + (int64_t)(r-b->read_off[l])*A;
      if (b->realign_read && !b->realign_read[r]) continue;
      int sb = (seed_in && seed_in[r] != -2) ? seed_in[r] : seed_base(b, l, r);
      if (sb == -2 || (seed_in && seed_in[r] != -2 && sb != -1 && (sb < 1 || sb > b->base_off[r+1]-b->base_off[r]-2))){ free(pmf); side_free(&fw); side_free(&rv); return 1; }
      seeds[r] = sb;
      if (sb == -1){ for (int k = 0; k < A; k++) out[k] = 0; continue; }     /* HapAligner.cpp:333-337 */
      int len = b->base_off[r+1]-b->base_off[r];
      const char* bases = b->bases + b->base_off[r]; const char* quals = b->quals + b->base_off[r];
      double* lw = malloc(sizeof(double)*len); double* lc = malloc(sizeof(double)*len);
      for (int j = 0; j < len; j++){ lw[j] = g_q_error[qual_index(quals[j])]; lc[j] = g_q_correct[qual_index(quals[j])]; }
      int nL = sb, nR = len-sb-1;
      char* rrd = malloc(nR+1); double* rlw = malloc(sizeof(double)*nR); double* rlc = malloc(sizeof(double)*nR);
      for (int j = 0; j < nR; j++){ rrd[j] = bases[len-1-j]; rlw[j] = lw[len-1-j]; rlc[j] = lc[len-1-j]; }
      OAln L, R;
      L.n = nL; L.rd = bases; L.blc = lc; L.blw = lw; L.art_size = L.art_pos = NULL; L.left_align = 1;
      R.n = nR; R.rd = rrd;  R.blc = rlc; R.blw = rlw; R.art_size = R.art_pos = NULL; R.left_align = 0;
      L.M = malloc(sizeof(double)*(size_t)nL*Hmax); L.I = malloc(sizeof(double)*(size_t)nL*Hmax); L.D = malloc(sizeof(double)*(size_t)nL*Hmax);
      R.M = malloc(sizeof(double)*(size_t)nR*Hmax); R.I = malloc(sizeof(double)*(size_t)nR*Hmax); R.D = malloc(sizeof(double)*(size_t)nR*Hmax);
      double* scratch = malloc(sizeof(double)*(Hmax+4));
      iter_reset(&it);
      int reuse = 0;
      do {
        if (b->realign_hap && !b->realign_hap[b->hap_off[l]+it.counter]){ reuse = 0; continue; }
        for (int k = 0; k < 3; k++){ fw.counts[k] = it.cnt[k]; rv.counts[k] = it.cnt[2-k]; }
        int lc_fw = it.last_changed, lc_rv = it.last_changed < 0 ? -1 : 2-it.last_changed;
        g_dbg_side = 0; align_side(&fw, reuse, lc_fw, &L, pmf, period);
        g_dbg_side = 1; align_side(&rv, reuse, lc_rv, &R, pmf, period);
        if (g_dbg_h[0] && it.counter == g_dbg_stop){ g_dbg_h[0] = g_dbg_h[1] = NULL; }
        out[it.counter] = combine(&fw, &L, &R, bases[sb], lw[sb], lc[sb], scratch, NULL);
        reuse = 1;
      } while (iter_next(&it));
      free(scratch);
      free(L.M); free(L.I); free(L.D); free(R.M); free(R.I); free(R.D);
      free(rrd); free(rlw); free(rlc); free(lw); free(lc);
    }
    out_off += (int64_t)(b->read_off[l+1]-b->read_off[l])*A;
    free(pmf); side_free(&fw); side_free(&rv);
  }
  return 0;
}

/* ---------------------------------------------- posteriors (A.8, a15/a16) */
int oracle_posteriors(const hipstr_post_batch_t* pb, double* log_post, double* sample_total_ll, int32_t* map_gt, double* locus_total_ll){
  oracle_init();
  int64_t post_off = 0, samp_off = 0, ll_off = 0;
  for (int l = 0; l < pb->n_loci; l++){
    int A = pb->n_alleles[l], S = pb->n_samples[l], nd = A*A;
    int hap = pb->haploid ? pb->haploid[l] : 0;
    /* genotyper.cpp:20-42 */
    double hom = hap ? -g_int_log[A] : g_int_log[2] - g_int_log[A] - g_int_log[A+1];
    double het = hap ? -DBL_MAX/2 : -g_int_log[A] - g_int_log[A+1];
    double* post = log_post + post_off;
    for (int s = 0; s < S; s++) for (int i = 0; i < A; i++) for (int j = 0; j < A; j++)
      post[(size_t)s*nd + i*A + j] = pb->log_prior ? pb->log_prior[post_off + (size_t)s*nd + i*A + j] : (i == j ? hom : het);   /* virtual init_log_sample_priors, genotyper.h:69 */
    /* genotyper.cpp:49-61 */
    for (int r = pb->read_off[l]; r < pb->read_off[l+1]; r++){
      const double* LL = pb->log_aln_probs + ll_off + (int64_t)(r-pb->read_off[l])*A;
      double* sp = post + (size_t)pb->sample_label[r]*nd;
      for (int i = 0; i < A; i++) for (int j = 0; j < A; j++)
        sp[i*A+j] += pb->read_weight[r]*oracle_fast_lse2(g_log_half + pb->log_p1[r] + LL[i], g_log_half + pb->log_p2[r] + LL[j]);
    }
    /* genotyper.cpp:63-72, 82-97 */
    double total = 0.0;
    for (int s = 0; s < S; s++){
      double* sp = post + (size_t)s*nd;
      double tot = oracle_log_sum_exp(sp, nd);
      sample_total_ll[samp_off+s] = tot;
      for (int i = 0; i < nd; i++) sp[i] -= tot;
      total += tot;
      double best = -DBL_MAX; int g1 = -1, g2 = -1;
      for (int i = 0; i < A; i++) for (int j = 0; j < A; j++) if (sp[i*A+j] > best){ best = sp[i*A+j]; g1 = i; g2 = j; }
      map_gt[2*(samp_off+s)] = g1; map_gt[2*(samp_off+s)+1] = g2;
    }
    locus_total_ll[l] = total;
    post_off += (int64_t)S*nd; samp_off += S; ll_off += (int64_t)(pb->read_off[l+1]-pb->read_off[l])*A;
  }
  return 0;
}

/* test hook: homopolymer index per matrix row (both sides) as in effect when allele k of a one-locus,
 * one-read batch is scored (rows of blocks that were reused keep the values of the allele they were computed under) */
/* ---------------------------------------------- Needleman-Wunsch (A.11)
 * NeedlemanWunsch::Align (NeedlemanWunsch.cpp:370-420) and its helpers; float arithmetic as in the reference. */
static int nw_base(char c){                                       /* base_to_int (NeedlemanWunsch.cpp:98-118) */
  switch (c){ case 'A': case 'a': return 0; case 'C': case 'c': return 1; case 'G': case 'g': return 2; case 'T': case 't': return 3; default: return 4; }
}
static float nw_best(float s1, float s2, float s3, int* c){        /* bestIndex (:120-140) */
  if (s2 > s1){ if (s2 > s3){ *c = 1; return s2; } *c = 2; return s3; }
  if (s3 > s1){ *c = 2; return s3; }
  *c = 0; return s1;
}
int oracle_nw_align(const hipstr_nw_batch_t* nb, hipstr_nw_out_t* o){
  const float MATCH = 2.0f, MISMATCH = -2.0f, GAPOPEN = 5.0f, GAPEXTEND = 0.125f, LARGE = 1000000.0f;
  o->aln_off[0] = 0; o->cigar_off[0] = 0;
  for (int p = 0; p < nb->n_pairs; p++){
    const char* ref = nb->ref_seqs + nb->ref_off[p]; int L1 = nb->ref_off[p+1] - nb->ref_off[p];
    const char* rd = nb->read_seqs + nb->read_off[p]; int L2 = nb->read_off[p+1] - nb->read_off[p];
    size_t W = (size_t)L1 + 1, sz = W*((size_t)L2 + 1);
    float* M = malloc(sizeof(float)*sz); float* R = malloc(sizeof(float)*sz); float* D = malloc(sizeof(float)*sz);
    int8_t* tM = malloc(sz); int8_t* tR = malloc(sz); int8_t* tD = malloc(sz);
    M[0] = 0.0f; R[0] = -LARGE; D[0] = -LARGE;                       /* initMatrices (:326-367) */
    for (int j = 1; j <= L1; j++){
      R[j] = !nb->use_ref_end_penalty ? 0.0f : -GAPOPEN - (j-1)*GAPEXTEND; tR[j] = 1;
      D[j] = -LARGE; tD[j] = -1; M[j] = -LARGE; tM[j] = -1;
    }
    for (int i = 1; i <= L2; i++){
      size_t x = (size_t)i*W;
      D[x] = -GAPOPEN - (i-1)*GAPEXTEND; tD[x] = 2; R[x] = -LARGE; tR[x] = -1; M[x] = -LARGE; tM[x] = -1;
    }
    for (int i = 1; i <= L2; i++){                                   /* nw_helper (:195-245) */
      int rb = nw_base(rd[i-1]);
      for (int j = 1; j <= L1; j++){
        size_t n = (size_t)i*W + j, od = n - W - 1, ol = n - 1, ou = n - W;
        int fb = nw_base(ref[j-1]), c;
        float sc = (fb == 4 || rb == 4 || fb == rb) ? MATCH : MISMATCH;
        M[n] = nw_best(M[od], R[od], D[od], &c) + sc; tM[n] = (int8_t)c;
        R[n] = nw_best(M[ol] - GAPOPEN, R[ol] - GAPEXTEND, D[ol] - GAPOPEN, &c); tR[n] = (int8_t)c;
        D[n] = nw_best(M[ou] - GAPOPEN, R[ou] - GAPOPEN, D[ou] - GAPEXTEND, &c); tD[n] = (int8_t)c;
      }
    }
    float best_val; int best_col, best_type;
    if (nb->use_ref_end_penalty){                                    /* findOptimalStopEndPenalty (:173-193) */
      size_t x = sz - 1; best_col = L1; best_val = M[x]; best_type = 0;
      if (R[x] > best_val){ best_val = R[x]; best_type = 1; }
      if (D[x] > best_val){ best_val = D[x]; best_type = 2; }
    } else {                                                         /* findOptimalStop (:142-171) */
      best_val = -LARGE; best_col = -1; best_type = -1;
      for (int j = 0; j <= L1; j++){
        size_t x = (size_t)L2*W + j;
        if (M[x] >= best_val){ best_val = M[x]; best_col = j; best_type = 0; }
        if (R[x] > best_val){ best_val = R[x]; best_col = j; best_type = 1; }
        if (D[x] > best_val){ best_val = D[x]; best_col = j; best_type = 2; }
      }
    }
    o->score[p] = best_val;
    /* traceAlignment (:247-324): built back to front */
    size_t cap = (size_t)L1 + L2 + 2; char* ra = malloc(cap); char* qa = malloc(cap); char* raw = malloc(cap);
    size_t na = 0, nr = 0;
    for (int j = L1; j > best_col; j--){ ra[na] = ref[j-1]; qa[na++] = '-'; }
    int row = L2, col = best_col, type = best_type, bad = 0;
    while (row > 0){
      size_t x = (size_t)row*W + col;
      if (type == 0){
        ra[na] = ref[col-1]; qa[na++] = rd[row-1];
        raw[nr++] = nw_base(ref[col-1]) == nw_base(rd[row-1]) ? '=' : 'X';
        type = tM[x]; row--; col--;
      } else if (type == 1){ ra[na] = ref[col-1]; qa[na++] = '-'; raw[nr++] = 'D'; type = tR[x]; col--; }
      else if (type == 2){ ra[na] = '-'; qa[na++] = rd[row-1]; raw[nr++] = 'I'; type = tD[x]; row--; }
      else { bad = 1; break; }
    }
    for (int j = col; j > 0; j--){ ra[na] = ref[j-1]; qa[na++] = '-'; }
    int rc = 0;
    if (bad || o->aln_off[p] + (int64_t)na > o->cap_aln) rc = 3;
    else {
      for (size_t k = 0; k < na; k++){ o->ref_al[o->aln_off[p] + k] = ra[na-1-k]; o->read_al[o->aln_off[p] + k] = qa[na-1-k]; }
      o->aln_off[p+1] = o->aln_off[p] + (int64_t)na;
      int64_t co = o->cigar_off[p];
      for (size_t k = nr; k > 0; ){                                  /* run-length encode the reversed raw string */
        char ch = raw[k-1]; int num = 0;
        while (k > 0 && raw[k-1] == ch){ num++; k--; }
        if (co >= o->cap_cigar){ rc = 3; break; }
        o->cigar_op[co] = ch; o->cigar_len[co++] = num;
      }
      o->cigar_off[p+1] = co;
      o->ok[p] = 1;                                                  /* the CIGAR never holds 'S' (:413-415) */
    }
    free(M); free(R); free(D); free(tM); free(tR); free(tD); free(ra); free(qa); free(raw);
    if (rc) return rc;
  }
  return 0;
}

/* Haplotype::aln_haps_to_ref + adjust_indels (Haplotype.cpp:8-86) -> get_aln_info() of every haplotype */
int oracle_hap_aln_info(const hipstr_batch_t* b, char* out, int64_t out_cap, int64_t* offs){
  int64_t pos = 0; int hi = 0, opt_cursor = 0;
  offs[0] = 0;
  for (int l = 0; l < b->n_loci; l++){
    const int32_t* nopts = b->blk_nopts + 3*l;
    int base[3]; for (int k = 0; k < 3; k++){ base[k] = opt_cursor; opt_cursor += nopts[k]; }
    int A = nopts[0]*nopts[1]*nopts[2];
    int rl = 0; for (int k = 0; k < 3; k++) rl += b->opt_off[base[k]+1] - b->opt_off[base[k]];
    char* ref = malloc(rl + 1); int rp = 0;
    for (int k = 0; k < 3; k++){ int n = b->opt_off[base[k]+1] - b->opt_off[base[k]]; memcpy(ref + rp, b->seq + b->opt_off[base[k]], n); rp += n; }
    for (int k = 0; k < A; k++, hi++){
      int32_t oi[3]; oracle_allele_options(nopts, k, oi);
      int al = 0; for (int x = 0; x < 3; x++) al += b->opt_off[base[x]+oi[x]+1] - b->opt_off[base[x]+oi[x]];
      char* alt = malloc(al + 1); int ap = 0;
      for (int x = 0; x < 3; x++){ int o = base[x]+oi[x], n = b->opt_off[o+1] - b->opt_off[o]; memcpy(alt + ap, b->seq + b->opt_off[o], n); ap += n; }
      int32_t ro[2] = { 0, rl }, ao[2] = { 0, al };
      hipstr_nw_batch_t nb = { 1, ro, ref, ao, alt, 1 };
      int64_t cap = (int64_t)rl + al + 8;
      float score; uint8_t ok; int64_t aoff[2], coff[2];
      char* ra = malloc(cap); char* qa = malloc(cap); char* cop = malloc(cap); int32_t* cl = malloc(sizeof(int32_t)*cap);
      hipstr_nw_out_t o = { &score, &ok, aoff, ra, qa, coff, cop, cl, cap, cap };
      int rc = oracle_nw_align(&nb, &o);
      int64_t n = aoff[1];
      if (rc == 0){                                                  /* adjust_indels (Haplotype.cpp:8-56) */
        int32_t ref_pos = b->blk_start[3*l], str_pos = b->blk_start[3*l+1];
        int64_t x = 0;
        while (x < n){
          if (qa[x] == '-' && ref_pos < str_pos){
            int64_t idx = x; while (idx < n && qa[idx] == '-') idx++;
            int32_t p2 = ref_pos; int64_t di = x; int32_t dsz = (int32_t)(idx - x);
            while (idx < n && p2 < str_pos && ra[di] == ra[idx]){ qa[di] = qa[idx]; qa[idx] = '-'; idx++; di++; p2++; }
            x = idx; ref_pos = p2 + dsz;
          } else if (ra[x] == '-' && ref_pos < str_pos){
            int64_t idx = x; while (idx < n && ra[idx] == '-') idx++;
            int32_t p2 = ref_pos; int64_t ii = x;
            while (idx < n && p2 < str_pos && qa[ii] == qa[idx]){ ra[ii] = ra[idx]; ra[idx] = '-'; idx++; ii++; p2++; }
            x = idx; ref_pos = p2;
          } else { if (ra[x] != '-') ref_pos++; x++; }
        }
        if (pos + n + 1 > out_cap) rc = 2;
        else {
          offs[hi] = pos;
          for (int64_t y = 0; y < n; y++) out[pos++] = ra[y] == '-' ? 'I' : (qa[y] == '-' ? 'D' : 'M');
          out[pos++] = 0;
        }
      }
      free(alt); free(ra); free(qa); free(cop); free(cl);
      if (rc){ free(ref); return rc; }
    }
    free(ref);
  }
  offs[hi] = pos;
  return 0;
}

/* ---------------------------------------------- de novo stutter EM (A.10)
 * EMStutterGenotyper (em_stutter_genotyper.h:55-102, em_stutter_genotyper.cpp:10-226), one locus at a time. */
static void stream_update(double lv, double* mx, double* tot);
static double exact_lse2(double a, double b);
typedef struct { double in_geom, in_up, in_down, out_geom, out_up, out_down; } OModel;
static int cmp_int(const void* a, const void* b){ int x = *(const int*)a, y = *(const int*)b; return (x > y) - (x < y); }

int oracle_em_train(const hipstr_em_batch_t* eb, uint8_t* trained, double* stutter, int32_t* n_iter, double* final_ll){
  oracle_init();
  for (int l = 0; l < eb->n_loci; l++){
    int S = eb->n_samples[l], r0 = eb->read_off[l], R = eb->read_off[l+1] - r0, p = eb->period[l];
    int hap = eb->haploid && eb->haploid[l];
    const int32_t* lab = eb->sample_label + r0; const int32_t* nb = eb->num_bps + r0;
    const double* lp1 = eb->log_p1 + r0; const double* lp2 = eb->log_p2 + r0;
    /* alleles: the reference size first, the other distinct sizes ascending (em_stutter_genotyper.h:59-78) */
    int* sizes = malloc(sizeof(int)*(R+1)); int A = 0;
    for (int r = 0; r < R; r++) if (nb[r] != eb->ref_allele) sizes[A++] = nb[r];
    qsort(sizes, A, sizeof(int), cmp_int);
    int u = 0; for (int i = 0; i < A; i++) if (i == 0 || sizes[i] != sizes[i-1]) sizes[u++] = sizes[i];
    A = u + 1;
    int* bps = malloc(sizeof(int)*A); bps[0] = eb->ref_allele; memcpy(bps+1, sizes, sizeof(int)*u);
    const size_t Rn = (size_t)(R > 0 ? R : 1);
    int* ai = malloc(sizeof(int)*Rn);
    for (int r = 0; r < R; r++) for (int a = 0; a < A; a++) if (bps[a] == nb[r]){ ai[r] = a; break; }
    int* per_sample = calloc(S, sizeof(int)); for (int r = 0; r < R; r++) per_sample[lab[r]]++;
    double* gtp = malloc(sizeof(double)*A);
    double* post = malloc(sizeof(double)*(size_t)S*A*A); double* prior = malloc(sizeof(double)*(size_t)S*A*A);
    double* LLm = malloc(sizeof(double)*Rn*A); double* totals = malloc(sizeof(double)*S);
    double* phase = malloc(sizeof(double)*Rn*A*A*2);
    int32_t* mapgt = malloc(sizeof(int32_t)*2*S); int32_t* w = malloc(sizeof(int32_t)*Rn);
    for (int r = 0; r < R; r++) w[r] = 1;
    /* init_log_gt_priors (:10-20) */
    for (int a = 0; a < A; a++) gtp[a] = 1;
    for (int r = 0; r < R; r++) gtp[ai[r]] += 1.0/per_sample[lab[r]];
    { double tot = 0; for (int a = 0; a < A; a++) tot += gtp[a]; double lt = X_LOG(tot); for (int a = 0; a < A; a++) gtp[a] = X_LOG(gtp[a]) - lt; }
    OModel m = { 0.9, 0.1, 0.1, 0.8, 0.01, 0.01 };                       /* init_stutter_model (:59-62) */
    int it = 1, ok = 0, done = 0; double LL = -DBL_MAX, new_LL = 0;
    int32_t one_A = A, one_S = S, roff[2] = { 0, R }; uint8_t hp = (uint8_t)hap;
    while (it <= eb->max_iter && !done){
      double sp[6] = { m.in_geom, m.in_up, m.in_down, m.out_geom, m.out_up, m.out_down };
      /* E-step: calc_hap_aln_probs (:146-150), priors (:129-144), posteriors, read phase posteriors (:152-169) */
      for (int r = 0; r < R; r++) for (int a = 0; a < A; a++) LLm[(size_t)r*A + a] = oracle_stutter_pmf(sp, p, bps[a], bps[ai[r]]);
      for (int s = 0; s < S; s++) for (int i1 = 0; i1 < A; i1++) for (int i2 = 0; i2 < A; i2++)
        prior[((size_t)s*A + i1)*A + i2] = !hap ? gtp[i1] + gtp[i2] : (i1 == i2 ? gtp[i1] : -DBL_MAX/2);
      hipstr_post_batch_t pb = { 1, &one_A, &one_S, roff, lab, lp1, lp2, w, LLm, &hp, prior };
      double ltot;
      oracle_posteriors(&pb, post, totals, mapgt, &ltot);
      new_LL = ltot;
      for (int r = 0; r < R; r++) for (int i1 = 0; i1 < A; i1++) for (int i2 = 0; i2 < A; i2++){
        double one = g_log_half + lp1[r] + oracle_stutter_pmf(sp, p, bps[i1], bps[ai[r]]);
        double two = g_log_half + lp2[r] + oracle_stutter_pmf(sp, p, bps[i2], bps[ai[r]]);
        double tot = oracle_fast_lse2(one, two);
        double* ph = phase + (((size_t)r*A + i1)*A + i2)*2;
        ph[0] = one - tot; ph[1] = two - tot;
      }
      if (new_LL < LL + 1e-10){ ok = 1; break; }                          /* :190-194 */
      /* M-step: recalc_log_gt_priors (:22-57) */
      {
        double* mx = malloc(sizeof(double)*A); double* tt = malloc(sizeof(double)*A);
        for (int a = 0; a < A; a++){ mx[a] = -DBL_MAX/2; tt[a] = 0; }
        for (int s = 0; s < S; s++) for (int i1 = 0; i1 < A; i1++)
          stream_update(oracle_log_sum_exp(post + ((size_t)s*A + i1)*A, A), &mx[i1], &tt[i1]);
        for (int s = 0; s < S; s++) for (int i1 = 0; i1 < A; i1++) for (int i2 = 0; i2 < A; i2++)
          stream_update(post[((size_t)s*A + i1)*A + i2], &mx[i2], &tt[i2]);
        for (int a = 0; a < A; a++) gtp[a] = mx[a] + X_LOG(tt[a]);
        double lt = oracle_log_sum_exp(gtp, A);
        for (int a = 0; a < A; a++) gtp[a] -= lt;
        free(mx); free(tt);
      }
      /* recalc_stutter_model (:64-127) */
      OModel prev = m;
      {
        size_t cap = (size_t)R*A*A*2 + 4;
        double* v[7]; size_t n[7];                   /* in_up, in_down, in_eq, in_diffs, out_up, out_down, out_diffs */
        for (int k = 0; k < 7; k++){ v[k] = malloc(sizeof(double)*cap); n[k] = 0; }
        v[0][n[0]++] = 0.0; v[1][n[1]++] = 0.0; v[3][n[3]++] = 0.0; v[3][n[3]++] = X_LOG(1.1);
        v[4][n[4]++] = 0.0; v[5][n[5]++] = 0.0; v[6][n[6]++] = 0.0; v[6][n[6]++] = X_LOG(1.1);
        v[2][n[2]++] = 0.0;
        for (int r = 0; r < R; r++){
          const double* gp = post + (size_t)lab[r]*A*A;
          for (int i1 = 0; i1 < A; i1++) for (int i2 = 0; i2 < A; i2++) for (int ph = 0; ph < 2; ph++){
            int gi = ph == 0 ? i1 : i2;
            int bd = bps[ai[r]] - bps[gi];
            double f = gp[(size_t)i1*A + i2] + phase[(((size_t)r*A + i1)*A + i2)*2 + ph];
            if (bd == 0) v[2][n[2]++] = f;
            else if (bd % p != 0){
              int eff = bd - bd/p;
              v[6][n[6]++] = f + g_int_log[abs(eff)];
              if (bd > 0) v[4][n[4]++] = f; else v[5][n[5]++] = f;
            } else {
              int eff = bd/p;
              v[3][n[3]++] = f + g_int_log[abs(eff)];
              if (bd > 0) v[0][n[0]++] = f; else v[1][n[1]++] = f;
            }
          }
        }
        double t[7];
        for (int k = 0; k < 7; k++){ t[k] = oracle_fast_lse_vec(v[k], (int)n[k]); free(v[k]); }
        double out_total = oracle_fast_lse2(t[4], t[5]);
        double in_pgeom = fmin(0.999, X_EXP(exact_lse2(t[0], t[1]) - t[3]));
        double out_pgeom = fmin(0.999, X_EXP(out_total - t[6]));
        double mx3 = fmax(fmax(t[0], t[1]), t[2]);
        double lse3 = mx3 + X_LOG(X_EXP(t[0]-mx3) + X_EXP(t[1]-mx3) + X_EXP(t[2]-mx3));       /* mathops.cpp:59-62 */
        double log_total = exact_lse2(lse3, out_total);
        m.in_geom = in_pgeom; m.in_up = X_EXP(t[0] - log_total); m.in_down = X_EXP(t[1] - log_total);
        m.out_geom = out_pgeom; m.out_up = X_EXP(t[4] - log_total); m.out_down = X_EXP(t[5] - log_total);
      }
      double abs_change = new_LL - LL, frac_change = -(new_LL - LL)/LL;
      int conv = 0;
      if (abs_change < eb->min_ll_abs_change && frac_change < eb->min_ll_frac_change) conv = 1;
      else if (fabs(prev.in_geom - m.in_geom) < 0.0001 && fabs(prev.in_up - m.in_up) < 0.0001 && fabs(prev.in_down - m.in_down) < 0.0001 &&
               fabs(prev.out_geom - m.out_geom) < 0.0001 && fabs(prev.out_up - m.out_up) < 0.0001 && fabs(prev.out_down - m.out_down) < 0.0001) conv = 1;
      if (conv){ ok = 1; done = 1; break; }
      LL = new_LL; it++;
    }
    trained[l] = (uint8_t)ok;
    n_iter[l] = it <= eb->max_iter ? it : eb->max_iter;
    final_ll[l] = new_LL;
    stutter[6*l] = m.in_geom; stutter[6*l+1] = m.in_up; stutter[6*l+2] = m.in_down; stutter[6*l+3] = m.out_geom; stutter[6*l+4] = m.out_up; stutter[6*l+5] = m.out_down;
    free(sizes); free(bps); free(ai); free(per_sample); free(gtp); free(post); free(prior); free(LLm); free(totals); free(phase); free(mapgt); free(w);
  }
  return 0;
}

/* ---------------------------------------------- genotype calls (A.9)
 * Genotyper::extract_genotypes_and_likelihoods (genotyper.cpp:129-251), calc_PLs (99-104), calc_gl_diff (106-127). */
static void stream_update(double lv, double* mx, double* tot){          /* mathops.cpp:72-80 */
  if (lv <= *mx) *tot += X_EXP(lv - *mx);
  else { *tot *= X_EXP(*mx - lv); *tot += 1.0; *mx = lv; }
}
static double exact_lse2(double a, double b){                           /* mathops.cpp:52-57 */
  return a > b ? a + X_LOG(1 + X_EXP(b - a)) : b + X_LOG(1 + X_EXP(a - b));
}

int oracle_gt_extract(const hipstr_post_batch_t* pb, const hipstr_gt_request_t* rq, hipstr_gt_out_t* o){
  oracle_init();
  const double LOG_E_BASE_10 = 0.4342944819, TOL = 1e-10;               /* mathops.cpp:10-11 */
  int64_t n_post = 0; int n_samp = 0;
  for (int l = 0; l < pb->n_loci; l++){ n_post += (int64_t)pb->n_samples[l]*pb->n_alleles[l]*pb->n_alleles[l]; n_samp += pb->n_samples[l]; }
  double* post = malloc(sizeof(double)*(n_post ? n_post : 1));
  double* totals = malloc(sizeof(double)*(n_samp ? n_samp : 1));
  double* ltot = malloc(sizeof(double)*(pb->n_loci ? pb->n_loci : 1));
  int rc = oracle_posteriors(pb, post, totals, o->best_hap, ltot);
  int calc_any = rq->calc_gls || rq->calc_pls || rq->calc_phased_gls;
  int64_t po = 0, g = 0, pg = 0; int so = 0, map_off = 0;
  for (int l = 0; l < pb->n_loci && rc == 0; l++){
    int A = pb->n_alleles[l], S = pb->n_samples[l], V = rq->n_variants[l];
    int hap = pb->haploid && pb->haploid[l];
    const int32_t* h2a = rq->hap_to_allele + map_off;
    double hom = hap ? -g_int_log[A] : g_int_log[2] - g_int_log[A] - g_int_log[A+1];
    double het = hap ? 0 : -g_int_log[A] - g_int_log[A+1];
    double gl_ncfg  = hap ? g_int_log[2] + g_int_log[A] - g_int_log[V] : g_int_log[2] + 2*(g_int_log[A] - g_int_log[V]);
    double pgl_ncfg = hap ? g_int_log[A] - g_int_log[V] : 2*(g_int_log[A] - g_int_log[V]);
    double* mx = malloc(sizeof(double)*V*V); double* T = malloc(sizeof(double)*V*V);
    int ngl = hap ? V : V*(V+1)/2, npgl = hap ? V : V*V;
    double* gls = malloc(sizeof(double)*ngl);
    for (int s = 0; s < S; s++, so++){
      const double* P = post + po + (int64_t)s*A*A;
      for (int i = 0; i < V*V; i++){ mx[i] = -DBL_MAX/2; T[i] = 0.0; }
      for (int i1 = 0; i1 < A; i1++)
        for (int i2 = 0; i2 < A; i2++){ int gi = V*h2a[i1] + h2a[i2]; stream_update(P[(int64_t)i1*A + i2], &mx[gi], &T[gi]); }
      for (int i = 0; i < V*V; i++) T[i] = mx[i] + X_LOG(T[i]);
      int ha = o->best_hap[2*so], hb = o->best_hap[2*so+1];
      int ga = h2a[ha], gb = h2a[hb];
      o->best_gt[2*so] = ga; o->best_gt[2*so+1] = gb;
      double pab = P[(int64_t)ha*A + hb], pba = P[(int64_t)hb*A + ha];
      o->hap_log_phased_post[so] = pab;
      o->hap_log_unphased_post[so] = (ha != hb) ? oracle_fast_lse2(pab, pba) : pab;
      double lp = T[V*ga + gb];
      o->log_phased_post[so] = lp;
      o->log_unphased_post[so] = (ga == gb) ? lp : exact_lse2(lp, T[V*gb + ga]);
      if (!calc_any) continue;
      int k = 0, pk = 0;
      for (int i1 = 0; i1 < V; i1++)
        for (int i2 = 0; i2 < V; i2++){
          int gi = i1*V + i2, alt = i2*V + i1;
          double corr = (i1 == i2) ? hom : het;
          if (i2 <= i1 && (!hap || i1 == i2))
            gls[k++] = (totals[so] - (corr + gl_ncfg) + oracle_fast_lse2(T[gi], T[alt]))*LOG_E_BASE_10;
          if (rq->calc_phased_gls && (!hap || i1 == i2))
            o->phased_gls[pg + pk++] = (totals[so] - (corr + pgl_ncfg) + T[gi])*LOG_E_BASE_10;
        }
      double max_gl = gls[0], second = -DBL_MAX;
      for (int i = 1; i < ngl; i++) if (gls[i] > max_gl) max_gl = gls[i];
      for (int i = 0; i < ngl; i++) if (gls[i] < max_gl && gls[i] > second) second = gls[i];
      if (second == -DBL_MAX) second = max_gl;
      if (A == 1) o->gl_diff[so] = -1000;
      else {
        int lo = ga < gb ? ga : gb, hi = ga < gb ? gb : ga;
        int gi = hap ? ga : hi*(hi+1)/2 + lo;
        o->gl_diff[so] = (fabs(max_gl - gls[gi]) < TOL) ? (max_gl - second) : gls[gi] - max_gl;
      }
      for (int i = 0; i < ngl; i++){
        if (rq->calc_gls) o->gls[g + i] = gls[i];
        if (rq->calc_pls){ int pl = (int)(-10*(gls[i] - max_gl)); o->pls[g + i] = pl < 999 ? pl : 999; }
      }
      g += ngl; pg += npgl;
    }
    free(mx); free(T); free(gls);
    po += (int64_t)S*A*A; map_off += A;
  }
  free(post); free(totals); free(ltot);
  return rc;
}

int oracle_debug_row_h(const hipstr_batch_t* b, int k, int32_t* h_fw, int32_t* h_rv, int cap){
  for (int i = 0; i < cap; i++){ h_fw[i] = 0; h_rv[i] = 0; }
  g_dbg_h[0] = h_fw; g_dbg_h[1] = h_rv; g_dbg_stop = k;
  int64_t n = (int64_t)b->read_off[b->n_loci] * 4096;
  double* probs = malloc(sizeof(double)*(n > 0 ? n : 1)); int32_t* seeds = malloc(sizeof(int32_t)*(b->read_off[b->n_loci]+1));
  int rc = oracle_process_reads(b, probs, seeds);
  g_dbg_h[0] = g_dbg_h[1] = NULL;
  free(probs); free(seeds);
  return rc;
}

/* ====================================================================== traceback (A.12)
 * HapAligner::trace_optimal_aln (HapAligner.cpp:711-722) -> process_read(retrace_aln = true) on ONE fixed haplotype
 * (HapAligner.cpp:573-709), HapAligner::retrace (363-571) and stitch_alignment_trace (AlignmentTraceback.cpp:55-144).
 * Restated on the flat batch; output flattened into hipstr_trace_out_t.                                            */
#define TRACE_LL_TOL 0.001                                   /* HapAligner.cpp:345 */
#define MIN_SNP_LOG_PROB_CORRECT (-0.0043648054)             /* HapAligner.cpp:24 */

typedef struct { char* s; int n, cap; } OStr;
static void ostr_init(OStr* o, int cap){ o->s = malloc(cap+1); o->n = 0; o->cap = cap; o->s[0] = 0; }
static void ostr_push(OStr* o, char c){ if (o->n + 1 >= o->cap){ o->cap *= 2; o->s = realloc(o->s, o->cap+1); } o->s[o->n++] = c; o->s[o->n] = 0; }
static void ostr_rev(OStr* o){ for (int i = 0, j = o->n-1; i < j; i++, j--){ char t = o->s[i]; o->s[i] = o->s[j]; o->s[j] = t; } }
static void ostr_cat(OStr* d, const OStr* a){ for (int i = 0; i < a->n; i++) ostr_push(d, a->s[i]); }

typedef struct {                                /* what AlignmentTrace accumulates (AlignmentTraceback.h:27-34) */
  int str_set, stutter_size; OStr str_seq;
  OStr flank[3];
  int flank_ins, flank_del;
  int n_indel, indel_pos[512], indel_size[512];
  int n_snp, snp_pos[512]; char snp_base[512];
} OTrace;

static int tri_idx(int rev, double v1, double v2, double v3){       /* HapAligner.cpp:346-358 */
  if (!rev){ if (v1 > v2+TRACE_LL_TOL) return (v1 > v3+TRACE_LL_TOL ? 0 : 2); return (v2 > v3+TRACE_LL_TOL ? 1 : 2); }
  if (v3 > v2+TRACE_LL_TOL) return (v3 > v1+TRACE_LL_TOL ? 2 : 0);
  return (v2 > v1+TRACE_LL_TOL ? 1 : 0);
}
static int pair_idx(int rev, double v1, double v2){                 /* HapAligner.cpp:360-361 */
  if (!rev) return (v1 > v2+TRACE_LL_TOL ? 0 : 1);
  return (v2 > v1+TRACE_LL_TOL ? 1 : 0);
}

/* block start coordinate as the (possibly reversed) haplotype reports it: HapBlock::reverse() builds HapBlock(end_-1, start_-1, ..),
 * RepeatBlock::reverse() keeps (start_, end_) (HapBlock.h:123-133, RepeatBlock.h:48-58) */
static int32_t side_block_start(const hipstr_batch_t* b, int rev, int bi){
  int fb = rev ? 2-bi : bi;
  if (!rev || fb == 1) return b->blk_start[fb];
  return b->blk_end[fb]-1;
}

/* HapAligner::retrace (HapAligner.cpp:363-571) */
static void retrace_side(const hipstr_batch_t* b, const OSide* sd, int rev, const OAln* a, int block_index, int base_index, long matrix_index,
                         OTrace* tr, OStr* aln){
  const int MATCH = 0, DEL = 1, INS = 2, NONE = -1;
  int n = a->n, seq_index = n-1, matrix_type = MATCH;
  const double* M = a->M; const double* I = a->I; const double* D = a->D;
  while (block_index >= 0){
    const OBlock* ob = &sd->blk[block_index]; int o = sd->counts[block_index];
    const char* bs = ob->seq[o]; int blen = ob->len[o];
    if (block_index == 1){
      int size = a->art_size[seq_index], apos = a->art_pos[seq_index];
      OStr ss; ostr_init(&ss, blen + 64);
      int i = 0;
      for (; i < (seq_index+1 < apos ? seq_index+1 : apos); i++){ ostr_push(aln, 'M'); ostr_push(&ss, a->rd[seq_index-i]); }
      if (size < 0) for (int k = 0; k < -size; k++) ostr_push(aln, 'D');
      else for (; i < (seq_index+1 < apos+size ? seq_index+1 : apos+size); i++){ ostr_push(aln, 'I'); ostr_push(&ss, a->rd[seq_index-i]); }
      for (; i < (blen+size < seq_index+1 ? blen+size : seq_index+1); i++){ ostr_push(aln, 'M'); ostr_push(&ss, a->rd[seq_index-i]); }
      if (!rev) ostr_rev(&ss);
      tr->str_set = 1; tr->stutter_size = size; tr->str_seq.n = 0; ostr_cat(&tr->str_seq, &ss);
      free(ss.s);
      if (blen + size >= seq_index+1) return;          /* sequence doesn't span the stutter block */
      matrix_index -= (blen + size + (long)n*blen);
      matrix_type = MATCH;
      seq_index -= (blen + size);
    } else {
      int prev_type = NONE;
      int32_t pos = side_block_start(b, rev, block_index) + (rev ? -base_index : base_index);
      const int32_t inc = rev ? 1 : -1;
      int indel_seq_index = -1; int32_t indel_position = -1;
      OStr fs; ostr_init(&fs, blen + 64);
      int out_block = rev ? 2-block_index : block_index;
      while (base_index >= 0 && seq_index >= 0){
        int h1 = hom_len(sd, block_index, base_index), h2 = hom_len(sd, block_index, base_index-1 > 0 ? base_index-1 : 0);
        int h = h1 > h2 ? h1 : h2; if (h > HIPSTR_MAX_HOMOP_LEN) h = HIPSTR_MAX_HOMOP_LEN;
        if (matrix_type != prev_type){
          if (prev_type == DEL){
            if (rev){ tr->indel_pos[tr->n_indel] = indel_position; tr->indel_size[tr->n_indel++] = indel_position - pos; }
            else    { tr->indel_pos[tr->n_indel] = pos+1;          tr->indel_size[tr->n_indel++] = pos - indel_position; }
          } else if (prev_type == INS){
            tr->indel_pos[tr->n_indel] = indel_position + (rev ? 0 : 1); tr->indel_size[tr->n_indel++] = indel_seq_index - seq_index;
          }
          if (matrix_type == DEL || matrix_type == INS){ indel_seq_index = seq_index; indel_position = pos; }
          prev_type = matrix_type;
        }
        if (matrix_type == MATCH){
          if (bs[base_index] != a->rd[seq_index] && a->blc[seq_index] > MIN_SNP_LOG_PROB_CORRECT){
            tr->snp_pos[tr->n_snp] = pos; tr->snp_base[tr->n_snp++] = a->rd[seq_index];
          }
          ostr_push(&fs, a->rd[seq_index]); ostr_push(aln, 'M'); seq_index--; base_index--; pos += inc;
        } else if (matrix_type == DEL){
          tr->flank_del++; ostr_push(aln, 'D'); base_index--; pos += inc;
        } else {
          tr->flank_ins++; ostr_push(&fs, a->rd[seq_index]); ostr_push(aln, 'I'); seq_index--;
        }
        if (seq_index == -1 || (base_index == -1 && block_index == 0)){
          while (seq_index != -1){ ostr_push(aln, 'S'); seq_index--; }
          if (!rev) ostr_rev(&fs);
          ostr_cat(&tr->flank[out_block], &fs);
          free(fs.s);
          return;
        }
        int best;
        if (matrix_type == MATCH){
          best = tri_idx(rev, I[matrix_index-1] + g_m2i[h], D[matrix_index-n-1] + g_m2d[h], M[matrix_index-n-1] + g_m2m[h]);
          if (best == 0){ matrix_type = INS; matrix_index -= 1; }
          else if (best == 1){ matrix_type = DEL; matrix_index -= (n+1); }
          else { matrix_type = MATCH; matrix_index -= (n+1); }
        } else if (matrix_type == DEL){
          best = pair_idx(rev, D[matrix_index-n] + D2D, M[matrix_index-n] + D2M);
          matrix_type = (best == 0) ? DEL : MATCH; matrix_index -= n;
        } else {
          best = pair_idx(rev, I[matrix_index-1] + I2I, M[matrix_index-n-1] + I2M);
          if (best == 0){ matrix_type = INS; matrix_index -= 1; }
          else { matrix_type = MATCH; matrix_index -= (n+1); }
        }
      }
      if (!rev) ostr_rev(&fs);
      ostr_cat(&tr->flank[out_block], &fs);
      free(fs.s);
    }
    block_index--;
    if (block_index >= 0) base_index = sd->blk[block_index].len[sd->counts[block_index]] - 1;
  }
}

/* AlignmentTraceback.cpp:7-52 */
static void stitch_dir(const char* hap_aln, int hlen, const char* read_aln, int rlen, int h_index, int r_index, int inc, OStr* out){
  while (r_index >= 0 && r_index < rlen){
    if (read_aln[r_index] == 'S'){ ostr_push(out, 'S'); r_index += inc; continue; }
    if (h_index < 0 || h_index >= hlen) return;
    if (hap_aln[h_index] == 'D'){
      if (read_aln[r_index] == 'I'){ ostr_push(out, 'M'); r_index += inc; h_index += inc; }
      else { ostr_push(out, 'D'); h_index += inc; }
    }
    else if (read_aln[r_index] == 'I'){ ostr_push(out, 'I'); r_index += inc; }
    else if (read_aln[r_index] == 'D'){
      if (hap_aln[h_index] == 'M') ostr_push(out, 'D');
      r_index += inc; h_index += inc;
    }
    else { ostr_push(out, hap_aln[h_index]); r_index += inc; h_index += inc; }
  }
}

static int put_pool(char* pool, int32_t* off, int idx, const char* s, int n, int cap){
  if (off[idx] + n > cap) return 1;
  memcpy(pool + off[idx], s, n); off[idx+1] = off[idx] + n; return 0;
}

int oracle_trace_seeded(const hipstr_batch_t* b, int32_t n_req, const int32_t* req_read, const int32_t* req_allele,
                        const int32_t* req_seed, const char* const* hap_to_ref, hipstr_trace_out_t* o);
int oracle_trace(const hipstr_batch_t* b, int32_t n_req, const int32_t* req_read, const int32_t* req_allele,
                 const char* const* hap_to_ref, hipstr_trace_out_t* o){
  return oracle_trace_seeded(b, n_req, req_read, req_allele, NULL, hap_to_ref, o);
}
/* trace_optimal_aln with the caller's seed_base (HapAligner.h:93); NULL / -2 entries: calc_seed_base */
int oracle_trace_seeded(const hipstr_batch_t* b, int32_t n_req, const int32_t* req_read, const int32_t* req_allele,
                        const int32_t* req_seed, const char* const* hap_to_ref, hipstr_trace_out_t* o){
  oracle_init();
  if (b->n_loci != 1) return 1;
  int period = b->period[0];
  OSide fw, rv;
  side_build(&fw, b, 0, 0, 0);
  side_build(&rv, b, 0, 0, 1);
  OIter it;
  for (int k = 0; k < 3; k++) it.n[k] = b->blk_nopts[k];
  int nso = fw.blk[1].nopts;
  double* pmf = malloc(sizeof(double)*nso*HIPSTR_NUM_ARTIFACTS);
  for (int oo = 0; oo < nso; oo++){
    int size = fw.blk[1].len[oo], t = 0;
    for (int art = -HIPSTR_MAX_STUTTER_REPS*period; art <= HIPSTR_MAX_STUTTER_REPS*period; art += period, t++)
      pmf[oo*HIPSTR_NUM_ARTIFACTS+t] = (size+art < 0) ? LARGE_NEGATIVE : oracle_stutter_pmf(b->stutter, period, size, size+art);
  }
  int Hmax = fw.max_size, rc = 0;
  o->hap_aln_off[0] = o->str_seq_off[0] = o->flank_seq_off[0] = o->indel_off[0] = o->snp_off[0] = 0;
  o->cigar_off[0] = o->aln_str_off[0] = 0;
  for (int q = 0; q < n_req && rc == 0; q++){
    int r = req_read[q];
    int sb = (req_seed && req_seed[q] != -2) ? req_seed[q] : seed_base(b, 0, r);
    if (sb < 0){ rc = 2; break; }
    iter_reset(&it);
    while (it.counter < req_allele[q]) if (!iter_next(&it)){ rc = 4; break; }
    if (rc) break;
    for (int k = 0; k < 3; k++){ fw.counts[k] = it.cnt[k]; rv.counts[k] = it.cnt[2-k]; }
    int len = b->base_off[r+1]-b->base_off[r];
    const char* bases = b->bases + b->base_off[r]; const char* quals = b->quals + b->base_off[r];
    double* lw = malloc(sizeof(double)*len); double* lc = malloc(sizeof(double)*len);
    for (int j = 0; j < len; j++){ lw[j] = g_q_error[qual_index(quals[j])]; lc[j] = g_q_correct[qual_index(quals[j])]; }
    int nL = sb, nR = len-sb-1;
    char* rrd = malloc(nR+1); double* rlw = malloc(sizeof(double)*nR); double* rlc = malloc(sizeof(double)*nR);
    for (int j = 0; j < nR; j++){ rrd[j] = bases[len-1-j]; rlw[j] = lw[len-1-j]; rlc[j] = lc[len-1-j]; }
    OAln L, R;
    L.n = nL; L.rd = bases; L.blc = lc; L.blw = lw; L.left_align = 1;
    R.n = nR; R.rd = rrd;  R.blc = rlc; R.blw = rlw; R.left_align = 0;
    OAln* sides[2] = { &L, &R };
    for (int s2 = 0; s2 < 2; s2++){
      OAln* a = sides[s2];
      a->M = malloc(sizeof(double)*(size_t)a->n*Hmax); a->I = malloc(sizeof(double)*(size_t)a->n*Hmax); a->D = malloc(sizeof(double)*(size_t)a->n*Hmax);
      a->art_size = malloc(sizeof(int)*a->n); a->art_pos = malloc(sizeof(int)*a->n);
    }
    double* scratch = malloc(sizeof(double)*(Hmax+4));
    align_side(&fw, 0, -1, &L, pmf, period);                 /* go_to() clears last_changed: nothing is reused (Haplotype.cpp:206) */
    align_side(&rv, 0, -1, &R, pmf, period);
    int max_index = 0;
    double LL = combine(&fw, &L, &R, bases[sb], lw[sb], lc[sb], scratch, &max_index);
    int H = 0, blen3[3];
    for (int k = 0; k < 3; k++){ blen3[k] = fw.blk[k].len[fw.counts[k]]; H += blen3[k]; }

    OTrace tr; memset(&tr, 0, sizeof tr);
    ostr_init(&tr.str_seq, 64); for (int k = 0; k < 3; k++) ostr_init(&tr.flank[k], 64);
    OStr left, right, full; ostr_init(&left, len + H + 8); ostr_init(&right, len + H + 8); ostr_init(&full, 2*(len + H) + 8);
    /* left of the seed (HapAligner.cpp:642-659) */
    int sblock = 0, scoord = max_index;
    for (int k = 0; k < 3; k++){ if (scoord < blen3[k]){ sblock = k; break; } scoord -= blen3[k]; }
    if (max_index == 0) for (int i = 0; i < nL; i++) ostr_push(&left, 'S');
    else {
      long mi = (long)nL*max_index - 1;
      if (scoord == 0) retrace_side(b, &fw, 0, &L, sblock-1, blen3[sblock-1]-1, mi, &tr, &left);
      else             retrace_side(b, &fw, 0, &L, sblock, scoord-1, mi, &tr, &left);
    }
    ostr_rev(&left);
    if (sblock != 1) ostr_push(&tr.flank[sblock], bases[sb]);       /* HapAligner.cpp:661-665 */
    /* right of the seed (HapAligner.cpp:667-684) */
    int rmax = H-1-max_index, rblock = 0, rcoord = rmax, rlen3[3] = { blen3[2], blen3[1], blen3[0] };
    for (int k = 0; k < 3; k++){ if (rcoord < rlen3[k]){ rblock = k; break; } rcoord -= rlen3[k]; }
    if (rmax == 0) for (int i = 0; i < nR; i++) ostr_push(&right, 'S');
    else {
      long mi = (long)nR*rmax - 1;
      if (rcoord == 0) retrace_side(b, &rv, 1, &R, rblock-1, rlen3[rblock-1]-1, mi, &tr, &right);
      else             retrace_side(b, &rv, 1, &R, rblock, rcoord-1, mi, &tr, &right);
    }
    ostr_cat(&full, &left); ostr_push(&full, 'M'); ostr_cat(&full, &right);

    o->ll[q] = LL; o->max_index[q] = max_index;
    rc |= put_pool(o->hap_aln, o->hap_aln_off, q, full.s, full.n, o->cap_chars);
    o->stutter_size[q] = tr.str_set ? tr.stutter_size : HIPSTR_NO_STR_DATA;
    rc |= put_pool(o->str_seq, o->str_seq_off, q, tr.str_seq.s, tr.str_set ? tr.str_seq.n : 0, o->cap_chars);
    rc |= put_pool(o->flank_seq, o->flank_seq_off, 2*q, tr.flank[0].s, tr.flank[0].n, o->cap_chars);
    rc |= put_pool(o->flank_seq, o->flank_seq_off, 2*q+1, tr.flank[2].s, tr.flank[2].n, o->cap_chars);
    o->flank_ins[q] = tr.flank_ins; o->flank_del[q] = tr.flank_del;
    int io = o->indel_off[q];
    for (int i = 0; i < tr.n_indel && io < o->cap_chars; i++, io++){ o->indel_pos[io] = tr.indel_pos[i]; o->indel_size[io] = tr.indel_size[i]; }
    o->indel_off[q+1] = io;
    int so = o->snp_off[q];
    for (int i = 0; i < tr.n_snp && so < o->cap_chars; i++, so++){ o->snp_pos[so] = tr.snp_pos[i]; o->snp_base[so] = tr.snp_base[i]; }
    o->snp_off[q+1] = so;

    /* stitch_alignment_trace (AlignmentTraceback.cpp:55-144) */
    o->cigar_off[q+1] = o->cigar_off[q]; o->aln_str_off[q+1] = o->aln_str_off[q]; o->aln_start[q] = o->aln_stop[q] = 0;
    if (hap_to_ref != NULL && rc == 0){
      const char* h2r = hap_to_ref[req_allele[q]]; int hlen = (int)strlen(h2r);
      int hap_index = max_index, hai = 0; int32_t seed_pos = b->blk_start[0];
      while (hap_index > 0 && hai < hlen){
        if (h2r[hai] == 'M' || h2r[hai] == 'I') hap_index--;
        if (h2r[hai] == 'M' || h2r[hai] == 'D') seed_pos++;
        hai++;
      }
      while (hai < hlen && h2r[hai] == 'D') hai++;
      int sbase = sb, rai = 0;
      while (sbase > 0 && rai < full.n){
        if (full.s[rai] == 'M' || full.s[rai] == 'I' || full.s[rai] == 'S') sbase--;
        rai++;
      }
      while (rai < full.n && full.s[rai] == 'D') rai++;
      OStr la, ra, fa; ostr_init(&la, full.n + hlen + 8); ostr_init(&ra, full.n + hlen + 8); ostr_init(&fa, 2*(full.n + hlen) + 8);
      stitch_dir(h2r, hlen, full.s, full.n, hai-1, rai-1, -1, &la);
      ostr_rev(&la);
      stitch_dir(h2r, hlen, full.s, full.n, hai+1, rai+1, 1, &ra);
      ostr_cat(&fa, &la); ostr_push(&fa, 'M'); ostr_cat(&fa, &ra);
      for (int i = 0; i < fa.n; i++){ if (fa.s[i] == 'I') fa.s[i] = 'S'; else break; }
      int32_t start = seed_pos, stop = seed_pos;
      for (int i = 0; i < la.n; i++) if (la.s[i] == 'D' || la.s[i] == 'M') start--;
      for (int i = 0; i < ra.n; i++) if (ra.s[i] == 'D' || ra.s[i] == 'M') stop++;
      o->aln_start[q] = start; o->aln_stop[q] = stop;
      int co = o->cigar_off[q]; char cc = fa.s[0]; int num = 1;
      for (int i = 1; i <= fa.n; i++){
        if (i == fa.n || fa.s[i] != cc){
          if (co < o->cap_chars){ o->cigar_op[co] = cc; o->cigar_len[co] = num; co++; }
          if (i < fa.n){ cc = fa.s[i]; num = 1; }
        } else num++;
      }
      o->cigar_off[q+1] = co;
      OStr as; ostr_init(&as, fa.n + 8);
      int ri = 0;
      for (int i = 0; i < fa.n; i++){
        if (fa.s[i] == 'S') ri++;
        else if (fa.s[i] == 'M' || fa.s[i] == 'I') ostr_push(&as, bases[ri++]);
        else ostr_push(&as, '-');
      }
      rc |= put_pool(o->aln_str, o->aln_str_off, q, as.s, as.n, o->cap_chars);
      free(la.s); free(ra.s); free(fa.s); free(as.s);
    }
    free(left.s); free(right.s); free(full.s); free(tr.str_seq.s); for (int k = 0; k < 3; k++) free(tr.flank[k].s);
    free(scratch);
    for (int s2 = 0; s2 < 2; s2++){ OAln* a = sides[s2]; free(a->M); free(a->I); free(a->D); free(a->art_size); free(a->art_pos); }
    free(rrd); free(rlw); free(rlc); free(lw); free(lc);
  }
  free(pmf); side_free(&fw); side_free(&rv);
  return rc;
}
