// Host-only check of the K-level piecewise closed form (layout.h HS_SHAPE_PWK, prep.cpp piecewise_k) against the entry-by-entry replay of
// the same visiting list (what hs_str_group_kernel_rp's visit_eval_grp does on the device and StutterAlignerClass.cpp:59-150 in the
// reference): random interrupted repeat blocks through the library's own preparation code, random emission tables, every bound of every
// list; the two must agree BIT FOR BIT.  The evaluator below restates pwk_eval_grp (hmm_kernels.hip) operation for operation on the host.
//   g++ -O1 -std=c++17 -ffp-contract=off tests/cpp/pwk_form_test.cpp hipstr_amd/csrc/prep.cpp -lpthread -o pwk_form_test
//   pwk_form_test <blocks> <seed>     prints "lists <n> pwk <n> evaluations <n> mismatches <n>"
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>
#include "../../hipstr_amd/csrc/prep.h"

namespace {
float as_float(uint32_t u){ float f; memcpy(&f, &u, 4); return f; }
uint32_t as_uint(float f){ uint32_t u; memcpy(&u, &f, 4); return u; }
float fasterexp(float p){          // fastonebigheader.h:206-218
  const float y = 1.442695040f * p;
  const float c = (y < -126.0f) ? -126.0f : y;
  const float z = 8388608.0f * (c + 126.94269504f);
  return as_float((uint32_t)z);
}
float fasterlog(float x){          // fastonebigheader.h:348-358
  float y = (float)as_uint(x);
  y *= 8.2629582881927490e-8f;
  return y - 87.989971088f;
}
const double LOG_THRESH = log(0.001);
struct Lse {
  double mx, tot;
  void start(int pass, double first){ if (pass == 0) mx = first; else tot = 0.0; }
  void push(int pass, double v){ if (pass == 0) mx = fmax(mx, v); else { const double d = v - mx; if (d > LOG_THRESH) tot += (double)fasterexp((float)d); } }
  double finish() const { return mx + (double)fasterlog((float)tot); }
};
inline int plane_of(int ch){ return (ch >> 1) & 3; }

// E[plane][column]; xx: the lane's column
double replay(const hs_visit_t* list, int llen, const std::vector<double>& ilog, const std::vector<double> E[4], int xx, double lp0, int lim, int limmax,
              int nsub, int stride, int tail){
  Lse acc; double lp = lp0; int nistop = 0;
  for (int pass = 0; pass < 2; pass++){
    lp = lp0; acc.start(pass, lp0); acc.push(pass, lp0);
    nistop = 0; bool stopped = false;
    for (int v = 0; v < llen; v++){
      const uint64_t meta = list[v].meta;
      const int ni = (int)(meta & 0xffff);
      if (ni >= limmax){ if (!stopped) nistop = ni; break; }
      const bool act = ni < lim;
      if (!act && !stopped){ nistop = ni; stopped = true; }
      const int U = (int)((meta >> 16) & 0xffff);
      if ((meta >> 48) & 1){ if (act) acc.push(pass, lp); }
      else if (U == 0){
        const int pla = plane_of((int)(meta >> 32) & 0xff), plb = plane_of((int)(meta >> 40) & 0xff);
        double t = lp;
        for (int m = 1; m <= nsub; m++){ const int c = std::max(xx - ni - m*stride, 0); t -= E[pla][c]; t += E[plb][c]; }
        lp = act ? t : lp;
        if (act) acc.push(pass, lp);
      } else { if (act) acc.push(pass, list[v].logU + lp); }
    }
    if (nistop < tail) acc.push(pass, ilog[std::max(tail - nistop, 0)] + lp);
  }
  return acc.finish();
}

double pwk_form(const double* slots, const std::vector<double>& ilog, const std::vector<double> E[4], int xx, double lp0, int lim, int nsub, int stride, int tail){
  auto lo = [&](int sl){ uint64_t u; memcpy(&u, slots + sl, 8); return (int)(uint32_t)u; };
  auto hi = [&](int sl){ uint64_t u; memcpy(&u, slots + sl, 8); return (int)(uint32_t)(u >> 32); };
  const int nseg = lo(0), term_ni = hi(0), pa = lo(1), pb = hi(1);
  const double NEG = -1.0e300;
  double Lv[HS_PWK_MAX + 1];
  Lv[0] = lp0;
  double mx = lp0;
  unsigned u = (unsigned)(term_ni - lim);
  for (int s = 0; s <= HS_PWK_MAX; s++){
    if (s < HS_PWK_MAX) Lv[s + 1] = Lv[s];
    if (s <= nseg){
      if (hi(2 + 3*s) > 0){
        const double v = slots[3 + 3*s] + Lv[s];
        mx = fmax(mx, (lo(2 + 3*s) < lim) ? v : NEG);
        u = std::min(u, (unsigned)(lo(2 + 3*s) - lim));
      }
      if (s < HS_PWK_MAX && s < nseg){
        const int b = lo(4 + 3*s), pla = plane_of(hi(4 + 3*s)), plb = plane_of(hi(4 + 3*s) >> 8);
        double t = Lv[s];
        for (int m = 1; m <= nsub; m++){ const int c = std::max(xx - b - m*stride, 0); t -= E[pla][c]; t += E[plb][c]; }
        Lv[s + 1] = (b < lim) ? t : Lv[s];
        mx = fmax(mx, Lv[s + 1]);
        u = std::min(u, (unsigned)(b - lim));
      }
    }
  }
  double Llast = Lv[0];
  for (int s = 1; s <= HS_PWK_MAX; s++) Llast = (s <= nseg) ? Lv[s] : Llast;
  const int np = std::min(std::max(lim - pa, 0), pb - pa);
  if (pb > pa) u = std::min(u, (lim < pb) ? (unsigned)std::max(pa - lim, 0) : 0xffffffffu);
  const int ns = (int)u + lim;
  const bool a_t = ns < tail;
  const double v_t = a_t ? ilog[std::max(tail - ns, 0)] + Llast : NEG;
  mx = fmax(mx, v_t);
  double tot = 0.0;
  auto fexp = [&](double dd){ const float z = (((float)dd) * 1.442695040f + 126.94269504f) * 8388608.0f; return as_float((uint32_t)z); };
  auto pair = [&](double a, bool on_a, double b, double wa){
    const double dd0 = a - mx, dd1 = b - mx;
    const float fe0 = (on_a && dd0 > LOG_THRESH) ? fexp(dd0) : 0.0f;
    const float fe1 = (dd1 > LOG_THRESH) ? fexp(dd1) : 0.0f;
    tot += wa * (double)fe0;
    tot += (double)fe1;
  };
  auto run_value = [&](int s) -> double {
    if (hi(2 + 3*s) <= 0) return NEG;
    const double v = slots[3 + 3*s] + Lv[s];
    return (lo(2 + 3*s) < lim) ? v : NEG;
  };
  pair(Lv[0], true, run_value(0), 1.0);
  for (int s = 1; s <= HS_PWK_MAX; s++) if (s <= nseg) pair(Lv[s], lo(4 + 3*(s - 1)) < lim, run_value(s), 1.0);
  pair(Llast, np > 0, v_t, (double)np);
  return mx + (double)fasterlog((float)tot);
}
}  // namespace

int main(int argc, char** argv){
  const int n_blocks = argc > 1 ? atoi(argv[1]) : 300;
  std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], 0, 10) : 12345);
  auto rnd = [&](int n){ return (int)(rng() % (uint64_t)n); };
  std::vector<double> ilog(10000); ilog[0] = -1000; for (int i = 1; i < 10000; i++) ilog[i] = log((double)i);
  const double stutter[6] = {0.05, 0.05, 0.9, 0.05, 0.05, 0.9};      // (the lists and their descriptors do not depend on the model)
  long n_lists = 0, n_pwk = 0, n_eval = 0, n_bad = 0, by_breaks[HS_PWK_MAX + 1] = {0};
  const int XC = 256;
  for (int blk_i = 0; blk_i < n_blocks; blk_i++){
    const int p = 1 + rnd(6), units = 3 + rnd(18);
    std::string unit; for (int i = 0; i < p; i++) unit.push_back("ACGT"[rnd(4)]);
    std::string blk; for (int u = 0; u < units; u++) blk += unit;
    if (rnd(4) == 0) blk += unit.substr(0, rnd(p));                  // a partial unit at the end
    const int nsubst = 1 + rnd(4);
    for (int k = 0; k < nsubst; k++){ const int at = rnd((int)blk.size()); blk[at] = "ACGT"[rnd(4)]; }
    if (rnd(5) == 0){ const int at = rnd((int)blk.size()); blk.insert(at, 1, "ACGT"[rnd(4)]); }      // an inserted base: the phase shifts
    const int B = (int)blk.size();
    hipstr::Prepared P;
    hipstr::append_stropt(blk, p, stutter, P);
    const hs_stropt_t& so = P.stropts.back();
    bool any = false;
    for (int k = 0; k <= HS_MAXREP; k++) any |= (so.shape[k] == HS_SHAPE_PWK);
    n_lists += HS_MAXREP + 1;
    if (!any) continue;
    const double* kslots = P.f64pool.data() + so.f64_off + 20 + (HS_MAXREP + 1)*HS_PW_SLOTS;
    // a read's emission table: log P(correct) / log P(error) per column by plane
    std::vector<double> E[4];
    for (int pl = 0; pl < 4; pl++) E[pl].resize(XC);
    for (int c = 0; c < XC; c++){
      const int q = 2 + rnd(40); const double perr = pow(10.0, -q/10.0);
      const double qc = log(1.0 - perr), qe = log(perr/3.0);
      const int base = rnd(4);
      for (int pl = 0; pl < 4; pl++) E[pl][c] = (pl == base) ? qc : qe;
    }
    for (int k = 0; k <= HS_MAXREP; k++){
      if (so.shape[k] != HS_SHAPE_PWK) continue;
      n_pwk++;
      const double* sl = kslots + k*HS_PWK_SLOTS;
      { uint64_t u0; memcpy(&u0, sl, 8); by_breaks[std::min((int)(uint32_t)u0, HS_PWK_MAX)]++; }
      const bool ins = (k == HS_MAXREP);
      const int tail = ins ? B : B - (k+1)*p;
      const hs_visit_t* list = P.visits.data() + (ins ? so.ins_off : so.del_off[k]);
      const int llen = ins ? so.ins_len : so.del_len[k];
      for (int nsub = 1; nsub <= (ins ? HS_MAXREP : 1); nsub++){
        const int stride = ins ? p : 0;
        for (int lim = 0; lim <= tail; lim++){
          for (int rep = 0; rep < 3; rep++){
            const int xx = rnd(XC);                                   // (small columns: the clamp at the read start is part of the form)
            const double lp0 = -(double)rnd(4000)/7.0 - 0.001*rnd(1000);
            const double a = replay(list, llen, ilog, E, xx, lp0, lim, tail, nsub, stride, tail);
            const double b = pwk_form(sl, ilog, E, xx, lp0, lim, nsub, stride, tail);
            n_eval++;
            if (memcmp(&a, &b, 8) != 0){
              if (n_bad < 10) fprintf(stderr, "MISMATCH block %s period %d list %d lim %d nsub %d xx %d lp0 %.17g: replay %.17g form %.17g\n", blk.c_str(), p, k, lim, nsub, xx, lp0, a, b);
              n_bad++;
            }
          }
        }
      }
    }
  }
  printf("lists %ld pwk %ld evaluations %ld mismatches %ld\n", n_lists, n_pwk, n_eval, n_bad);
  printf("breaks"); for (int k = 0; k <= HS_PWK_MAX; k++) printf(" %ld", by_breaks[k]); printf("\n");
  return n_bad ? 1 : 0;
}
