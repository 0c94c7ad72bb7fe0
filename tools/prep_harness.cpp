// Host-only timing harness of prepare_batch (no device): g++ -O2 [-pg] tools/prep_harness.cpp hipstr_amd/csrc/prep.cpp hipstr_amd/synth/synth.cpp -lpthread
//   prep_harness <loci> <reads> <alleles> <read_len> <flank> <str_bp> <threads> <reps>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include "../hipstr_amd/csrc/prep.h"
extern "C" void* synth_create_at(int32_t first_locus, int32_t n_loci, int32_t reads_per_locus, int32_t n_str_alleles, int32_t read_len, int32_t flank_len,
                                 int32_t str_bp, int32_t n_flank_opts, uint64_t seed, double mask_rate);
extern "C" const hipstr_batch_t* synth_batch(void* h);
int main(int argc, char** argv){
  if (argc < 9){ fprintf(stderr, "usage\n"); return 2; }
  int a[8]; for (int i = 0; i < 8; i++) a[i] = atoi(argv[i+1]);
  void* h = synth_create_at(0, a[0], a[1], a[2], a[3], a[4], a[5], 1, 20260928, 0.0);
  const hipstr_batch_t* b = synth_batch(h);
  hipstr::set_host_threads(a[6]);
  double best = 1e9;
  for (int r = 0; r < a[7]; r++){
    hipstr::Prepared P; std::string err;
    hipstr::adopt_recycled(P);
    const auto t0 = std::chrono::steady_clock::now();
    if (hipstr::prepare_batch(b, P, err)){ fprintf(stderr, "%s\n", err.c_str()); return 1; }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (s < best) best = s;
    hipstr::recycle_prepared(P);
  }
  hipstr::prep_profile_print();
  printf("prepare_batch best %.2f ms = %.2f us per locus\n", 1e3*best, 1e6*best/a[0]);
  return 0;
}
