/* flow_launcher.c — loads one of the genotype-flow libraries (oracle/_ref/libflow_ref.so, libflow_mi355x.so) with LAZY binding
 * and calls its flow_main.  The libraries contain SeqStutterGenotyper, whose VCF/BAM/visualisation code refers to htslib
 * symbols that are not built here; those functions are never called, and lazy binding lets the symbols stay unresolved
 * (the same arrangement as oracle/_ref/libhipstr_ref.so).  TEST INFRASTRUCTURE. */
#include <dlfcn.h>
#include <stdio.h>

int main(int argc, char** argv){
  if (argc < 2){ fprintf(stderr, "usage: %s <libflow_*.so> [flow arguments]\n", argv[0]); return 2; }
  void* h = dlopen(argv[1], RTLD_LAZY | RTLD_GLOBAL);
  if (!h){ fprintf(stderr, "%s\n", dlerror()); return 2; }
  int (*fn)(int, char**) = (int (*)(int, char**))dlsym(h, "flow_main");
  if (!fn){ fprintf(stderr, "%s\n", dlerror()); return 2; }
  return fn(argc - 1, argv + 1);
}
