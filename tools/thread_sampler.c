/* thread_sampler.c — where is another thread of THIS process right now?  ptrace is not permitted on the GPU boxes, so: a SIGUSR2 handler that
 * takes backtrace() on whichever thread the signal is sent to (tgkill), and ts_sample(tid) that sends it and returns the symbolised frames.
 * gcc -O1 -g -shared -fPIC -o /tmp/libts.so tools/thread_sampler.c -ldl     (tools/r06_rt_stack.py) */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>
static void* g_frames[48]; static volatile int g_n = -1;
static void on_sig(int s){ (void)s; g_n = backtrace(g_frames, 48); }
int ts_init(void){ struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = on_sig; sa.sa_flags = SA_RESTART; return sigaction(SIGUSR2, &sa, NULL); }
/* frames of thread `tid` as "lib+offset symbol" lines into out; returns the number of frames, -1 if the signal was not taken in 200 ms */
int ts_sample(int tid, char* out, int cap){
  g_n = -1;
  if (syscall(SYS_tgkill, getpid(), tid, SIGUSR2) != 0) return -2;
  for (int i = 0; i < 2000 && g_n < 0; i++) usleep(100);
  if (g_n < 0) return -1;
  int n = g_n, used = 0; out[0] = 0;
  for (int i = 0; i < n; i++){
    Dl_info di; const char* lib = "?"; const char* sym = "?"; unsigned long off = 0;
    if (dladdr(g_frames[i], &di)){ if (di.dli_fname) lib = di.dli_fname; if (di.dli_sname) sym = di.dli_sname; off = (unsigned long)((char*)g_frames[i] - (char*)di.dli_fbase); }
    const char* base = strrchr(lib, '/'); base = base ? base + 1 : lib;
    used += snprintf(out + used, cap - used > 0 ? cap - used : 0, "%s+0x%lx %s\n", base, off, sym);
    if (used >= cap) break;
  }
  return n;
}
