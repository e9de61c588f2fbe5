/* LD_PRELOAD helper: print a native backtrace on SIGSEGV (debugging aid for the multi-threaded batched builder).  gcc -shared -fPIC -o build/libsegv_bt.so tools/probes/segv_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void h(int sig, siginfo_t* si, void* u) { void* bt[64]; int n = backtrace(bt, 64); const char* m = "\n== SIGSEGV native backtrace ==\n"; write(2, m, strlen(m)); backtrace_symbols_fd(bt, n, 2); _exit(139); }
__attribute__((constructor)) static void init(void) { struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = h; sa.sa_flags = SA_SIGINFO | SA_ONSTACK; sigaction(SIGSEGV, &sa, 0); }
