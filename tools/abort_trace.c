// Debugging aid: LD_PRELOAD=tools/abort_trace.so prints the C stack of whoever calls abort() (or raises SIGABRT) to stderr.
// gcc -shared -fPIC -O1 -o tools/abort_trace.so tools/abort_trace.c   (run pytest with -s: its fd capture hides stderr)
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
static void dump(const char *why) {
    void *bt[64];
    int n = backtrace(bt, 64);
    dprintf(2, "\n=== abort_trace: %s ===\n", why);
    backtrace_symbols_fd(bt, n, 2);
    dprintf(2, "=== end ===\n");
}
void abort(void) {
    dump("abort() called");
    signal(SIGABRT, SIG_DFL);
    raise(SIGABRT);
    _exit(134);
}
static void on_sig(int s) {
    dump(s == SIGABRT ? "SIGABRT" : "SIGSEGV");
    signal(s, SIG_DFL);
    raise(s);
}
__attribute__((constructor)) static void init(void) {
    void *bt[4];
    backtrace(bt, 4);  /* load libgcc now */
    signal(SIGABRT, on_sig);
}
