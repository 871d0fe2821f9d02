/* LD_PRELOAD shim: native backtrace of the thread that raises SIGABRT / SIGSEGV (diagnosing aborts inside HIP / HSA / glibc that leave
   no message).  gcc -shared -fPIC -o abort_bt.so abort_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>
#include <sys/prctl.h>

static void handler(int sig)
{
    void *frames[64];
    const char *msg = sig == SIGABRT ? "\n== SIGABRT, native backtrace of the raising thread:\n" : "\n== SIGSEGV, native backtrace:\n";
    write(2, msg, strlen(msg));
    char name[32] = "thread: ";
    prctl(PR_GET_NAME, name + 8, 0, 0, 0);   /* which thread raised it: the Python main thread, an HSA / HIP / RCCL worker ... */
    write(2, name, strlen(name));
    write(2, "\n", 1);
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void)
{
    signal(SIGABRT, handler);
    signal(SIGSEGV, handler);
}
