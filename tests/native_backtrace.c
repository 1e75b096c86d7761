/* Test infrastructure: a SIGSEGV / SIGBUS / SIGABRT handler that prints the NATIVE frames (library + offset, resolvable with
 * addr2line against the same build) before the process dies.  Loaded by tests/mp_worker.py when PANGULU_TEST_NATIVE_BACKTRACE=1
 * (python's faulthandler, the default, shows only the python frames of a crash inside the solver library). */
#include <execinfo.h>
#include <signal.h>
#include <string.h>
#include <unistd.h>

static void handler(int sig)
{
    void *frames[64];
    const char msg[] = "[native backtrace] signal caught, frames (library(+offset)):\n";
    if (write(2, msg, sizeof msg - 1) < 0) { }
    int n = backtrace(frames, 64);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

void pg_test_install_native_backtrace(void)
{
    void *warm[4];
    backtrace(warm, 4); /* (loads libgcc now, not inside the handler) */
    signal(SIGSEGV, handler);
    signal(SIGBUS, handler);
    signal(SIGABRT, handler);
}
