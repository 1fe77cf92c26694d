// debugging aid: print the native backtrace on SIGSEGV (load with ctypes, call install_segv_bt())
#include <execinfo.h>
#include <signal.h>
#include <unistd.h>
static void on_segv(int sig)
{
    void *frames[64];
    int n = backtrace(frames, 64);
    const char msg[] = "\n== native backtrace ==\n";
    (void)!write(2, msg, sizeof(msg) - 1);
    backtrace_symbols_fd(frames, n, 2);
    _exit(139);
}
void install_segv_bt(void) { signal(SIGSEGV, on_segv); }
