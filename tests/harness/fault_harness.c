/* Test infrastructure (loaded by tests/conftest.py, never by the product): what a process that dies inside the GPU suite leaves
 * behind.  A HIP / HSA runtime thread that hits a queue error or a memory fault prints one line to stderr -- which pytest has
 * redirected into a temporary file that dies with the process -- and calls abort().  The handler below runs ON the aborting
 * thread: it writes that thread's native stack (rocr's fault handler?  a queue-error callback?  an assert?) and whatever the
 * process had written to the captured stderr since the current test began into <dir>/fault_<pid>.txt, then lets the signal
 * take its default course (core dump; rocgdb is run on it by the conftest of the parent when there is one). */
#define _GNU_SOURCE
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>

static char g_path[512];
static struct sigaction g_prev[64];

static void put(int fd, const char *s) { (void)!write(fd, s, strlen(s)); }
static void put_num(int fd, long v) {
    char b[32];
    int n = 0;
    if (v < 0) { put(fd, "-"); v = -v; }
    do { b[n++] = (char)('0' + v % 10); v /= 10; } while (v && n < 31);
    while (n) (void)!write(fd, &b[--n], 1);
}

static void on_fault(int sig, siginfo_t *si, void *uc) {
    (void)uc;
    const int fd = open(g_path, O_WRONLY | O_CREAT | O_APPEND, 0644);
    if (fd >= 0) {
        put(fd, "==== signal "); put_num(fd, sig);
        put(fd, " on thread "); put_num(fd, (long)syscall(SYS_gettid));
        put(fd, " of process "); put_num(fd, (long)getpid());
        if (sig == SIGSEGV || sig == SIGBUS) { put(fd, ", address "); put_num(fd, (long)si->si_addr); }
        put(fd, "\n---- native stack of the faulting thread\n");
        void *bt[96];
        const int n = backtrace(bt, 96);
        backtrace_symbols_fd(bt, n, fd);
        /* stderr as pytest captured it (a regular file while capture is on): the runtime's own last words */
        struct stat st;
        if (fstat(2, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0) {
            put(fd, "---- captured stderr of the current test (tail)\n");
            char buf[4096];
            off_t off = st.st_size > 65536 ? st.st_size - 65536 : 0;
            for (;;) {
                const ssize_t r = pread(2, buf, sizeof buf, off);
                if (r <= 0) break;
                (void)!write(fd, buf, (size_t)r);
                off += r;
            }
            put(fd, "\n");
        }
        put(fd, "==== end\n");
        close(fd);
    }
    /* default course (and the handler that was there before, e.g. Python's faulthandler, has already run or will not) */
    sigaction(sig, &g_prev[sig], NULL);
    raise(sig);
}

int cm_fault_harness_install(const char *dir) {
    snprintf(g_path, sizeof g_path, "%s/fault_%ld.txt", dir, (long)getpid());
    void *warm[4];
    (void)backtrace(warm, 4);          /* loads libgcc now: not async-signal-safe to do inside the handler */
    const int sigs[] = {SIGABRT, SIGSEGV, SIGBUS, SIGILL, SIGFPE};
    for (unsigned i = 0; i < sizeof sigs / sizeof sigs[0]; ++i) {
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_sigaction = on_fault;
        sa.sa_flags = SA_SIGINFO | SA_NODEFER | SA_ONSTACK;
        sigemptyset(&sa.sa_mask);
        if (sigaction(sigs[i], &sa, &g_prev[sigs[i]]) != 0) return -1;
    }
    return 0;
}
