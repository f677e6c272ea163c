// Diagnostics only (tools/repro_plan_churn.py, bench.py PAFC_BENCH_WATCHDOG): where is the MAIN thread's native stack when a
// run stops making progress?  faulthandler shows the Python frames; this shows the C frames under them (is the host inside
// hipblasLtMatmulAlgoGetHeuristic, a module load, hipStreamSynchronize, hipEventRecord ...), without a debugger attached to a
// process that owns GPU queues.
//
//   gcc -O1 -g -shared -fPIC -o tools/micro/_build/libnative_watchdog.so tools/micro/native_watchdog.c -lpthread
//
// nw_arm(seconds): call from the thread to be watched.  A helper thread sleeps in steps; nw_pet() restarts the count.  On
// expiry it sends SIGUSR2 to the watched thread, whose handler writes backtrace() of that thread to stderr (fd 2).
#define _GNU_SOURCE
#include <execinfo.h>
#include <pthread.h>
#include <signal.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

static pthread_t g_watched;
static volatile int g_limit = 0, g_left = 0, g_fired = 0;

static void on_usr2(int sig) {
    (void)sig;
    void *frames[96];
    static const char head[] = "\n==== native_watchdog: C stack of the watched thread ====\n";
    static const char tail[] = "==== end of C stack ====\n";
    (void)!write(2, head, sizeof(head) - 1);
    int n = backtrace(frames, 96);
    backtrace_symbols_fd(frames, n, 2);
    (void)!write(2, tail, sizeof(tail) - 1);
}

static void *waiter(void *arg) {
    (void)arg;
    for (;;) {
        struct timespec ts = {1, 0};
        nanosleep(&ts, NULL);
        if (g_limit <= 0) continue;
        if (--g_left <= 0 && !g_fired) {
            g_fired = 1;
            pthread_kill(g_watched, SIGUSR2);
        }
    }
    return NULL;
}

int nw_arm(int seconds) {
    static int started = 0;
    struct sigaction sa;
    memset(&sa, 0, sizeof(sa));
    sa.sa_handler = on_usr2;
    sigemptyset(&sa.sa_mask);
    if (sigaction(SIGUSR2, &sa, NULL) != 0) return -1;
    void *warm[4];
    (void)backtrace(warm, 4);            // loads libgcc's unwinder now, not inside the handler
    g_watched = pthread_self();
    g_limit = g_left = seconds;
    g_fired = 0;
    if (!started) {
        pthread_t t;
        if (pthread_create(&t, NULL, waiter, NULL) != 0) return -2;
        pthread_detach(t);
        started = 1;
    }
    return 0;
}

void nw_pet(void) { g_left = g_limit; g_fired = 0; }

void nw_disarm(void) { g_limit = 0; }
