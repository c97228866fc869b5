/* LD_PRELOAD helper for tools/probes/order_probe.py: on SIGABRT (glibc's "double free or corruption") print the C backtrace of the
 * aborting thread and the ROCm / torch libraries mapped into the process, then let the abort proceed.  Diagnostic only.
 *   gcc -O1 -g -shared -fPIC -o abort_bt.so abort_bt.c */
#define _GNU_SOURCE
#include <execinfo.h>
#include <signal.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

static void on_abort(int sig) {
    void* frames[64];
    int n = backtrace(frames, 64);
    const char* head = "\n=== abort_bt: backtrace of the aborting thread ===\n";
    (void)!write(2, head, strlen(head));
    backtrace_symbols_fd(frames, n, 2);
    const char* mid = "=== abort_bt: mapped ROCm / torch libraries (first mapping of each) ===\n";
    (void)!write(2, mid, strlen(mid));
    FILE* f = fopen("/proc/self/maps", "r");
    if (f) {
        char line[1024], last[512] = "";
        while (fgets(line, sizeof line, f)) {
            char* p = strchr(line, '/');
            if (!p) continue;
            if (!(strstr(p, "rccl") || strstr(p, "amdhip") || strstr(p, "hsa-runtime") || strstr(p, "rocm_smi") || strstr(p, "roctracer") ||
                  strstr(p, "rocprofiler") || strstr(p, "libtorch") || strstr(p, "lbm_hip") || strstr(p, "libc10") || strstr(p, "comgr")))
                continue;
            if (strncmp(p, last, sizeof last - 1) == 0) continue;
            strncpy(last, p, sizeof last - 1);
            (void)!write(2, line, strlen(line));
        }
        fclose(f);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void install(void) {
    void* warm[4];
    backtrace(warm, 4);   /* (loads libgcc now, not inside the handler) */
    signal(SIGABRT, on_abort);
}
