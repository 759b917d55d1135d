/* exit_line.c -- bench.py's last resort, in C: a line to write to a descriptor if the process ends through exit() before bench.py has
 * printed its ONE JSON line (a native exit() of the library inside an optional leg: HIP_CALL failures end the process like the
 * reference's cudaCheckError).  A plain atexit handler inside this small shared object -- not a Python callable registered with libc:
 * that one would be called after the interpreter is gone on every NORMAL exit, which forced bench.py to leave through os._exit() and
 * cost a profiler (rocprofv3) its own exit handlers.
 *     gcc -O2 -shared -fPIC tools/exit_line.c -o tools/_build/libexit_line.so       (tools/bench_legs.py builds it on first use) */
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static char* g_text = NULL;
static size_t g_len = 0;
static int g_fd = -1;
static int g_registered = 0;

static void at_exit(void)
{
    if (g_fd >= 0 && g_text != NULL && g_len > 0) {
        size_t done = 0;
        while (done < g_len) {
            const ssize_t r = write(g_fd, g_text + done, g_len - done);
            if (r <= 0) break;
            done += (size_t)r;
        }
        g_len = 0;
    }
}

/* the line to print at exit (copied); replaces an earlier one */
void exit_line_set(int fd, const char* text, size_t len)
{
    char* copy = (char*)malloc(len + 1);
    if (copy == NULL) return;
    memcpy(copy, text, len);
    copy[len] = 0;
    g_len = 0;                      /* (an exit between the two stores prints nothing rather than a torn line) */
    free(g_text);
    g_text = copy;
    g_fd = fd;
    g_len = len;
    if (!g_registered) { atexit(at_exit); g_registered = 1; }
}

/* the line has gone out by other means: nothing to print at exit */
void exit_line_clear(void) { g_len = 0; }
