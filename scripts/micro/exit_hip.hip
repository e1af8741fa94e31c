// exit_hip.hip -- what a process's END costs after HIP work: the parent times from the child's last line to waitpid's return.
//   /tmp/exit_hip <scenario>   0: init only, 1: + stream, 2: + 3 streams, 3: + 12 GiB device, 4: + 264 MiB pinned, 5: all + 40 GiB touched
// Also: two threads creating streams at once (do they overlap?).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include <thread>
static double now() { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
int main(int argc, char **argv)
{
    for (int sc = 0; sc <= 6; ++sc) {
        int fd[2];
        if (pipe(fd)) return 1;
        const double t_fork = now();
        pid_t p = fork();
        if (p == 0) {
            close(fd[0]);
            (void)hipInit(0);
            const double t_init = now();
            hipStream_t s[3];
            double t_par = 0;
            if (sc == 6) {   // two streams made by two threads at once
                const double a = now();
                std::thread t1([&] { (void)hipSetDevice(0); (void)hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking); });
                std::thread t2([&] { (void)hipSetDevice(0); (void)hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking); });
                t1.join(), t2.join();
                t_par = now() - a;
            }
            if (sc >= 1 && sc != 6) (void)hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking);
            if ((sc >= 2 && sc != 6) ) { (void)hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking); (void)hipStreamCreateWithFlags(&s[2], hipStreamNonBlocking); }
            void *d = nullptr, *h = nullptr;
            if (sc == 3 || sc == 5) { (void)hipMalloc(&d, (size_t)(sc == 5 ? 40 : 12) << 30); (void)hipMemset(d, 1, (size_t)(sc == 5 ? 40 : 12) << 30); (void)hipDeviceSynchronize(); }
            if (sc == 4 || sc == 5) { (void)hipHostMalloc(&h, (size_t)264 << 20, hipHostMallocDefault); }
            double v[3] = {now(), t_init, t_par};
            if (write(fd[1], v, sizeof v) != (ssize_t)sizeof v) _exit(2);
            _exit(0);
        }
        close(fd[1]);
        double v[3] = {0, 0, 0};
        if (read(fd[0], v, sizeof v) != (ssize_t)sizeof v) return 2;
        int st;
        waitpid(p, &st, 0);
        const double t_end = now();
        printf("scenario %d: hipInit %6.1f ms, setup %6.1f ms, exit %6.1f ms%s\n", sc, (v[1] - t_fork) * 1e3, (v[0] - v[1]) * 1e3, (t_end - v[0]) * 1e3,
               sc == 6 ? "  (two streams by two threads at once)" : "");
        if (sc == 6) printf("   the two concurrent hipStreamCreate together: %.1f ms\n", v[2] * 1e3);
        close(fd[0]);
        fflush(stdout);
    }
    return 0;
}
