// What does an output file of a few GB cost beyond its write() calls on this box?  g++ -O2 scripts/micro/close_cost.cpp -o /tmp/close_cost -lpthread
//   /tmp/close_cost <file> <MiB>   -- write in 16 MiB calls (one thread), then time close(); the same with the size set first
//   (ftruncate / fallocate), and with 4 / 8 threads of pwrite.
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <thread>
#include <vector>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    const size_t total = (size_t)atoll(argv[2]) << 20, slab = (size_t)16 << 20;
    char *src = (char *)malloc(slab);
    memset(src, 'A', slab);
    for (int mode = 0; mode < 5; ++mode) {
        int fd = open(argv[1], O_CREAT | O_TRUNC | O_WRONLY, 0666);
        double t0 = now();
        if (mode == 1 && ftruncate(fd, (off_t)total)) return 1;
        if (mode == 2 && posix_fallocate(fd, 0, (off_t)total)) perror("fallocate");
        const double t_pre = now() - t0;
        t0 = now();
        if (mode < 3) {
            for (size_t at = 0; at < total; at += slab)
                if (write(fd, src, slab) != (ssize_t)slab) return 2;
        } else {
            const int T = mode == 3 ? 4 : 8;
            for (size_t at = 0; at < total; at += slab) {
                std::vector<std::thread> th;
                const size_t piece = slab / T;
                for (int t = 0; t < T; ++t)
                    th.emplace_back([=] { for (size_t d = t * piece; d < (t + 1) * piece;) { ssize_t k = pwrite(fd, src + d, (t + 1) * piece - d, at + d); if (k <= 0) break; d += k; } });
                for (auto &x : th) x.join();
            }
        }
        const double t_w = now() - t0;
        t0 = now();
        close(fd);
        const double t_c = now() - t0;
        t0 = now();
        unlink(argv[1]);
        const double t_u = now() - t0;
        const char *nm[] = {"write 16 MiB x1", "ftruncate first", "fallocate first", "pwrite x4", "pwrite x8"};
        printf("%-16s pre %6.1f ms  write %7.1f ms (%5.2f GB/s)  close %6.1f ms  unlink %6.1f ms\n", nm[mode], t_pre * 1e3, t_w * 1e3, total / t_w / 1e9, t_c * 1e3, t_u * 1e3);
    }
}
