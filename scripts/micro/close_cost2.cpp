// close_cost2: the bedGraph writer's pattern -- FILE*, fwrite of 16 MiB slices from ANOTHER thread, fclose on the main thread;
// with and without a 3 GB file of the same name already there (O_TRUNC).   g++ -O2 scripts/micro/close_cost2.cpp -o /tmp/close_cost2 -lpthread
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <thread>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main(int argc, char **argv)
{
    const size_t total = (size_t)atoll(argv[2]) << 20, slab = (size_t)16 << 20;
    char *src = (char *)malloc(slab);
    memset(src, 'A', slab);
    for (int round = 0; round < 3; ++round) {     // round 0: no file there; 1, 2: the file of the round before is there
        double t0 = now();
        int fd = open(argv[1], O_CREAT | O_WRONLY | O_TRUNC, 0666);
        FILE *f = fdopen(fd, "wb");
        const double t_open = now() - t0;
        t0 = now();
        std::thread w([&] { for (size_t at = 0; at < total; at += slab) fwrite(src, 1, slab - (at ? 0 : 13), f); });
        w.join();
        const double t_w = now() - t0;
        t0 = now();
        fclose(f);
        const double t_c = now() - t0;
        printf("round %d: open(O_TRUNC) %6.1f ms  fwrite %7.1f ms  fclose %6.1f ms\n", round, t_open * 1e3, t_w * 1e3, t_c * 1e3);
    }
    unlink(argv[1]);
}
