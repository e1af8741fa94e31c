"""CPU: small pieces of the tools' host side on their own (csrc/host/cpus.hpp), compiled into a throw-away program.
  * lanes_worth: how many devices an input gets -- k lanes stream in T1 / k and cost k x c to set up, least at sqrt(T1 / c);
  * parse_cpulist: sysfs CPU lists ("64-127,192-255") cut to what the process may use;
  * bind_before_runtime / initial_cpus on a host without /dev/dri: nothing moves.
No reference counterpart: the reference starts kt_for's threads wherever the scheduler puts them (klib/kthread.c:48)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r'''
#include "../highperformancengs_amd/csrc/host/cpus.hpp"
using namespace hpn;
int main()
{
    const uint64_t GB = 1000000000ull, plain = (uint64_t)3200 << 20, bam = (uint64_t)2560 << 20;
    printf("%d %d %d %d %d %d\n", lanes_worth(1 * GB, plain, 8), lanes_worth(16 * GB, plain, 8), lanes_worth(64 * GB, plain, 8),
           lanes_worth(200 * GB, plain, 8), lanes_worth(200 * GB, plain, 4), lanes_worth(16 * GB, plain, 1));
    printf("%d %d\n", lanes_worth(10 * GB + 600000000ull, bam, 8), lanes_worth(93 * GB, bam, 8));
    cpu_set_t allowed, out;
    CPU_ZERO(&allowed);
    for (int c = 0; c < 200; ++c) CPU_SET(c, &allowed);
    printf("%d ", parse_cpulist("64-127,192-255\n", allowed, &out));
    printf("%d %d %d %d\n", CPU_ISSET(63, &out), CPU_ISSET(64, &out), CPU_ISSET(199, &out), CPU_ISSET(200, &out));
    printf("%d %d\n", parse_cpulist("3\n", allowed, &out), parse_cpulist("", allowed, &out));
    cpu_set_t before, after;
    sched_getaffinity(0, sizeof before, &before);
    bind_before_runtime();
    sched_getaffinity(0, sizeof after, &after);
    printf("%d\n", CPU_EQUAL(&before, &after) || CPU_COUNT(&after) >= 4);
    return 0;
}
'''


def test_host_units(tmp_path):
    src = tmp_path / "u.cpp"
    src.write_text(SRC.replace("../highperformancengs_amd", os.path.join(ROOT, "highperformancengs_amd")))
    exe = tmp_path / "u"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-lpthread"])
    out = subprocess.run([str(exe)], stdout=subprocess.PIPE, check=True).stdout.decode().split("\n")
    assert out[0].split() == ["1", "2", "4", "8", "4", "1"]       # 1 GB: one device; 16 GB: two; 64 GB: four; 200 GB: all eight (or all four)
    assert out[1].split() == ["2", "6"]                               # the C4-shaped 10.6 GB BAM: two workers; the 93 GB one: six
    assert out[2].split() == ["72", "0", "1", "1", "0"]              # 64..127 + 192..199 of the 200 allowed CPUs
    assert out[3].split() == ["1", "0"]
    assert out[4].strip() == "1"
