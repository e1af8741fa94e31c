// How fast ONE output file takes bytes on this box: slabs of 128 MiB written by T threads with pwrite (mode 0: what write_slab does) or
// copied into a shared mapping of the extended file (mode 1).   g++ -O2 scripts/micro/write_bw.cpp -o /tmp/write_bw -lpthread && /tmp/write_bw <file> <MiB> <threads> <mode>
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
#include <thread>
#include <vector>
static double now(){timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+1e-9*t.tv_nsec;}
int main(int argc,char**argv){
  const size_t total=(size_t)atoll(argv[2])<<20, slab=(size_t)128<<20; const int T=atoi(argv[3]); const int mode=atoi(argv[4]);
  char*src=(char*)malloc(slab); memset(src,'A',slab);
  int fd=open(argv[1],O_CREAT|O_TRUNC|O_RDWR,0666);
  double t0=now();
  for(size_t at=0;at<total;at+=slab){
    std::vector<std::thread> th; size_t piece=slab/T;
    if(mode==0){ for(int t=0;t<T;++t) th.emplace_back([=]{ size_t lo=t*piece; for(size_t d=lo;d<lo+piece;){ssize_t k=pwrite(fd,src+d,lo+piece-d,at+d); if(k<=0)break; d+=k;} }); }
    else { if(ftruncate(fd,at+slab)) return 1; char*m=(char*)mmap(nullptr,slab,PROT_WRITE|PROT_READ,MAP_SHARED,fd,at); if(m==MAP_FAILED){perror("mmap");return 1;}
      for(int t=0;t<T;++t) th.emplace_back([=]{ memcpy(m+t*piece,src+t*piece,piece); });
      for(auto&x:th)x.join(); th.clear(); munmap(m,slab); }
    for(auto&x:th)x.join();
  }
  double dt=now()-t0; printf("mode %d threads %d: %.2f GB/s\n",mode,T,total/dt/1e9); close(fd); unlink(argv[1]); }
