# Which CPUs feed the GPU fastest?  bam_sliding_count's ingest under CPU / memory placements  -> gpurun_out/r05/numa_probe.txt
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O; out=$PWD/$O/numa_probe.txt; : > $out
B=$PWD/highperformancengs_amd/bin
lscpu | grep -E "NUMA|Socket|Model name|^CPU\(s\)" >> $out
which numactl taskset >> $out 2>&1
for d in /sys/class/drm/card*/device; do echo "$d numa_node=$(cat $d/numa_node 2>/dev/null) local_cpulist=$(cat $d/local_cpulist 2>/dev/null)" >> $out; done
cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -E "^(cpu_cores_count|simd_count|numa|location_id)" | paste - - - | head -12 >> $out
grep -E "Cpus_allowed_list|Mems_allowed_list" /proc/self/status >> $out
cd /tmp/r05in || exit 1
t() { l=$1; shift; for i in 1 2; do rm -f s.txt; s=$(date +%s%N); "$@" > /dev/null 2> err.txt; e=$(date +%s%N); echo "$l : $(( (e - s) / 1000000 )) ms   $(grep -E 'read-ahead|ingest done|GPU stream open' err.txt | sed 's/\[hpn\] BGZF read-ahead: //' | tr '\n' ' ')" >> $out; done; }
export HPN_TIMING=2 HPN_NGPU=1
t "default" $B/bam_sliding_count -w 20000 -o s hg38.bam
nodes=$(lscpu | grep -E "NUMA node[0-9]+ CPU" | wc -l)
for n in $(seq 0 $((nodes - 1))); do
  cpus=$(lscpu | grep -E "NUMA node$n CPU" | awk '{print $NF}')
  t "taskset node$n ($cpus)" taskset -c $cpus $B/bam_sliding_count -w 20000 -o s hg38.bam
  which numactl > /dev/null 2>&1 && t "numactl cpu+mem node$n" numactl --cpunodebind=$n --membind=$n $B/bam_sliding_count -w 20000 -o s hg38.bam
done
cat $out
rm -rf /tmp/r05in     # (boxes are reused: leave the disk as it was found)
