# fixed per-process HIP cost on this box, with a few runtime settings  -> gpurun_out/r05/startup.txt
mkdir -p gpurun_out/r05; out=gpurun_out/r05/startup.txt; : > $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/startup_hip.hip -o /tmp/startup_hip 2>/dev/null
run() { echo "== $*" >> $out; for i in 1 2; do s=$(date +%s%N); env "$@" /tmp/startup_hip > /tmp/st.txt; e=$(date +%s%N); echo "wall $(( (e - s) / 1000000 )) ms" >> $out; done; cat /tmp/st.txt >> $out; }
run A=1
run HSA_ENABLE_INTERRUPT=0
run ROCR_VISIBLE_DEVICES=0
run GPU_MAX_HW_QUEUES=2
run HIP_ENABLE_DEFERRED_LOADING=1 AMD_LOG_LEVEL=0
run HSA_ENABLE_SDMA=0
s=$(date +%s%N); /bin/true; e=$(date +%s%N); echo "/bin/true wall $(( (e - s) / 1000000 )) ms" >> $out
s=$(date +%s%N); highperformancengs_amd/bin/fastq_count -h > /dev/null 2>&1; e=$(date +%s%N); echo "fastq_count -h wall $(( (e - s) / 1000000 )) ms" >> $out
ls /sys/class/kfd/kfd/topology/nodes | wc -l >> $out
cat $out
