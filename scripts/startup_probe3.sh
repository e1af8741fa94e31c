# Does a full page cache slow the runtime's start?  scripts/micro/startup_hip.hip before and after 48 GB of files are written and read,
# with and without transparent huge pages for the process  -> gpurun_out/r05/startup_pagecache.txt
mkdir -p gpurun_out/r05 /tmp/spc; out=gpurun_out/r05/startup_pagecache.txt; : > $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 scripts/micro/startup_hip.hip -o /tmp/startup_hip 2>/dev/null
cat /sys/kernel/mm/transparent_hugepage/enabled /sys/kernel/mm/transparent_hugepage/defrag >> $out 2>&1
run() { echo "== $*" >> $out; for i in 1 2 3; do s=$(date +%s%N); env "$@" /tmp/startup_hip > /tmp/st.txt; e=$(date +%s%N); echo "wall $(( (e - s) / 1000000 )) ms  $(grep -E 'hipInit|hipStreamCreate|hipHostMalloc 64' /tmp/st.txt | awk '{printf "%s %s ms; ", $1, $2}')" >> $out; done; }
grep -E "MemFree|^Cached|AnonHuge" /proc/meminfo >> $out
run A=1
run NO_THP=1
for k in 1 2 3 4 5 6; do head -c 8000000000 /dev/zero > /tmp/spc/f$k; done; cat /tmp/spc/f* > /dev/null
grep -E "MemFree|^Cached|AnonHuge" /proc/meminfo >> $out
run A=1
run NO_THP=1
run A=1
rm -rf /tmp/spc
cat $out
