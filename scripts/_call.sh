hipcc -O3 --offload-arch=gfx950 scripts/micro/cumask.hip -o /tmp/cumask 2>/dev/null
for nb in 16 32 64; do /tmp/cumask $nb; done 2>&1 | tee gpurun_out/cumask.txt
echo "=== plain"; timeout 900 bash scripts/prof_r06_plain.sh b 2>&1 | grep -E "run [0-9]|^\[hpn\] p|lanes|Finished" | cut -c1-220
