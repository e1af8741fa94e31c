HPN_AB_STRETCHES=6144:6144 bash scripts/ab_inflate.sh diag_NOASSEMBLE diag_NOFETCH diag_NOJUMP diag_NOPUT
grep -v "^Traceback\|^  File\|^    " gpurun_out/ab_inflate.txt | cut -c1-330
