cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 1500 python -m pytest tests/test_cli_gpu.py -x -q -m gpu 2>&1 | tail -3
R06_BASE=head R06_OUT=r06_gz3 R06_FINDDIAG= bash scripts/r06_gz.sh 2>&1 | tail -40
