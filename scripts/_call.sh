cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout 2400 python -m pytest tests -x -q -m gpu > /tmp/pt.log 2>&1; echo "rc=$?"; grep -E "passed|failed|rror" /tmp/pt.log | tail -5
