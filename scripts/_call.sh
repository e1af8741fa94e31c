timeout 300 python -m pytest tests/test_abi_c.py -q -m gpu -k rccl 2>&1 | tail -40
( time timeout 3300 python -m pytest tests -q -m gpu 2>&1 | tail -15 ) 2>&1 | tail -20
