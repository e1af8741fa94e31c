timeout 900 python -m pytest tests/test_fastq_text_gpu.py -x -q -m gpu -k "damage_deep or tile_boundary" 2>&1 | tail -15
