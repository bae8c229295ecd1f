python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python scripts/dev_encode_latency.py
