#!/bin/bash
cd "$(dirname "$0")/.."
timeout 300 python -m pytest tests/test_gpu_benched_shapes.py -m gpu -x -q -p no:cacheprovider -k "device_io" 2>&1 | tail -8
