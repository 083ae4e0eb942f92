#!/bin/bash
export TMPDIR=/tmp
( timeout 70 python -m pytest tests/test_gpu_golden.py -m gpu -q --timeout=60 -k "device_resident" 2>&1 | tail -12 | cut -c1-250 )
