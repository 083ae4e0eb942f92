#!/bin/bash
export TMPDIR=/tmp
( timeout 100 python -m pytest tests/test_gpu_golden.py -m gpu -q --timeout=120 2>&1 | tail -2 )
( timeout 60 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 )
