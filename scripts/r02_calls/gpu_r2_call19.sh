#!/bin/bash
export TMPDIR=/tmp
timeout 300 python scripts/micro/encode_b1.py 2>&1 | grep "B="
