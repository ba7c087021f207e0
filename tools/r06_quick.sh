#!/bin/bash
# GPU box: quick check of a build — a parity subset, the C2 bench line with C3 beside it (usage: r06_quick.sh [pytest -k expr])
K=${1:-"k31_default or c2_ or resident or toy"}
timeout 900 python -m pytest tests -m gpu -x -q -k "$K" 2>&1 | tail -4
timeout 300 python bench.py --no-cpu-baseline | python tools/bsum.py C2
