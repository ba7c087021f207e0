#!/bin/bash
# GPU box: why toy-graph lists leave resident mode
G2S_DEBUG=1 timeout 600 python -m pytest tests/test_gpu_resident.py -x -q -k "toy_graphs" -s 2>&1 | grep -E "goes to the host path|passed|failed" | sort | uniq -c | sort -rn | head -20
