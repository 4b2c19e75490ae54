#!/bin/bash
# A/B of kernel builds on one box: bash tools/r3_ab.sh <script.py and args, quoted> <variant names...>
CMD=$1; shift
for v in "$@"; do
  RG_LIB=$PWD/red-gnn_amd/libredgnn_$v.so python $CMD 2>/dev/null | tail -1
done
