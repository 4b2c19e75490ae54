#!/bin/bash
# bash tools/build_variant.sh <name> <source.hip> <extra hipcc flags...>: red-gnn_amd/libredgnn_<name>.so = the current objects with
# one source recompiled under extra flags (kernel A/B builds; load with RG_LIB=<path>)
set -e
NAME=$1; SRC=$2; shift 2
cd "$(dirname "$0")/.."
C=red-gnn_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Iinclude -I$C "$@" -c $C/$SRC -o /tmp/variant_$NAME.o
OBJS=$(ls $C/build/*.o | grep -v "/${SRC%.hip}.o")
hipcc --offload-arch=gfx950 -shared -fPIC -o red-gnn_amd/libredgnn_$NAME.so $OBJS /tmp/variant_$NAME.o
echo built red-gnn_amd/libredgnn_$NAME.so
