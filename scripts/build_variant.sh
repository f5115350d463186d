#!/bin/bash
# build/libsearcharray_hip_<name>.so: the library with ONE source (VARIANT_FILE, default sa_stage) compiled with extra flags (A/B builds for
# scripts/ab.py --libs / SA_LIB; measurement builds, not the product).    VARIANT_FILE=sa_spans scripts/build_variant.sh rows8 -DSA_SPAN_FROWS=8
set -e
NAME=$1; shift
cd "$(dirname "$0")/../searcharray_amd/csrc"
mkdir -p ../../build/var_$NAME
for f in sa_index sa_build sa_bm25 sa_stage sa_queue sa_sparse sa_ops sa_setops sa_phrase sa_phrase_batch sa_spans sa_vec sa_io sa_comm sa_sort sa_sharded; do
  if [ $f = ${VARIANT_FILE:-sa_stage} ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off "$@" -c $f.hip -o ../../build/var_$NAME/$f.o
  else
    cp build/$f.o ../../build/var_$NAME/$f.o
  fi
done
OUT=../../build/libsearcharray_hip_$NAME.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT ../../build/var_$NAME/*.o -L/opt/rocm/lib -lrccl -Wl,--disable-new-dtags,-rpath,/opt/rocm/lib
ls -la $OUT
