#!/bin/bash
# build/libsearcharray_hip_probe.so: the library with -DSA_PROBE (cycle counters inside the grouped kernel; scripts/ab.py prints
# them when the library exports sa_debug_probe_read).  Measurement build, not the product.
#   SA_PROBE_EXTRA=-DSA_PROBE_FINE SA_PROBE_SUFFIX=_fine  -> ..._probe_fine.so: sub-sections of the item's preamble, vector loads
#   waited for where they are charged
set -e
cd "$(dirname "$0")/../searcharray_amd/csrc"
mkdir -p ../../build/probe_obj
for f in sa_index sa_build sa_bm25 sa_stage sa_queue sa_sparse sa_ops sa_setops sa_phrase sa_phrase_batch sa_spans sa_vec sa_io sa_comm sa_sort sa_sharded; do
  if [ $f = sa_bm25 ] || [ $f = sa_stage ] || [ $f = sa_spans ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DSA_PROBE ${SA_PROBE_EXTRA} -c $f.hip -o ../../build/probe_obj/$f.o
  else
    cp build/$f.o ../../build/probe_obj/$f.o
  fi
done
OUT=../../build/libsearcharray_hip_probe${SA_PROBE_SUFFIX}.so
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT ../../build/probe_obj/*.o -L/opt/rocm/lib -lrccl -Wl,--disable-new-dtags,-rpath,/opt/rocm/lib
ls -la $OUT
