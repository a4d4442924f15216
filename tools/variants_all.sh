#!/bin/bash
# A/B builds of the WHOLE library with extra flags: tools/variants_all.sh name1 "-DFLAG=1 ..." name2 "..." -> lib/variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p contourist_amd/lib/variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  objs=""
  for src in contourist_amd/csrc/*.hip; do
    obj=contourist_amd/lib/variants/$(basename $src).$name.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $src -o $obj -Wall -Wno-unused-function $flags &
    objs="$objs $obj"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o contourist_amd/lib/variants/lib_$name.so $objs
  echo built $name
done
