#!/bin/bash
# A/B builds: tools/variants.sh <source.hip> name1 "-DFLAG=1 ..." name2 "..." -> contourist_amd/lib/variants/lib_<name>.so
# (the other objects come from the regular build; load one with CX_DEBUG=1 CX_LIB_PATH=...)
set -e
cd "$(dirname "$0")/.."
python -m contourist_amd.build > /dev/null
src=$1; shift
mkdir -p contourist_amd/lib/variants
while [ $# -gt 1 ]; do
  name=$1; flags=$2; shift 2
  obj=contourist_amd/lib/variants/$(basename $src).$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c contourist_amd/csrc/$src -o $obj -Wall -Wno-unused-function $flags
  others=$(ls contourist_amd/lib/*.o | grep -v "/$src.o")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o contourist_amd/lib/variants/lib_$name.so $obj $others
  echo built $name
done
