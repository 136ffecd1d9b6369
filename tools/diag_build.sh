#!/bin/bash
# tools/diag_build.sh <name> <file.hip> <-DFLAG ...>: a copy of the library with ONE translation unit rebuilt with
# diagnostic defines -> tools/diag/libsgc_<name>.so (timing experiments only; results of such builds are garbage)
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p tools/diag/obj_$name
objs=""
for f in sgcdet_amd/csrc/*.hip; do
  o=sgcdet_amd/csrc/$(basename ${f%.hip}).o
  if [ "$(basename $f)" == "$src" ]; then
    o=tools/diag/obj_$name/$(basename ${f%.hip}).o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c $f -o $o -Xclang -target-feature -Xclang -packed-fp32-ops "$@"
  fi
  objs="$objs $o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o tools/diag/libsgc_$name.so
echo tools/diag/libsgc_$name.so
