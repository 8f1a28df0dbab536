#!/bin/bash
# Build an A/B variant of the library into tools/_probe/ :  tools/ab_build.sh <name> [-DFLAG ...]
# then run e.g.  BOXDREAMER_HIP_LIB=tools/_probe/libbd_<name>.so python bench.py   on the same box as the default build.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_probe
FLAGS=$(python -c "from boxdreamer_amd import build; print(' '.join(build.FLAGS))")
objs=""
for f in boxdreamer_amd/csrc/*.hip; do
  o=/tmp/ab_${name}_$(basename $f .hip).o
  hipcc $FLAGS "$@" -I include -c $f -o $o &
  objs="$objs $o"
done
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/_probe/libbd_${name}.so $objs
echo built tools/_probe/libbd_${name}.so
