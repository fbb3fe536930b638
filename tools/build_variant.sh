#!/bin/bash
# Experiments: build a variant of libsslam_hip.so in which ONE source file is replaced (the others come from csrc/_obj).
#   tools/build_variant.sh NAME path/to/variant_of_X.hip X.hip [extra hipcc flags]   ->  tools/microbench/_variants/NAME.so
# (run `make -C semantic-slam-master_amd/csrc` first; the .so is git-ignored and travels to the GPU box with gpurun)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd); C=$ROOT/semantic-slam-master_amd/csrc; OUT=$ROOT/tools/microbench/_variants
NAME=$1; SRC=$2; REPL=${3%.hip}; shift 3
mkdir -p $OUT/_o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
    -I$C -I$ROOT/include "$@" -x hip -c $SRC -o $OUT/_o/$NAME.o
OBJS=$(ls $C/_obj/*.o | grep -v "/$REPL.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/$NAME.so $OBJS $OUT/_o/$NAME.o
echo $OUT/$NAME.so
