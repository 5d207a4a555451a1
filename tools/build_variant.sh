#!/bin/bash
# (development) a second build of the library with extra compiler flags for ONE source, beside the product library:
#   tools/build_variant.sh prof lsd_regions.hip "-DRGW_PROF"   ->  cube_slam_amd/variants/libcubeslam_prof.so   (use: CUBESLAM_LIB=$PWD/cube_slam_amd/variants/libcubeslam_prof.so)
set -e
cd "$(dirname "$0")/../cube_slam_amd/csrc"
name=$1; src=$2; extra=$3
mkdir -p ../variants
make -s -j8 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fopenmp -Wall -Wno-unused-function -Wno-unused-result -Wno-unused-value -Wno-unused-variable $extra -c $src -o ../variants/${src%.hip}_$name.o
objs=$(ls *.o | grep -v "^${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fopenmp -o ../variants/libcubeslam_$name.so $objs ../variants/${src%.hip}_$name.o
ls -la ../variants/libcubeslam_$name.so
