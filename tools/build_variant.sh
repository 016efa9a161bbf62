#!/bin/bash
# Builds the CURRENT csrc/ tree into build_ab/lib_<name>.so without touching the library in the package (for same-box A/B
# runs with tools/ab_lib.sh).   usage: tools/build_variant.sh <name> [extra compiler flags]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
dst=$root/build_ab/v/$name
mkdir -p $dst/mcevidence_amd $dst/include
cp -a $root/include/. $dst/include/
mkdir -p $dst/mcevidence_amd/csrc
cp -a $root/mcevidence_amd/csrc/*.hip $root/mcevidence_amd/csrc/*.hpp $root/mcevidence_amd/csrc/*.cpp $root/mcevidence_amd/csrc/Makefile $dst/mcevidence_amd/csrc/
make -C $dst/mcevidence_amd/csrc -j7 CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $*" ../libmcevidence_hip.so > $dst/build.log 2>&1 || { tail -20 $dst/build.log; exit 1; }
cp $dst/mcevidence_amd/libmcevidence_hip.so $root/build_ab/lib_$name.so
echo "built build_ab/lib_$name.so"
