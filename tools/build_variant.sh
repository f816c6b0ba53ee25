#!/bin/bash
# A variant of the library with other flags for traverse_wide.hip, beside the default one (A/B runs: tools/exp_libs.sh).
#   bash tools/build_variant.sh <tag> <flags...>   -> yart_amd/libyart_amd_<tag>.so
tag=$1; shift
cd "$(dirname "$0")/../yart_amd/csrc" || exit 1
make >/dev/null || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result "$@" --offload-arch=gfx950 -c traverse_wide.hip -o _obj/traverse_wide_$tag.o || exit 1
objs=$(ls _obj/*.o | grep -v "traverse_wide" | tr '\n' ' ')
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $objs _obj/traverse_wide_$tag.o -ldl -o ../libyart_amd_$tag.so && echo built ../libyart_amd_$tag.so
