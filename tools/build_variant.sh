#!/bin/bash
# tools/build_variant.sh TAG [-DMACRO=V ...] : builds circkit_amd/libcirckit_hip_TAG.so for tools/try_variants.sh
set -e
R=$(cd $(dirname $0)/.. && pwd)
tag=$1; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC "$@" -o $R/circkit_amd/libcirckit_hip_$tag.so \
  $R/circkit_amd/csrc/circkit_hip.hip $R/circkit_amd/csrc/fasta_host.cpp
