#!/bin/bash
# GPU box: A/B of library variants on ONE box, per workload.  usage: tools/r02_ab.sh "<variants>" rounds [workload args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
variants=$1; rounds=$2; shift 2
AB_ARGS="$*" tools/ab.sh $rounds $variants
