#!/bin/bash
# Builds dietgpu_amd/lib/v_<name>.so from the sources of a git revision (A/B against an earlier commit).
# Usage: tools/build_variant_git.sh <name> <rev>
set -e
name=$1; rev=$2
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/variant_git_$name
rm -rf $tmp && mkdir -p $tmp && (cd $root && git archive $rev dietgpu_amd/csrc include | tar -x -C $tmp)
(cd $tmp && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $root/dietgpu_amd/lib/v_$name.so dietgpu_amd/csrc/capi.hip)
echo built dietgpu_amd/lib/v_$name.so from $rev
