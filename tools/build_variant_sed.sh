#!/bin/bash
# Builds dietgpu_amd/lib/v_<name>.so from a scratch copy of the sources with sed expressions applied (the product
# tree has no compile-time knobs: A/B candidates are edits).  Usage: tools/build_variant_sed.sh <name> <file> <sed-expr> [<file> <sed-expr> ...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/variant_$name
rm -rf $tmp && mkdir -p $tmp/dietgpu_amd && cp -r $root/dietgpu_amd/csrc $tmp/dietgpu_amd/ && cp -r $root/include $tmp/
while [ $# -ge 2 ]; do sed -i "$2" $tmp/dietgpu_amd/csrc/$1; shift 2; done
(cd $tmp && diff -r $root/dietgpu_amd/csrc dietgpu_amd/csrc | head -20; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $root/dietgpu_amd/lib/v_$name.so dietgpu_amd/csrc/capi.hip)
echo built dietgpu_amd/lib/v_$name.so
