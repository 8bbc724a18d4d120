#!/bin/bash
# Builds dietgpu_amd/lib/v_<name>.so from a scratch copy of the sources with patches (git apply) and/or sed
# expressions applied.  Usage: tools/build_variant_patch.sh <name> [-p <patch>]... [-s <file under csrc> <sed-expr>]...
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=/tmp/variant_$name
rm -rf $tmp && mkdir -p $tmp/dietgpu_amd && cp -r $root/dietgpu_amd/csrc $tmp/dietgpu_amd/ && cp -r $root/include $tmp/
while [ $# -ge 2 ]; do
  case $1 in
    -p) (cd $tmp && git apply --include='dietgpu_amd/csrc/*' --include='include/*' "$(cd $root && realpath $2)"); shift 2;;
    -s) sed -i "$3" $tmp/dietgpu_amd/csrc/$2; shift 3;;
    *) echo "bad argument $1"; exit 1;;
  esac
done
(cd $tmp && diff -r $root/dietgpu_amd/csrc dietgpu_amd/csrc | head -8; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o $root/dietgpu_amd/lib/v_$name.so dietgpu_amd/csrc/capi.hip)
echo built dietgpu_amd/lib/v_$name.so
