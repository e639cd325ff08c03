#!/bin/bash
# Build libgsrast_hip.so of the WORKING TREE with extra flags into tools/ab/lib_<name>.so.  usage: tools/ab_build_flags.sh <name> "<BLEND_EXTRA flags>" ["<PRE_EXTRA flags: preprocess / binning / extra / tsdf / mvloss units>"] ["<EXTRA: every unit>"]
set -e
name=$1; flags=$2; pre=$3; all=$4
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/gs-sr_amd" "$root/tools/ab"
cp -r "$root/gs-sr_amd/csrc" "$tmp/gs-sr_amd/csrc"; cp -r "$root/include" "$tmp/include"
rm -f "$tmp"/gs-sr_amd/csrc/*.o
make -C "$tmp/gs-sr_amd/csrc" -j8 OUT="$root/tools/ab/lib_$name.so" BLEND_EXTRA="$flags" PRE_EXTRA="$pre" EXTRA="$all" >/dev/null
rm -rf "$tmp"
echo "built tools/ab/lib_$name.so with BLEND_EXTRA=$flags"
