#!/bin/bash
# Build libgsrast_hip.so of a git ref (default HEAD) into tools/ab/lib_<name>.so for same-box A/B runs (tools/ab_run.sh).  usage: tools/ab_build.sh <name> [ref]
set -e
name=$1; ref=${2:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
git -C "$root" archive "$ref" gs-sr_amd/csrc include | tar -x -C "$tmp"
mkdir -p "$root/tools/ab"
make -C "$tmp/gs-sr_amd/csrc" -j8 OUT="$root/tools/ab/lib_$name.so" >/dev/null
rm -rf "$tmp"
echo "built tools/ab/lib_$name.so from $ref"
