#!/bin/bash
# Build another flavour of libbgsa_hip.so from the SAME sources with generator / make variables changed, for same-box A/B runs:
#     scripts/build_variant.sh NAME [VAR=VALUE ...]        -> bgsa_amd/_ab/libbgsa_hip_NAME.so   (git-ignored; travels with gpurun)
# e.g.  scripts/build_variant.sh split4 BGSA_GEN_MYERS_SPLIT=4
# The tree is copied to a temporary directory, generated and compiled there: the working tree's .inc files and objects stay as they are.
set -euo pipefail
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/bgsa_variant.XXXXXX)
trap 'rm -rf "$tmp"' EXIT
mkdir -p "$tmp/bgsa_amd" "$root/bgsa_amd/_ab"
cp -r "$root/include" "$tmp/include"
mkdir "$tmp/bgsa_amd/csrc"
cp "$root"/bgsa_amd/csrc/*.{hip,inl,h,py} "$root/bgsa_amd/csrc/Makefile" "$tmp/bgsa_amd/csrc/"
( cd "$tmp/bgsa_amd/csrc" && env "$@" python3 gen_rows_asm.py && env "$@" make -j8 all >/dev/null )
cp "$tmp/bgsa_amd/libbgsa_hip.so" "$root/bgsa_amd/_ab/libbgsa_hip_$name.so"
echo "built bgsa_amd/_ab/libbgsa_hip_$name.so ($*)"
