#!/bin/bash
# Developer tool (here, CPU box): build the library of another git revision into rslqr_amd/librslqr_amd_prev.so for the
# same-box A/B scripts (tools/ab_prev.sh, tools/ab_prev_config5.sh).    bash tools/mk_prev.sh [git-ref, default HEAD]
set -e
ref=${1:-HEAD}
root=$(git rev-parse --show-toplevel)
tmp=$(mktemp -d /tmp/prevbuild.XXXXXX)
git -C "$root" archive "$ref" | tar -x -C "$tmp"
(cd "$tmp" && python3 -m rslqr_amd.build > /dev/null)
cp "$tmp/rslqr_amd/librslqr_amd.so" "$root/rslqr_amd/librslqr_amd_prev.so"
rm -rf "$tmp"
echo "rslqr_amd/librslqr_amd_prev.so = $ref ($(git -C "$root" rev-parse --short "$ref"))"
