#!/bin/bash
# A/B of CODE changes on one box: export a git revision into tools/_ab/<name> (its own package + built library; git-ignored, travels
# with the gpurun snapshot) and alternate bench runs with tools/ab_code.sh.      usage: tools/mkbase.sh <rev> [name=base]
set -e
rev=$1; name=${2:-base}
root=$(cd "$(dirname "$0")/.." && pwd)
dst=$root/tools/_ab/$name
rm -rf "$dst"; mkdir -p "$dst"
git -C "$root" archive "$rev" -- vipformer_amd include bench.py tests/helpers.py tests/__init__.py tests/golden oracle profiles | tar -x -C "$dst"
(cd "$dst" && python3 -m vipformer_amd.build > build.log 2>&1 && tail -1 build.log)
echo "$rev" > "$dst/REV"
