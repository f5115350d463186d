#!/bin/bash
# TEST INFRASTRUCTURE -- builds the REFERENCE itself (softwaredoug/searcharray, the checkout under
# /root/reference) into oracle/_ref/ so that it can be imported where /root/reference does not exist
# (the GPU box): the package's .py files are staged and its eight Cython extensions are compiled with the
# reference's own setup.py (gcc -O2, its flags), exactly as SURVEY.md appendix B verified.
#
#   oracle/_ref/            is listed in .gitignore (never enters the history: no reference source is
#                           committed) but NOT in .gpurunignore, so the built tree travels with a gpurun push
#   oracle/ref_loader.py    imports it; bench.py's cpu_baseline leg times it ("kind": "reference"),
#                           tests/test_oracle_golden.py re-checks the committed goldens against it when present
#
# Only tests/, __graft_entry__.build()/smoke() and bench.py's cpu_baseline leg use anything under oracle/.
set -euo pipefail
SRC=${REF_SRC:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
DST=$HERE/_ref
if [ ! -d "$SRC/searcharray" ]; then
    echo "build_ref: no reference checkout at $SRC -- keeping whatever is in $DST" >&2
    exit 0
fi
STAMP=$DST/.built_from
WANT=$(cd "$SRC" && find searcharray setup.py -type f \( -name '*.py' -o -name '*.pyx' -o -name '*.pxd' -o -name '*.h' -o -name '*.c' \) -print0 | sort -z | xargs -0 sha1sum | sha1sum | cut -d' ' -f1)
if [ -f "$STAMP" ] && [ "$(cat "$STAMP")" = "$WANT" ] && ls "$DST"/searcharray/bm25/*.so >/dev/null 2>&1; then
    exit 0                                   # up to date
fi
rm -rf "$DST"
mkdir -p "$DST"
cp -r "$SRC/searcharray" "$SRC/setup.py" "$DST/"
for f in README.md pyproject.toml setup.cfg MANIFEST.in requirements.txt; do
    [ -f "$SRC/$f" ] && cp "$SRC/$f" "$DST/" || true
done
chmod -R u+w "$DST"
( cd "$DST" && python setup.py -q build_ext --inplace >"$DST/build.log" 2>&1 ) || { tail -30 "$DST/build.log" >&2; exit 1; }
rm -rf "$DST/build"
find "$DST" -name '*.c' -newer "$DST/setup.py" -path '*searcharray*' -size +100k -delete 2>/dev/null || true   # cythonized C, not needed at run time
echo "$WANT" > "$STAMP"
echo "build_ref: reference built into $DST"
