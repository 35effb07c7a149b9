#!/bin/bash
# Sanitizer evidence of a round (build container only: the pool has no GPU sanitizers).  Run it LAST, after tools/profile_round.sh on the GPU box and after the
# last code commit:   bash tools/sanitize_round.sh r05
# Runs `make -C tests/hostcheck sanitize`, writes profiles/<tag>_sanitizers.txt headed by bench.source_hash() of the tree it ran on, and FAILS when that hash
# differs from the one the counter files of the same round carry (profiles/<tag>_pmc_traffic.json, <tag>_pmc_traffic_throughput.json): evidence files of a
# round must describe one and the same tree.
set -o pipefail
tag=${1:-r05}
root=$(cd "$(dirname "$0")/.." && pwd)
cd "$root" || exit 1
hash=$(python3 -c "import bench; print(bench.source_hash())") || exit 1
out=profiles/${tag}_sanitizers.txt
log=$(mktemp)
make -C tests/hostcheck sanitize > "$log" 2>&1; rc=$?
{
  echo "make -C tests/hostcheck sanitize - ${tag}, build container ($(nproc) CPUs, no GPU), on the final sources (bench.source_hash ${hash})."
  echo "Abridged: compiler command lines cut at 260 characters; everything else as printed."
  grep -v "^make\[" "$log" | cut -c1-260
  echo "sanitize rc $rc"
} > "$out"
rm -f "$log"
[ $rc -eq 0 ] || { echo "sanitize failed (rc $rc): see $out"; exit $rc; }
bad=0
for f in profiles/${tag}_pmc_traffic.json profiles/${tag}_pmc_traffic_throughput.json; do
  [ -f "$f" ] || { echo "$f missing: take the counter passes (tools/profile_round.sh $tag pmc pmcthr) before the sanitizer evidence"; bad=1; continue; }
  h=$(python3 -c "import json,sys; print(json.load(open('$f')).get('_meta',{}).get('source_hash'))")
  [ "$h" = "$hash" ] || { echo "$f was taken from sources $h, the tree is $hash: re-take it (one evidence pass per round, after the last code commit)"; bad=1; }
done
[ $bad -eq 0 ] && echo "$out: sources $hash, same as the counter files of $tag"
exit $bad
