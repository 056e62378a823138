#!/bin/bash
# Usage (on the GPU box): scripts/ab_trav.sh <tag> ...  -- k_region_traversal time per library variant (scripts/build_variant.sh)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
for SPEC in "$@"; do
  TAG=$SPEC
  if [ "$TAG" = "default" ]; then unset OHMHIP_LIB; else export OHMHIP_LIB=$PWD/ohm_amd/lib/variants/libohmhip_$TAG.so; fi
  echo "== $SPEC"
  timeout 250 scripts/prof_cmd.sh scripts/traversal_probe.py < /dev/null 2>&1 | grep -E "k_region_traversal|traversal:"
  timeout 60 python scripts/traversal_probe.py 2>&1 | tail -1
done
