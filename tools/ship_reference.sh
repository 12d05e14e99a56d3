#!/bin/bash
# The unmodified reference on the GPU box -- OUTSIDE the repository.  gpurun ships /root/repo only, so the tree travels
# as one git-ignored archive next to the sources and is unpacked to /tmp on the box; nothing of it is ever committed.
#   here:        bash tools/ship_reference.sh pack        (writes .ref_ship.tgz, listed in .gitignore)
#   on the box:  bash tools/ship_reference.sh unpack      (-> /tmp/reference; export SCNERF_REFERENCE_ROOT=/tmp/reference)
#   here, after: bash tools/ship_reference.sh clean
set -e
cd "$(dirname "$0")/.."
case "$1" in
  pack)   tar czf .ref_ship.tgz -C /root --exclude=.git --exclude='*.png' --exclude='*.jpg' --exclude='*.gif' reference ;;
  unpack) mkdir -p /tmp && tar xzf .ref_ship.tgz -C /tmp && echo /tmp/reference ;;
  clean)  rm -f .ref_ship.tgz ;;
  *) echo "usage: $0 pack|unpack|clean"; exit 2 ;;
esac
