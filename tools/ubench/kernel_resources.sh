#!/bin/bash
# registers / scratch / LDS of every kernel in a shared library's gfx950 code objects:  tools/ubench/kernel_resources.sh lib.so [name filter]
set -e
so=$(readlink -f "$1"); filt=${2:-.}
d=$(mktemp -d /tmp/kres.XXXXXX)
cp "$so" "$d/lib.so"
(cd "$d" && /opt/rocm/lib/llvm/bin/llvm-objdump --offloading lib.so > /dev/null)
for f in "$d"/*amdgcn*; do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
    name = g("name")
    if re.search(sys.argv[1], name):
        print("%-90s vgpr %s agpr %s sgpr %s scratch %s lds %s spills v%s s%s" % (name[:90], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count"), g("sgpr_spill_count")))
' "$filt"
done
echo "$d"
