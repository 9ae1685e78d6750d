#!/bin/bash
# profiles/tools/regs.sh LIB PATTERN: vgpr / spill counts of the kernels of LIB whose name contains PATTERN
for f in $1.*gfx950; do [ -f "$f" ] || /opt/rocm/lib/llvm/bin/llvm-objdump --offloading $1 > /dev/null 2>&1; break; done
for f in $1.*gfx950; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes $f | python3 -c "
import sys,re
t=sys.stdin.read()
for blk in t.split('- .agpr_count')[1:]:
    n=re.search(r'\.name:\s+(\S+)',blk); v=re.search(r'\.vgpr_count:\s+(\d+)',blk); s=re.search(r'\.vgpr_spill_count:\s+(\d+)',blk); a=re.search(r'^:\s+(\d+)',blk)
    if n and '$2' in n.group(1): print(n.group(1)[:60], 'agpr', a.group(1) if a else '?', 'vgpr', v.group(1), 'spill', s.group(1))
"; done
