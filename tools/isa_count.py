#!/usr/bin/env python3
"""Static instruction mix of one kernel in a gfx950 assembly listing.
Usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o k.s file.hip
       tools/isa_count.py k.s <kernel name substring>"""
import collections
import re
import sys

text = open(sys.argv[1]).read()
for m in re.finditer(r'^(\w+):.*?\.end_amdhsa_kernel', text, re.S | re.M):
    name = m.group(1)
    if sys.argv[2] not in name:
        continue
    c = collections.Counter()
    for line in m.group(0).split('\n'):
        t = line.strip().split()
        if not line.startswith('\t') or not t or t[0].startswith('.') or t[0].startswith(';'):
            continue
        i = t[0]
        kind = ('valu' if i.startswith('v_') else 'salu' if i.startswith('s_') else 'lds' if i.startswith('ds_')
                else 'vmem' if i.split('_')[0] in ('global', 'buffer', 'flat', 'scratch') else 'other')
        c[kind] += 1
    meta = re.findall(r'; (NumVgprs|NumSgprs|ScratchSize|Occupancy|LDSByteSize): ?(\d+)', text[m.end():m.end() + 3000])
    print(name, dict(c), dict(meta))
