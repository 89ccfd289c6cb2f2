#!/usr/bin/env python3
"""Exhaustive 0-1-principle check of the comparator tables in csrc/sort_net.h (SortNet<16>, SortNet<8>)."""
import re, sys, os
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytorch3d_pointops_amd", "csrc", "sort_net.h")).read()
def table(name, n):
    start = src.index("struct SortNet<%d> {" % n)
    m = src[start:src.index("template <int N>\n__device__ __forceinline__ void bitonic_sort", start)]
    nxt = m.find("struct SortNet<", 10)
    if nxt > 0:
        m = m[:nxt]
    arr = re.search(name + r"\[\d+\] = \{(.*?)\};", m, re.S).group(1)
    return [int(x) for x in arr.replace("\n", " ").split(",") if x.strip()]
ok = True
for n in (16, 8):
    A, B = table("kA", n), table("kB", n)
    assert len(A) == len(B)
    good = True
    for bits in range(1 << n):
        a = [(bits >> i) & 1 for i in range(n)]
        for i, j in zip(A, B):
            if a[i] > a[j]:
                a[i], a[j] = a[j], a[i]
        if any(a[k] > a[k + 1] for k in range(n - 1)):
            good = False
            break
    print(f"SortNet<{n}>: {len(A)} comparators, sorts all 2^{n} 0-1 inputs: {good}")
    ok &= good
sys.exit(0 if ok else 1)
