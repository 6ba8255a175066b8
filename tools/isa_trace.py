#!/usr/bin/env python3
"""Compact instruction trace of one kernel from a hipcc -S listing: M mfma, d ds_read, w ds_write, G global load,
W global store, s/l scratch store/load, | barrier, . s_waitcnt, B branch, v other VALU (counted, printed as digits).
    tools/isa_trace.py /tmp/attn.s <mangled-name-substring>"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r"^(\S*" + re.escape(pat) + r"\S*):\s*(;.*)?\n", s, re.M)
i = m.end(); j = s.index(".Lfunc_end", i)
out, nv = [], 0
def flush():
    global nv
    if nv: out.append(f"{nv}" if nv > 1 else "v")
    nv = 0
for l in s[i:j].split("\n"):
    t = l.strip()
    c = None
    if t.startswith("v_mfma"): c = "M"
    elif t.startswith("scratch_store"): c = "s"
    elif t.startswith("scratch_load"): c = "l"
    elif t.startswith("s_barrier"): c = "|"
    elif t.startswith("global_load") or t.startswith("buffer_load"): c = "G"
    elif t.startswith("global_store") or t.startswith("buffer_store"): c = "W"
    elif t.startswith("ds_read") or t.startswith("ds_load"): c = "d"
    elif t.startswith("ds_write") or t.startswith("ds_store"): c = "w"
    elif t.startswith("s_cbranch") or t.startswith("s_branch"): c = "B"
    elif t.startswith("s_waitcnt"): c = "." + re.sub(r"s_waitcnt\s*", "", t).replace("vmcnt", "v").replace("lgkmcnt", "l").replace(" ", "") + " "
    elif t.startswith("v_") : nv += 1; continue
    else: continue
    flush(); out.append(c)
flush()
print("".join(out))
