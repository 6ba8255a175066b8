export TMPDIR=/tmp
for d in 0 1 2 4 8 16 3 9 11 15 31; do echo "dbg=$d"; VITPE_T2_DBG=$d timeout -k 10 60 python tools/kb_tail.py 2>&1 | grep "gen 2"; done
