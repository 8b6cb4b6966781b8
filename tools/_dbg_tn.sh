for d in 0 1; do echo "dbg=$d"; LNX_TN_DBG=$d timeout -k 10 200 python tools/bench_tn.py 2>&1 | grep -E "^r0|^r1|^s" | cut -c1-60; done
