for w in 4 8 4 8 16; do GE_CLUSTER_WORKERS=$w python tools/seed_wait.py 30 2>&1 | tail -1; done
GE_CLUSTER_WORKERS=4 GE_GM_FIRST=0 python tools/seed_wait.py 30 2>&1 | tail -1
GE_CLUSTER_WORKERS=8 GE_GM_FIRST=0 python tools/seed_wait.py 30 2>&1 | tail -1
