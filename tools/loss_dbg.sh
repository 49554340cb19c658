cd $GRAFT_REPO_ROOT
L=dino_amd/lib/libdino_hip.so
cp $L /tmp/keep.so
for v in 3; do cp build_ab/loss_dbg$v.so $L; DINO_LOSS_DBG=$v timeout -k 10 100 python tools/loss_debug.py 65536 2 12 2>&1 | grep dbg; done
cp /tmp/keep.so $L
timeout -k 10 100 python tools/loss_debug.py 65536 2 12 2>&1 | grep "max rel"
