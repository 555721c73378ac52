cd $GRAFT_REPO_ROOT
L=liftreg_amd/csrc
LIFTREG_CONV0_PC=1 python3 tools/abconv.py --block 0 --split --libs $L/libliftreg_hip.so,$L/libx_ah2.so,$L/libx_ah3.so,$L/libx_ah4.so 2>&1 | tail -n 8
LIFTREG_CONV0_DBG=2 python3 tools/abconv.py --block 0 --split --libs $L/libliftreg_hip.so,$L/libx_ah2.so,$L/libx_ah3.so,$L/libx_ah4.so 2>&1 | tail -n 4
