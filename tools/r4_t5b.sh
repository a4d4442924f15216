#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CX_DEBUG=1
CX_LIB_PATH=$GRAFT_REPO_ROOT/contourist_amd/lib/variants/lib_trcp.so TAG=tq_rcp timeout -k 10 200 python3 tools/stream_ab.py 512 2>&1 | grep -E "stream:|Error|error"
CX_LIB_PATH=$GRAFT_REPO_ROOT/contourist_amd/lib/variants/lib_tnost.so TAG=tq_rcp_nostore timeout -k 10 200 python3 tools/stream_ab.py 512 2>&1 | grep -E "stream:|Error|error"
CX_NO_TQ=1 TAG=gather timeout -k 10 200 python3 tools/stream_ab.py 512 2>&1 | grep -E "stream:|Error|error"
