#!/bin/bash
# round 5, call 13: stream count / batch size variants on top of the sub-batched layers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out; export TMPDIR=/tmp
O=gpurun_out
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 "" "PF_STREAMS=3" "PF_STREAMS=4" "PF_VIT_BATCH_ALL=1" ) > $O/r5m_image_ab_streams.md 2> $O/r5m_image_ab.err
echo "== streams"; cat $O/r5m_image_ab_streams.md; tail -2 $O/r5m_image_ab.err
( timeout 600 python tools/image_ab.py --steps 4 --rounds 3 --process-num 4 "" "PF_STREAMS=3" "PF_STREAMS=4" ) > $O/r5m_image_ab_pn4.md 2> $O/r5m_image_ab_pn4.err
echo "== process_num 4"; cat $O/r5m_image_ab_pn4.md; tail -2 $O/r5m_image_ab_pn4.err
