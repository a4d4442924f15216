#!/bin/bash
# A/B builds on the GPU box: each line of flags is built and timed in turn.  usage: tools/ab_build.sh "flags1" "flags2" ...
set -e
for flags in "$@"; do
  CX_EXTRA_FLAGS="$flags" python3 contourist_amd/build.py > /dev/null 2>&1
  TAG="[$flags]" timeout -k 10 120 python3 ${AB_TOOL:-tools/quick_time.py} 512 ${AB_ARGS:-} 2>/dev/null | grep -v "^{"
done
python3 contourist_amd/build.py > /dev/null 2>&1
