#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of a command, per-kernel stats as CSV on stdout.
#   bash scripts/prof_kernels.sh <tag> <command...>
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
rocprofv3 --kernel-trace --stats -d /tmp/prof_$TAG -o kt -- "$@" > /tmp/prof_$TAG.out 2>&1
DB=$(find /tmp/prof_$TAG -name "*.db" | head -1)
python $REPO/scripts/rocpd_summary.py $DB 25
