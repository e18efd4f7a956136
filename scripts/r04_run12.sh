#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6
timeout 3000 bash scripts/r04_profile_all.sh 2>&1 | tail -60
