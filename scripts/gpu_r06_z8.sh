#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python scripts/rebuild_modes.py 2>/dev/null | tail -40
