#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 300 python scripts/two_stream_probe.py 200 2>/dev/null | tail -1
