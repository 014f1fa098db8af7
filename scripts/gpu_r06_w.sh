#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout 900 python -m pytest tests/test_popstepsampler.py -m gpu -x -q -k "rounds_equal" 2>&1 | tail -5
