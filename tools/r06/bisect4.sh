#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in "48,48,32;96,96,16;96,48,32;192,192,8" "48,48,32" "96,96,16" "96,48,32" "192,192,8" "48,48,32;96,96,16" "96,48,32;192,192,8"; do CASES=$c timeout 300 python tools/r06/poison2.py 2>&1 | grep "golden"; done
