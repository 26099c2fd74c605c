#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
bash tools/profile.sh c1 --workload c1 > /dev/null 2>&1
