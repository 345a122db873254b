#!/bin/bash
# Same-box A/B against the end of round 3 (git worktree .r03_tree, built in the container): scripts/ab_r03.sh <script.py> [args]
echo "== round 3 tree"; (cd .r03_tree && python "$@" 2>&1 | grep -v amdgpu.ids)
echo "== this tree";   python "$@" 2>&1 | grep -v amdgpu.ids
