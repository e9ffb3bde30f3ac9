#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python tools/exp/merge_debug2.py 2>&1 | grep -v "amdgpu.ids\|Exception ignored\|Traceback\|binding.py\|TypeError" | tail -40
