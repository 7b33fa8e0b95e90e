#!/bin/bash
cd /root/repo
echo "--- alloc"; DBG_ALLOC=1 python scripts/debug_prodigy.py 2>&1 | grep -v amdgpu | cut -c1-150
git stash -q; git checkout -q 63d0fb4 2>/dev/null || git checkout -q HEAD~3; git log --oneline | head -1
make -C sliders_amd/csrc -j16 > /dev/null 2>&1
echo "--- alloc @ older tree"; DBG_ALLOC=1 python scripts/debug_prodigy.py 2>&1 | grep -v amdgpu | cut -c1-150
