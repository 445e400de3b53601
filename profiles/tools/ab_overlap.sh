#!/bin/bash
# overlap.py per profiles/tools/libs/*.so
for lib in profiles/tools/libs/*.so; do
  [ "$(basename $lib)" = "stamps.so" ] && continue
  echo "== $(basename $lib .so)"
  MLD_HIP_LIBRARY=$PWD/$lib timeout 200 python profiles/tools/overlap.py 1024 20 2>/dev/null | grep -v verified
done
