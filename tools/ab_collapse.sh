cd /tmp && export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -k "collapse or bvh4 or wide or mirror or config4 or image" 2>&1 | grep -E "passed|failed|error" | tail -3
cd /tmp
for v in poll coll main poll main; do
  if [ "$v" = main ]; then unset BVH_MI355X_LIB; else export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_$v.so; fi
  echo "== $v"; timeout 200 python /root/repo/tools/time_collapse.py 262144 2>&1 | grep collapse4
done
unset BVH_MI355X_LIB; timeout 200 python /root/repo/tools/time_collapse.py 10000000 2>&1 | grep collapse4
