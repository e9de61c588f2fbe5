cd /tmp && export TMPDIR=/tmp
export BVH_MI355X_LIB=/root/repo/build/variants/libbvh_abl.so
for d in 1 2 3 4 0; do BVH_HPLOC_DEBUG=$d timeout 100 python /root/repo/tools/time_hploc.py block 10000000 20 2>&1 | grep "emit min"; done
bash /root/repo/tools/prof_hploc_phases.sh 2>&1 | grep -v amdgpu
