R=$GRAFT_REPO_ROOT; cd $R/tools
for cfg in "16 32" "16 64" "16 128" "14 64" "14 128" "13 128"; do set -- $cfg; echo "== shift $1 cap $2"; FAR3D_SUMS_SHIFT=$1 FAR3D_SUMS_CAP=$2 timeout 120 python bench_ese.py 2>&1 | grep -v amdgpu.ids; done
