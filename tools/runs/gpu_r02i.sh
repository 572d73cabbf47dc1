export LAMD_CACHE=0 LAMD_PREP_BATCH=64 LAMD_PREP_MIN_THREADS=16384
for cfg in "1 0 0" "1 1 0" "0 0 1" "1 1 1"; do set -- $cfg
echo "== shared=$1 poison=$2 timing=$3"
DIAG_SHARED=$1 DIAG_POISON=$2 DIAG_TIMING=$3 timeout 200 python tools/diag_loop.py 1000000 20 2>&1 | grep -v amdgpu.ids | cut -c1-700
done
