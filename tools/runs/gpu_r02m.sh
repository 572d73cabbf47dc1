for lanes in 4 2 1 6; do echo "== cold engine alone, LAMD_LANES=$lanes"; LAMD_LANES=$lanes LAMD_CACHE=0 DIAG_SHARED=1 timeout 200 python tools/diag_loop.py 1000000 20 2>&1 | grep -v amdgpu.ids | tail -3; done
echo "== with a second (idle) engine alive first"; LAMD_CACHE=0 DIAG_SHARED=1 DIAG_IDLE_ENGINE=1 timeout 200 python tools/diag_loop.py 1000000 20 2>&1 | grep -v amdgpu.ids | tail -3
