for b in 0 32 64 96; do echo "== d$b"; SRMAP_LIB=gpurun_ab/d$b/libsrmap.so timeout 120 python tools/march_check.py --notime 2>&1 | grep -v amdgpu.ids | tail -3; done
