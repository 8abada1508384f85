for r in 1 2; do
for n in f32p3 f32p2 f32p1; do echo -n "$n  "; SRMAP_LIB=gpurun_ab/$n/libsrmap.so python tools/occ_probe.py f32 cfg2 | tail -1; done
for n in base f64p1; do echo -n "$n  "; SRMAP_LIB=gpurun_ab/$n/libsrmap.so python tools/occ_probe.py f64 cfg2 | tail -1; done
done
